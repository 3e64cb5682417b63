"""The JNI shim and the Java plugin classes are committed as complete sources (jni/, java/) but cannot be compiled
against a JDK here (none in the image).  What can be checked without one:
  * every native method of HipNative.java has exactly one JNI function in jsdr_jni.c, with the matching argument
    count, and vice versa;
  * every native method the Hip* plugin classes call exists;
  * every jsdr_* function the shim calls is declared in include/jsdr_hip.h and exported by libjsdr_hip.so;
  * the shim type-checks (gcc -fsyntax-only -Wall -Werror) against prototype-only JNI declarations;
  * the plugin classes implement the reference's handler interfaces, keep its constructor shapes and ARE what jsdr hosts:
    IUIComponent subclasses with hotKey and paintComponent (IAudioHandler.java:3-6, IRawHandler.java:3-6,
    IUIComponent.java:5-7, jsdr.java:475-483)."""
import os
import re
import subprocess

import java_sdr_amd as J

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JDIR = os.path.join(ROOT, "java", "com", "ashbysoft", "java_sdr")
SHIM = os.path.join(ROOT, "jni", "jsdr_jni.c")
PREFIX = "Java_com_ashbysoft_java_1sdr_HipNative_"


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    return re.sub(r"//[^\n]*", " ", src)


def native_methods():
    src = strip_comments(open(os.path.join(JDIR, "HipNative.java")).read())
    out = {}
    for m in re.finditer(r"static\s+native\s+\w+(?:\[\])?\s+(\w+)\s*\(([^)]*)\)\s*;", src):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def jni_functions():
    src = strip_comments(open(SHIM).read())
    out = {}
    for m in re.finditer(r"JNIEXPORT\s+\w+\s+JNICALL\s+" + PREFIX + r"(\w+)\s*\(([^)]*)\)", src):
        out[m.group(1)] = len([a for a in m.group(2).split(",") if a.strip()])
    return out


def test_every_native_method_has_its_jni_function_and_vice_versa():
    nat, jni = native_methods(), jni_functions()
    assert len(nat) == 29  # 22 of the four plugin classes + 7 of the group (round 5)
    assert set(nat) == set(jni)
    for name, nargs in nat.items():
        assert jni[name] == nargs + 2, (name, nargs, jni[name])  # JNIEnv*, jclass + the Java arguments


def test_plugin_classes_call_only_declared_native_methods():
    nat = native_methods()
    used = set()
    for fn in ("HipFft.java", "HipFUNcubeBPSKDemod.java", "HipDemod.java", "HipPhase.java", "HipDemodGroup.java"):
        src = strip_comments(open(os.path.join(JDIR, fn)).read())
        used |= set(re.findall(r"HipNative\.(\w+)\s*\(", src))
    assert used and used <= set(nat), used - set(nat)
    # nothing declared is dead, except fecDecode (a utility for code that holds soft symbols itself)
    assert set(nat) - used <= {"fecDecode"}


def test_shim_calls_only_functions_of_the_c_abi():
    src = strip_comments(open(SHIM).read())
    called = set(re.findall(r"\b(jsdr_[a-z0-9_]+)\s*\(", src))
    header = strip_comments(open(os.path.join(ROOT, "include", "jsdr_hip.h")).read())
    declared = set(re.findall(r"\b(jsdr_[a-z0-9_]+)\s*\(", header))
    assert called and called <= declared, called - declared
    assert called <= set(J.EXPORTED_SYMBOLS), called - set(J.EXPORTED_SYMBOLS)
    lib = J.lib()
    for name in called:
        assert hasattr(lib, name), name


def test_shim_type_checks_against_prototype_only_jni_declarations():
    r = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                        "-I" + os.path.join(ROOT, "tests", "jni_decls"), "-I" + os.path.join(ROOT, "include"), SHIM],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_plugin_classes_keep_the_reference_surface():
    fft = strip_comments(open(os.path.join(JDIR, "HipFft.java")).read())
    bpsk = strip_comments(open(os.path.join(JDIR, "HipFUNcubeBPSKDemod.java")).read())
    dem = strip_comments(open(os.path.join(JDIR, "HipDemod.java")).read())
    pha = strip_comments(open(os.path.join(JDIR, "HipPhase.java")).read())
    for src in (fft, bpsk, dem, pha):
        assert "package com.ashbysoft.java_sdr;" in src
        # jsdr.java:475-483 hosts its plugins with tabs.add(name, new X(...)): a Swing component, i.e. the reference's
        # IUIComponent (IUIComponent.java:5-7: abstract class extends JPanel, abstract hotKey(char))
        assert re.search(r"public class Hip\w+ extends IUIComponent implements\s+IAudioHandler", src)
        assert "public void hotKey(char c)" in src
        assert "public void paintComponent(Graphics g)" in src and "import java.awt.Graphics;" in src
        assert "repaint();" in src
        assert re.search(r"implements\s+IAudioHandler", src)
        assert "public synchronized void receive(float[] buf)" in src
        assert '"audio-change".equals(key)' in src
        assert not re.search(r"same pattern|TODO|\.\.\.", src)
    for src in (fft, bpsk):
        assert "IRawHandler" in src and "public synchronized void receive(byte[] raw)" in src
    assert re.search(r"public HipFft\(IConfig \w+, IPublish \w+, ILogger \w+, IUIHost \w+, IAudio \w+\)", fft)
    assert re.search(r"public HipPhase\(IConfig \w+, IPublish \w+, ILogger \w+, IUIHost \w+, IAudio \w+\)", pha)  # jsdr.java:475
    assert re.search(r"public HipDemod\(IConfig \w+, IPublish \w+, ILogger \w+, IUIHost \w+, IAudio \w+\)", dem)
    assert re.search(r"public HipFUNcubeBPSKDemod\(int \w+, IConfig \w+, IPublish \w+, ILogger \w+, IUIHost \w+, IAudio \w+\)", bpsk)
    for key in ('"bpsk-tuning"', '"bpsk-dofft"', '"bpsk-upper"', '"-bpsk-centre"', '"-bpsk-tune"'):
        assert key in bpsk, key
    for key in ('"demod-filter-low"', '"demod-filter-high"', '"demod-mode"', '"demod-fir-enable"', '"demod-agc-enable"'):
        assert key in dem, key
    assert '"fft-psd"' in fft


def test_shim_never_calls_the_device_inside_a_critical_region_and_checks_lengths():
    """no GetPrimitiveArrayCritical around GPU calls (the JNI specification forbids blocking there); every
    receive compares the Java array's length with the handle's frame before it copies"""
    src = strip_comments(open(SHIM).read())
    assert "GetPrimitiveArrayCritical" not in src
    for fn in ("fftReceive", "fftReceiveRaw", "bpskReceive", "bpskReceiveRaw", "demodReceive", "phaseReceive"):
        body = src[src.index(PREFIX + fn + "("):]
        body = body[:body.index("\n}\n")]
        assert "bad_length(" in body and "null_handle(" in body, fn
        assert body.index("bad_length(") < body.index("ArrayRegion("), fn
    assert "IllegalArgumentException" in src


def test_java_handles_are_zeroed_before_a_create_that_may_throw():
    """setup() must not keep a freed native pointer when the following create throws"""
    for fn, kind in (("HipFft.java", "fft"), ("HipFUNcubeBPSKDemod.java", "bpsk"), ("HipDemod.java", "demod"),
                     ("HipPhase.java", "phase")):
        src = strip_comments(open(os.path.join(JDIR, fn)).read())
        setup = src[src.index("void setup(IAudio"):]
        setup = setup[:setup.index("\n    }\n")]
        assert setup.index("handle = 0;") < setup.index("HipNative.%sDestroy(old)" % kind) < setup.index("HipNative.%sCreate(" % kind), fn
        assert "HipNative.%sDestroy(handle)" % kind not in src, fn


def test_painting_never_takes_the_receive_monitor_except_where_the_reference_does():
    """paintComponent runs on the Swing EDT while the audio thread holds the plugin's monitor for the whole of receive()
    (SURVEY 8b): fft / FUNcube / demod paint from copies under their own small locks; phase.java's own paintComponent is
    synchronized (phase.java:42), so HipPhase's may wait for a frame as well"""
    for fn in ("HipFft.java", "HipFUNcubeBPSKDemod.java", "HipDemod.java"):
        src = strip_comments(open(os.path.join(JDIR, fn)).read())
        body = src[src.index("public void paintComponent(Graphics g)"):]
        body = body[:body.index("\n    }\n")]
        assert "HipNative." not in body, fn            # no device call from the painter
        assert not re.search(r"synchronized\s*\(\s*this\s*\)", body), fn


def test_painter_getters_use_the_lock_free_snapshot():
    src = strip_comments(open(os.path.join(JDIR, "HipFUNcubeBPSKDemod.java")).read())
    assert "HipNative.bpskSnapshot(" in src
    for getter in ("getCounters", "getState", "getDecoded", "getBits", "isDecodeOK"):
        m = re.search(r"public (synchronized )?[\w\[\]]+ %s\(" % getter, src)
        assert m and not m.group(1), getter  # not on the receive() monitor
