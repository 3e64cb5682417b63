#!/usr/bin/env python3
"""Build libjsdr_hip.so (gfx950 only) in-tree with hipcc.  No CPU fallback is built.

    python java-sdr_amd/build.py [--force] [--verbose]

Per-file flags: the exact-order FP64 translation units are compiled with -ffp-contract=off so that
every multiply and add rounds separately, as Java's do.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libjsdr_hip.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result"]
SOURCES = {
    "runtime.hip": [],
    "fft_psd.hip": [],
    "fft_mixed.hip": [],
    "fft_any.hip": ["-ffp-contract=off"],
    "fft_rt.hip": [],
    "fir_phase.hip": ["-ffp-contract=off"],
    "fir_batch.hip": ["-ffp-contract=off"],
    "fec.hip": [],
    "synth.hip": [],
    "formats.hip": [],
    "demod.hip": ["-ffp-contract=off"],
    "bpsk.hip": ["-ffp-contract=off"],
    "bpsk_fft.hip": ["-ffp-contract=off"],
    # (no atomic optimiser: it turns the one-lane ticket atomicAdd into atomic + s_waitcnt vmcnt(0) + v_readfirstlane on the spot,
    #  i.e. a round trip to memory at the top of a frame for a value that is wanted a pass later)
    "bpsk_acq.hip": ["-ffp-contract=off", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"],
    "bpsk_fftm.hip": ["-ffp-contract=off"],
    "bpsk_acqg.hip": ["-ffp-contract=off"],
    "group.hip": [],
}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "jsdr_hip.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(op)
        if force or newer(op, [sp] + headers):
            jobs.append([hipcc()] + COMMON + extra + ["-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if force or jobs or newer(OUT, objs):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lpthread"])
    return OUT


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
