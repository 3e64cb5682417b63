"""The C++ mirror of the reference's plugin surface (java-sdr_amd/host) driven by the headless audio loop,
on the reference's own input fixture: fft, phase and FUNcubeBPSKDemod handlers fed frame by frame."""
import os
import re
import subprocess

import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "java-sdr_amd", "host", "jsdr_harness")


def test_harness_on_sine4410(golden_dir):
    assert os.path.exists(HARNESS), "build with __graft_entry__.build() (make -C java-sdr_amd/host)"
    fx = os.path.join(golden_dir, "sine4410.raw")
    r = subprocess.run([HARNESS, fx, "96000", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    assert len(lines) == 2
    raw = np.fromfile(fx, dtype="<i2")
    buf = O.convert_i16(raw)
    want_bits = (25, 50)
    for k, line in enumerate(lines):
        m = re.search(r"max (\S+) dB @ (\S+) Hz phase-max (\S+) bpsk raw=(\d+) ds=(\d+) bit=(\d+) fec=(\d+) dec=(\d+) tune=(\d+)", line)
        assert m, line
        ref = O.fft_receive(buf[k * 4096:(k + 1) * 4096], 96000)
        assert abs(float(m.group(1)) - ref[2049]) < 1e-3
        assert abs(float(m.group(2))) == 9609.0
        assert abs(float(m.group(3)) - O.phase_maxabs(buf[k * 4096:(k + 1) * 4096])) < 1e-6
        assert int(m.group(4)) == 2048 * (k + 1)
        assert int(m.group(6)) == want_bits[k] and int(m.group(7)) == 0
        assert int(m.group(9)) == 12000


def test_harness_at_the_reference_default_frame(tmp_path):
    """unmodified jsdr uses blen = rate*size/10 = 38400 bytes -> n = 9600 samples (JavaAudio.java:58-59)"""
    n = 9600
    iq, _, _ = O.make_dbpsk_stream(3, 0, 2 * n)
    fx = tmp_path / "two_frames.raw"
    iq.tofile(fx)
    r = subprocess.run([HARNESS, str(fx), "96000", "38400"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    assert len(lines) == 2
    buf = O.convert_i16(iq)
    o = O.Bpsk(blen=38400)
    for k, line in enumerate(lines):
        m = re.search(r"max (\S+) dB @ (\S+) Hz .* ds=(\d+) bit=(\d+)", line)
        ref = O.fft_receive(buf[k * 2 * n:(k + 1) * 2 * n], 96000)
        assert abs(float(m.group(1)) - ref[n + 1]) < 1e-3 and float(m.group(2)) == ref[n]
        o.receive(buf[k * 2 * n:(k + 1) * 2 * n])
        assert int(m.group(3)) == o.counters()["cntDS"] and int(m.group(4)) == o.counters()["cntBit"]


def test_harness_reports_handler_failure_like_the_audio_loop(golden_dir):
    # blen 100000 -> n=25000: no kernel for that frame size (above 20000 samples) -> the fft plugin's setup fails, the
    # loop ends with a status message and a non-zero exit instead of wrong data
    fx = os.path.join(golden_dir, "sine4410.raw")
    r = subprocess.run([HARNESS, fx, "96000", "100000"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "Audio oops" in r.stderr


def test_harness_reads_wav_like_openfile_and_paints_the_waterfall(golden_dir):
    """JavaAudio.openFile (:369-395): the reference's own WAV fixture (44.1 kHz stereo PCM-16) is accepted at the
    matching rate, refused ("Incompatible audio format") at another; the waterfall listener's top row puts the
    tone where waterfall.paintLine would"""
    import wave
    fx = os.path.join(golden_dir, "sine4410.wav")
    r = subprocess.run([HARNESS, fx, "44100", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    with wave.open(fx, "rb") as w:
        data = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    assert len(lines) == data.size // 4096
    buf = O.convert_i16(data)
    for k in (0, 1, len(lines) - 1):
        m = re.search(r"max (\S+) dB @ (\S+) Hz .* wf-peak-col=(\d+)", lines[k])
        ref = O.fft_receive(buf[k * 4096:(k + 1) * 4096], 44100)
        assert abs(float(m.group(1)) - ref[2049]) < 1e-3 and float(m.group(2)) == ref[2048]
        row = O.waterfall_line(ref, 2048, 1024)
        want_cols = np.flatnonzero((row & 0xFF) == (row & 0xFF).max())
        assert int(m.group(3)) in want_cols
    r = subprocess.run([HARNESS, fx, "96000", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "Incompatible audio format" in r.stderr


def test_harness_demod_handler_matches_oracle(golden_dir):
    """the demod mirror in the audio loop (AM, filter + AGC on, band -12..-6 kHz around the fixture's tone):
    per-frame max / avg and the sum of the audio bytes equal the oracle's"""
    fx = os.path.join(golden_dir, "sine4410.raw")
    r = subprocess.run([HARNESS, fx, "96000", "8192"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    buf = O.convert_i16(np.fromfile(fx, dtype="<i2"))
    o = O.Demod(96000)
    o.configure(2, 1, 0, 1)
    o.d.flo, o.d.fhi = -12000, -6000
    assert o.filter_move(0, 0)  # setup() -> filterMove(0,0) -> weights()
    for k, line in enumerate(lines):
        m = re.search(r"am-max=(\S+) am-avg=(\S+) am-sum=(-?\d+)", line)
        audio = o.receive(buf[k * 4096:(k + 1) * 4096])
        assert np.float32(float(m.group(1))) == o.max and np.float32(float(m.group(2))) == o.avg
        assert int(m.group(3)) == int(audio.astype(np.int64).sum())


def test_harness_at_the_192k_default_frame(tmp_path):
    """FUNcube Dongle Pro+ rate: blen = 192000*4/10 = 76800 bytes -> n = 19200 samples per frame: every handler
    of the harness (fft as 2 x 9600 + combine, phase, waterfall, BPSK at decimation 20, demod) takes it"""
    n = 19200
    iq, _, _ = O.make_dbpsk_stream(4, 0, 2 * n, rate=192000)
    fx = tmp_path / "two_frames_192k.raw"
    iq.tofile(fx)
    r = subprocess.run([HARNESS, str(fx), "192000", "76800"], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("frame")]
    assert len(lines) == 2
    buf = O.convert_i16(iq)
    o = O.Bpsk(rate=192000, blen=76800)
    for k, line in enumerate(lines):
        m = re.search(r"max (\S+) dB @ (\S+) Hz .* ds=(\d+) bit=(\d+)", line)
        ref = O.fft_receive(buf[k * 2 * n:(k + 1) * 2 * n], 192000)
        assert abs(float(m.group(1)) - ref[n + 1]) < 1e-3 and float(m.group(2)) == ref[n]
        o.receive(buf[k * 2 * n:(k + 1) * 2 * n])
        assert int(m.group(3)) == o.counters()["cntDS"] and int(m.group(4)) == o.counters()["cntBit"]


def test_bpsk_snapshot_for_a_concurrent_reader():
    """outputs double-buffered for a reader on another thread (the reference's Swing thread paints the
    demodulator's fields while the audio thread is inside receive()).  A reader thread polls the snapshot without any
    lock while the main thread feeds frames; every snapshot must be one whole frame's results (never torn), frames
    must only move forward, and the final one must equal the getters."""
    import threading
    import oracle_lib as O
    n = 2048 * 120
    iq, _, _ = O.make_dbpsk_stream(8, 0, n, noise_sigma=700.0)
    buf = O.convert_i16(iq)
    d = J.Bpsk(nstreams=1)
    with pytest.raises(J.JsdrError):
        d.snapshot()  # nothing received yet
    seen, stop, bad = [], threading.Event(), []

    def reader():
        last = 0
        while not stop.is_set():
            try:
                sn = d.snapshot()
            except J.JsdrError:
                continue
            c = list(sn.counters)
            if sn.frames < last or c[0] != 2048 * sn.frames or c[1] != (2048 * sn.frames) // 10:
                bad.append((sn.frames, c[:3]))
            last = sn.frames
            seen.append(sn.frames)

    t = threading.Thread(target=reader)
    t.start()
    for k in range(120):
        d.receive(buf[k * 4096:(k + 1) * 4096])
    stop.set()
    t.join()
    assert not bad, bad[:5]
    assert len(set(seen)) > 3  # the reader really ran beside the writer
    sn = d.snapshot()
    assert sn.frames == 120
    c = d.counters()
    assert list(sn.counters) == [c[k] for k in J.binding.COUNTER_NAMES]
    assert np.array_equal(np.array(sn.state[:]), d.state())
    assert np.array_equal(np.frombuffer(bytes(sn.decoded), np.uint8), d.decoded())
    assert np.array_equal(np.array(sn.bits[:sn.nbits], np.int8), d.bits())


@pytest.mark.parametrize("form", ["raw", "float", "float_offgrid"])
def test_bpsk_receive_frame_by_frame_through_fec_frames(form):
    """The drop-in form end to end: one stream, 2048-sample frames through receive(), across two whole FEC frames.  Round 3
    moved work INTO kernels for this form -- the hit ordering into k_sync_t, FEC stage 2 and the snapshot pack into the
    block of k_fec_bpsk that finishes last, a short-call k_fm, the tables behind the frame in one copy, float frames that
    are (float)s/32767f values through the int16 kernels -- so after EVERY frame the snapshot (what the Java getters
    read) must equal the oracle fed the same frames: counters, state doubles, decoded[], the frame's bits.
    "float_offgrid": floats that are NOT conversions of shorts (scaled by 0.999) keep the float kernels."""
    import oracle_lib as O
    n = 2048 * 420  # 860 160 samples: two sync hits
    iq, pay, _ = O.make_dbpsk_stream(20020109, 3, n, noise_sigma=600.0)
    buf = O.convert_i16(iq)
    if form == "float_offgrid":
        buf = (buf * np.float32(0.999)).astype(np.float32)
    d = J.Bpsk(nstreams=1)
    o = O.Bpsk(trace=8)
    names = J.binding.COUNTER_NAMES
    hits = 0
    for k in range(n // 2048):
        fr_f = buf[k * 4096:(k + 1) * 4096]
        if form == "raw":
            d.receive_raw(iq[k * 4096:(k + 1) * 4096])
            o.receive_i16(iq[k * 4096:(k + 1) * 4096])
        else:
            d.receive(fr_f)
            o.receive(fr_f)
        sn = d.snapshot()
        oc = o.counters()
        assert sn.frames == k + 1
        assert list(sn.counters) == [oc[x] for x in names], (k, list(sn.counters), oc)
        if k % 37 == 0 or oc["cntFEC"] != hits:
            assert np.array_equal(np.array(sn.state[:]), o.state()), k
            assert np.array_equal(np.frombuffer(bytes(sn.decoded), np.uint8), o.decoded()), k
        hits = oc["cntFEC"]
    assert hits == 2 and o.counters()["cntDec"] == 2
    assert np.array_equal(d.decoded(), pay[1]) if form != "float_offgrid" else True
    assert d.front_kernel_name() == ("k_front" if form == "float_offgrid" else "k_fm")
