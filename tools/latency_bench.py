#!/usr/bin/env python3
"""tools/latency_bench.py -- the drop-in form: ONE stream, one 2048-sample frame per receive() call (IAudioHandler cadence:
a frame every 21.3 ms at 96 kHz).  Prints the mean host-side latency of fft.receive, FUNcubeBPSKDemod.receive (float and
raw forms) and demod.receive, each including the host->device copy of the frame and the wait for its results."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import java_sdr_amd as J  # noqa: E402

n, reps = 2048, 400
rng = np.random.default_rng(1)
raw = rng.integers(-8000, 8000, 2 * n).astype(np.int16)
buf = J.convert_i16(raw) if hasattr(J, "convert_i16") else (raw / 32767.0).astype(np.float32)
buf = np.ascontiguousarray(buf, np.float32).reshape(-1)[:2 * n]


def timeit(name, fn):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:34s} {dt * 1e6:8.1f} us per 2048-sample frame  ({21333.0 / (dt * 1e6):6.1f}x real time at 96 kHz)")


f = J.Fft(n, 96000)
timeit("fft.receive(float[])", lambda: f.receive(buf))
timeit("fft.receive(byte[]) raw", lambda: f.receive_raw(raw))
d = J.Bpsk(nstreams=1)
timeit("FUNcubeBPSKDemod.receive(float[])", lambda: d.receive(buf))
d2 = J.Bpsk(nstreams=1)
timeit("FUNcubeBPSKDemod.receive(byte[])", lambda: d2.receive_raw(raw))
d3 = J.Bpsk(nstreams=1, do_fft=1)
timeit("FUNcubeBPSKDemod FFT-acquire", lambda: d3.receive(buf))
dm = J.Demod(rate=96000, n=n, nstreams=1)
dm.configure(3, 1, 1, 1)
dm.weights(3000, 15000)
timeit("demod.receive NFM", lambda: dm.receive(buf))
# the frames other sound-card rates deliver (JavaAudio.java:58-59: a tenth of a second): fft.receive at 44.1 kHz (k_fft_rt since
# round 5; the O(n^2) kernel before) and at 11.025 kHz (k_dft_any), FUNcubeBPSKDemod FFT-acquire at 44.1 kHz
for nn, rate in ((4410, 44100), (1102, 11025)):
    rawn = rng.integers(-8000, 8000, 2 * nn).astype(np.int16)
    fn_ = J.Fft(nn, rate)
    t0 = None
    for _ in range(20):
        fn_.receive_raw(rawn)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn_.receive_raw(rawn)
    dt = (time.perf_counter() - t0) / reps
    print(f"fft.receive(byte[]) n={nn:<5d} {fn_.kernel_name():10s} {dt * 1e6:8.1f} us per frame  ({1e5 / (dt * 1e6):6.1f}x real time at {rate} Hz)")
raw44 = rng.integers(-8000, 8000, 2 * 4410).astype(np.int16)
d44 = J.Bpsk(nstreams=1, do_fft=1, rate=44100, blen=4 * 4410)
for _ in range(20):
    d44.receive_raw(raw44)
t0 = time.perf_counter()
for _ in range(reps):
    d44.receive_raw(raw44)
dt = (time.perf_counter() - t0) / reps
print(f"FUNcubeBPSKDemod FFT-acquire n=4410   {dt * 1e6:8.1f} us per frame  ({1e5 / (dt * 1e6):6.1f}x real time at 44100 Hz)")
