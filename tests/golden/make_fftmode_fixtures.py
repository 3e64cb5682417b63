#!/usr/bin/env python3
"""Generate tests/golden/fftmode_fixtures.npz: the FFT-acquire front end (FUNcubeBPSKDemod.java:399-464, `bpsk-dofft`)
restated in pure Python from the Java text (java_restatement.DemodFFT) in front of the restated demodulator chain.

What this pins and what it cannot: the two JTransforms calls of doBufferFFT (DoubleFFT_1D.complexForward, :422-423,
complexInverse, :459) cross a dependency whose source is not in the reference tree; this project DEFINES a transform in
their place (radix-2 decimation in time, every product and sum rounded on its own, one fixed twiddle table -- the header
comment of oracle/o_fft.c).  java_restatement.fft_radix2_dit is written from that definition, its twiddle table is DATA
(taken from the definition's generator, stored here, compared with the correctly rounded values in the CPU test); the
spectrum magnitudes, the 100-wide boxcar and first maximum, the peak-power IIR and centre-bin rule, the 204-bin gather,
the inverse transform's use and RxDownSample(re, re) come from the Java text alone.  The C oracle (CPU test) and the HIP
kernels (GPU test) must reproduce centre bins, bits, counters and every state double.  Runs in the build container:

    python tests/golden/make_fftmode_fixtures.py        (about a minute)
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import java_restatement as J  # noqa: E402
import oracle_lib as O  # noqa: E402  (input generator and the twiddle table's generator only)

from fixture_cases import FFT_STREAMS, fft_stream_input  # noqa: E402

TRACE = 1024


def main():
    t0 = time.time()
    out = {}
    tw = {}
    for name, p in FFT_STREAMS.items():
        n = p["frame"]
        if n not in tw:
            tw[n] = O.fft_twiddles_f64(n).copy()
            out[f"twiddles_{n}"] = tw[n]
        raw = fft_stream_input(name)
        buf = J.convert_i16(raw)
        d = J.DemodFFT(n, [float(v) for v in tw[n]], rate=p["rate"], do_up=bool(p["do_up"]), trace_cap=TRACE)
        for f in range(p["n"] // n):
            d.receive(buf[2 * n * f:2 * n * (f + 1)])
        k = "x_" + name + "_"
        out[k + "sha256"] = np.frombuffer(hashlib.sha256(raw.tobytes()).digest(), np.uint8)
        out[k + "counters"] = np.array(d.counters(), np.int32)
        st = d.state()
        st[6], st[7] = d.avePeakPower, d.aveCentreBin
        out[k + "state"] = np.array(st, np.float64)
        out[k + "istate"] = np.array(d.istate(), np.int32)
        out[k + "bits"] = np.array(d.bits, np.int8)
        out[k + "centre"] = np.array(d.centre_log, np.int32)
        out[k + "trace"] = np.array(d.trace, np.float64).reshape(-1, 2)
        print(f"{name:6s} counters={d.counters()[:4]} centre={d.centre_log[:6]}..{d.centre_log[-1]}  [{time.time() - t0:.0f} s]", flush=True)
    path = os.path.join(HERE, "fftmode_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
