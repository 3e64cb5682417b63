"""Inputs of tests/golden/reference_fixtures.npz: shared by the generator (make_reference_fixtures.py, build
container only) and by the tests that hold the C oracle and the HIP path to the fixtures.

Inputs are data: the reference's own sine4410.raw, and synthetic DBPSK streams from the repo's integer generator
(oracle/o_synth.c == csrc/synth.hip); the fixture stores their sha256, not the samples.
"""
import os

import numpy as np

import oracle_lib as O  # input generator only (jo_synth_*)

HERE = os.path.dirname(os.path.abspath(__file__))

STREAMS = {
    "sine": dict(rate=96000, tuning=12000, ic=0, qc=0, n=4096, synth=None),
    "clean": dict(rate=96000, tuning=12000, ic=0, qc=0, n=458752, synth=dict(seed=20020109, stream=5, noise_sigma=300.0)),
    "noisy": dict(rate=96000, tuning=12000, ic=0, qc=0, n=458752, synth=dict(seed=20020109, stream=5, noise_sigma=3600.0)),
    "fail": dict(rate=96000, tuning=12000, ic=0, qc=0, n=458752, synth=dict(seed=20020109, stream=5, noise_sigma=4500.0)),
    "r48k_dc": dict(rate=48000, tuning=12000, ic=37, qc=-21, n=233472, synth=dict(seed=20020110, stream=2, noise_sigma=1500.0)),
    "negtune": dict(rate=96000, tuning=-1000, ic=0, qc=0, n=40960, synth=dict(seed=20020111, stream=1, noise_sigma=1500.0)),
    "r44k1": dict(rate=44100, tuning=12000, ic=0, qc=0, n=102400, synth=dict(seed=20020113, stream=3, noise_sigma=1500.0)),
    "r192k": dict(rate=192000, tuning=12000, ic=-5, qc=9, n=204800, synth=dict(seed=20020114, stream=4, noise_sigma=1500.0)),
}
FEC_CASES = ["clean", "flips200", "burst400", "soft", "flips350", "flips520", "flips700", "garbage", "rs8", "rs16", "rs16_0", "rs17"]
TRACE = 2048


def stream_input(name):
    p = STREAMS[name]
    if p["synth"] is None:
        raw = np.fromfile(os.path.join(HERE, "sine4410.raw"), dtype="<i2")
    else:
        s = p["synth"]
        raw, _, _ = O.make_dbpsk_stream(s["seed"], s["stream"], p["n"], rate=p["rate"], carrier_hz=13200.0, amp=3000,
                                        noise_sigma=s["noise_sigma"])
    assert raw.size == 2 * p["n"], (name, raw.size)
    return raw


# ---- FFT-acquire mode (tests/golden/fftmode_fixtures.npz, make_fftmode_fixtures.py): frame = samples per receive()
FFT_STREAMS = {
    "sine": dict(rate=96000, frame=2048, do_up=0, n=4096, synth=None),
    "clean": dict(rate=96000, frame=2048, do_up=0, n=65536, synth=dict(seed=20020121, stream=3, noise_sigma=300.0, carrier=13200.0)),
    "noisy": dict(rate=96000, frame=2048, do_up=0, n=49152, synth=dict(seed=20020122, stream=1, noise_sigma=3000.0, carrier=13200.0)),
    "upper": dict(rate=96000, frame=2048, do_up=1, n=49152, synth=dict(seed=20020123, stream=2, noise_sigma=600.0, carrier=30000.0)),
    "f4096": dict(rate=96000, frame=4096, do_up=0, n=65536, synth=dict(seed=20020124, stream=4, noise_sigma=600.0, carrier=13200.0)),
    "f1024": dict(rate=48000, frame=1024, do_up=0, n=32768, synth=dict(seed=20020125, stream=6, noise_sigma=600.0, carrier=6000.0)),
}


def fft_stream_input(name):
    p = FFT_STREAMS[name]
    if p["synth"] is None:
        raw = np.fromfile(os.path.join(HERE, "sine4410.raw"), dtype="<i2")
    else:
        s = p["synth"]
        raw, _, _ = O.make_dbpsk_stream(s["seed"], s["stream"], p["n"], rate=p["rate"], carrier_hz=s["carrier"], amp=3000,
                                        noise_sigma=s["noise_sigma"])
    assert raw.size == 2 * p["n"], (name, raw.size)
    return raw
