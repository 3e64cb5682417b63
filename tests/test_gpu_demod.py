"""GPU parity for the demod.java AM/FM chain through the C ABI: int16 audio, frame max / mean
and the carrier phase bit-identical to the CPU restatement, for every mode and switch combination, several
streams, several frames per call and state carried across calls."""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
f32 = np.float32


def same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def fm_am_signal(rng, nsamples, rate, fc=7000.0, seed_shift=0.0):
    t = np.arange(nsamples)
    msg = np.sin(2 * np.pi * (440.0 + seed_shift) * t / rate)
    ph = 2 * np.pi * fc * t / rate + 3.0 * np.cumsum(msg) * (2 * np.pi * 2500.0 / rate) / 3.0
    amp = 0.35 * (1.0 + 0.6 * np.sin(2 * np.pi * 300.0 * t / rate))
    iq = np.empty(2 * nsamples)
    iq[0::2] = amp * np.cos(ph)
    iq[1::2] = amp * np.sin(ph)
    iq += rng.standard_normal(2 * nsamples) * 0.01
    return np.clip(np.round(iq * 32767), -32768, 32767).astype(np.int16)


def run_both(mode, dofir, dodwn, doagc, n, S, frames_per_call, rate=96000, band=(3000, 11000), ic=0, qc=0):
    rng = np.random.default_rng(100 * mode + 10 * dofir + dodwn + 7 * n)
    total = sum(frames_per_call) * n
    raws = [fm_am_signal(rng, total, rate, fc=5000.0 + 900.0 * s, seed_shift=31.0 * s) for s in range(S)]
    d = J.Demod(rate=rate, n=n, nstreams=S, max_batch_samples=max(frames_per_call) * n)
    d.configure(mode, dofir, dodwn, doagc)
    oracles = [O.Demod(rate) for _ in range(S)]
    for o in oracles:
        o.configure(mode, dofir, dodwn, doagc)
    if band is not None:
        w, phi = d.weights(*band)
        ow, ophi = oracles[0].weights(*band)
        for o in oracles[1:]:
            o.weights(*band)
        assert np.array_equal(w, ow) and phi == ophi
    pos = 0
    for nf in frames_per_call:
        L = nf * n
        chunk = np.stack([r[2 * pos:2 * (pos + L)] for r in raws])
        got = d.batch_host_i16(chunk, L, ic, qc)
        for s in range(S):
            buf = O.convert_i16(chunk[s], ic=ic, qc=qc)
            for f in range(nf):
                want = oracles[s].receive(buf[2 * f * n:2 * (f + 1) * n])
                assert np.array_equal(got[s, 2 * f * n:2 * (f + 1) * n], want), (mode, dofir, dodwn, doagc, s, f)
            mx, av = d.frame_stats(s)
            assert same(mx, oracles[s].max) and same(av, oracles[s].avg)
        if dodwn:
            assert d.state()[0] == oracles[0].car
        pos += L
    return d


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
def test_demod_modes_and_switches_bit_exact(mode):
    for dofir, dodwn, doagc in ((0, 0, 0), (1, 0, 1), (1, 1, 1), (0, 1, 1), (1, 1, 0)):
        run_both(mode, dofir, dodwn, doagc, n=2048, S=3, frames_per_call=[1, 3, 2])


def test_demod_other_frame_sizes_rates_and_dc_correction():
    run_both(2, 1, 1, 1, n=9600, S=2, frames_per_call=[2, 1])            # the reference's default frame
    run_both(3, 1, 1, 1, n=9600, S=2, frames_per_call=[1, 2], ic=37, qc=-1200)
    run_both(4, 1, 0, 1, n=4800, S=2, frames_per_call=[2], rate=48000, band=(-9000, 2000))
    run_both(2, 1, 1, 0, n=256, S=5, frames_per_call=[4, 4], rate=44100, band=(100, 4000))
    run_both(3, 1, 1, 1, n=2051, S=2, frames_per_call=[2, 2])            # frame that is not a multiple of anything
    run_both(3, 1, 1, 1, n=19200, S=2, frames_per_call=[1, 1], rate=192000)  # > 5 tiles: the three-kernel path in an FM mode
    run_both(1, 0, 1, 1, n=10240, S=1, frames_per_call=[2])               # exactly 5 full tiles, fused


def test_demod_default_weights_are_zero_like_the_reference():
    # no weights() call (filterMove's range check fails at the default filter points): filter on -> silence
    run_both(1, 1, 0, 0, n=2048, S=1, frames_per_call=[2], band=None)


def test_demod_receive_float_frames_and_edge_values():
    n = 512
    d = J.Demod(rate=48000, n=n, nstreams=1)
    o = O.Demod(48000)
    rng = np.random.default_rng(8)
    for mode, doagc in ((1, 1), (2, 1), (3, 0)):
        d.configure(mode, 0, 0, doagc)
        o.configure(mode, 0, 0, doagc)
        for k in range(3):
            buf = (rng.standard_normal(2 * n) * 0.3).astype(f32)
            if k == 1:
                buf[:] = 0            # silence with AGC: 0 * (1/0) = NaN -> 0
            if k == 2:
                buf[10] = np.nan      # Math.max propagates the NaN into the AGC factor
                buf[20] = 7.0e4       # saturates (int), wraps (short)
            got = d.receive(buf)
            want = o.receive(buf)
            assert np.array_equal(got, want), (mode, k)
            mx, av = d.frame_stats(0)
            assert same(mx, o.max) and same(av, o.avg)


def test_demod_int_cast_saturation_nan_and_infinities():
    """(short)(sam * 32767f) (:478-481) at the edges of Java's float -> int rule: NaN -> 0, +-inf and anything beyond
    the int range saturate (then wrap to short), everything else truncates toward zero"""
    n = 256
    d = J.Demod(rate=48000, n=n, nstreams=1)
    o = O.Demod(48000)
    for mode in (1, 2):      # RAW: the fused kernel; AM: the three-kernel path (mean subtraction on top)
        d.configure(mode, 0, 0, 0)
        o.configure(mode, 0, 0, 0)
        buf = np.zeros(2 * n, f32)
        edge = [7.0e4, -7.0e4, np.inf, -np.inf, np.nan, 65535.9, -65536.1, 65538.0, -65538.5, 3.0e38, -3.0e38, 0.99999,
                -0.99999, 1.0000001, -1.0000001, 2.5 / 32767, -2.5 / 32767, 1e-40, -0.0, 1e-20, 3.3e-20, 7.7e-23, 1.1e19]
        buf[0:2 * len(edge):2] = np.array(edge, f32)
        got = d.receive(buf)
        want = o.receive(buf)
        assert np.array_equal(got, want), mode


def test_demod_api_errors():
    with pytest.raises(J.JsdrError):
        J.Demod(rate=0)
    d = J.Demod(n=2048, nstreams=2, max_batch_samples=4096)
    with pytest.raises(J.JsdrError):
        d.configure(7)
    with pytest.raises(J.JsdrError):
        d.receive(np.zeros(4096, f32))  # frame-by-frame form is 1-stream
    buf = J.DeviceBuffer(2 * 2 * 8192)
    with pytest.raises(J.JsdrError):
        d.batch_i16(buf, 2 * 3000, 3000, buf, 2 * 3000)  # not whole frames
    with pytest.raises(J.JsdrError):
        d.batch_i16(buf, 2 * 8192, 8192, buf, 2 * 8192)  # beyond max_batch_samples
    with pytest.raises(J.JsdrError):
        d.frame_stats(0)  # nothing processed yet


def test_demod_full_length_batch_sampled_streams_vs_oracle():
    """BASELINE's batch length (2^20 samples per stream, 512 frames of 2048) in one call: the first and the last
    stream against the oracle, every frame -- the FM detector / filter / NCO state run through all 512 frames"""
    rate, n, S, L = 96000, 2048, 8, 1 << 20
    rng = np.random.default_rng(21)
    base = fm_am_signal(rng, L, rate, fc=6500.0)
    raws = [np.roll(base, 2 * 997 * s) for s in range(S)]  # cheap distinct streams (circular shifts, I/Q kept paired)
    for mode in (2, 3):
        d = J.Demod(rate=rate, n=n, nstreams=S, max_batch_samples=L)
        d.configure(mode, 1, 1, 1)
        d.weights(2000, 12000)
        got = d.batch_host_i16(np.stack(raws), L)
        for s in (0, S - 1):
            o = O.Demod(rate)
            o.configure(mode, 1, 1, 1)
            o.weights(2000, 12000)
            buf = O.convert_i16(raws[s])
            for f in range(L // n):
                want = o.receive(buf[2 * f * n:2 * (f + 1) * n])
                assert np.array_equal(got[s, 2 * f * n:2 * (f + 1) * n], want), (mode, s, f)
            mx, av = d.frame_stats(s)
            assert same(mx, o.max) and same(av, o.avg)
        assert d.state()[0] == o.car
