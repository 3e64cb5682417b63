"""GPU parity: fir.java, phase.java, FECDecoder.java and the synthetic generators, through the C ABI.
Integer work is bit-exact; fir.filter's double accumulation is exact-order (bit-exact)."""
import os

import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))


# ------------------------------------------------------------------ fir.java
def test_fir_weights_and_filter_bit_exact():
    g, o = J.Fir(44100.0), O.Fir()
    assert np.array_equal(g.weights(500, 1500), o.weights(500, 1500, 44100.0))
    assert np.array_equal(g.weights(500, 1500), G["fir_w_500_1500"])
    rng = np.random.default_rng(2)
    for blk in (1, 19, 20, 21, 500, 4097):  # delay line carried across ragged blocks
        xs = rng.integers(-32768, 32768, blk).astype(np.int32)
        assert np.array_equal(g.filter_block(xs), o.filter_block(xs))
    g.weights(500, 1500)
    assert np.array_equal(g.filter_block(G["fir_in"]), G["fir_out"])


def test_fir_allpass_and_large_values():
    g, o = J.Fir(), O.Fir()
    g.weights(-2 ** 31, -2 ** 31)
    o.weights(-2 ** 31, -2 ** 31)
    xs = np.array([2 ** 31 - 1, -2 ** 31, 5, -5] * 16, np.int32)
    assert np.array_equal(g.filter_block(xs), o.filter_block(xs))


def test_fir_complex_gen_and_mod():
    g = J.Fir(44100.0)
    a = g.complex_gen(1000, 50000)
    assert np.array_equal(a[:256], G["fir_cgen_1000"])
    assert np.array_equal(a[44100:44100 + 256], a[:256])  # counter wraps at (int)rate
    assert np.array_equal(a[:4096], O.fir_complex_gen(1000, 4096))
    b = g.complex_gen(500, 4096, start=44000)
    assert np.array_equal(b, O.fir_complex_gen(500, 4096, start=44000))
    assert np.array_equal(g.complex_mod(a[:4096], b), O.fir_complex_mod(a[:4096], b))
    big = np.array([[2 ** 31 - 1, -2 ** 31], [123456789, -987654321]], np.int32)
    assert np.array_equal(g.complex_mod(big, big[::-1].copy()), O.fir_complex_mod(big, big[::-1].copy()))


# ------------------------------------------------------------------ phase.java
def test_phase_handle_is_the_drop_in_form(golden_dir):
    """jsdr_phase_* (what HipPhase.java binds): frame by frame, max and column means equal the oracle's"""
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    buf = O.convert_i16(raw)
    p = J.Phase(2048)
    assert p.max() == 0.0  # the reference paints a zero-filled dpy before the first frame (phase.java:23,75-80)
    for f in range(2):
        fr = buf[4096 * f:4096 * (f + 1)]
        p.receive(fr)
        assert p.max() == O.phase_maxabs(fr)
        for bx in (0, 1, 200, 317, 2048, 5000):
            pg, po = p.columns(bx), O.phase_columns(fr, bx)
            assert all(np.array_equal(a, b) for a, b in zip(pg, po)), (f, bx)
    with pytest.raises(J.JsdrError):
        J.Phase(0)


def test_phase_maxabs_and_columns(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    buf = O.convert_i16(raw)
    m = J.phase_maxabs(buf, 2048)
    assert m[0] == O.phase_maxabs(buf[:4096]) == G["phase_maxabs"][0]
    assert m[1] == O.phase_maxabs(buf[4096:])
    assert J.phase_maxabs(np.zeros(4096, np.float32), 2048)[0] == 0.0
    for bx in (300, 1, 2048, 5000, 0):
        pg = J.phase_columns(buf[:4096], bx)
        po = O.phase_columns(buf[:4096], bx)
        for a, b in zip(pg, po):
            assert np.array_equal(a, b)


# ------------------------------------------------------------------ FECDecoder.java
def test_fec_encode_matches_oracle():
    rng = np.random.default_rng(3)
    datas = rng.integers(0, 256, (9, 256), dtype=np.uint8)
    datas[0] = 0
    datas[1] = 255
    syms = J.fec_encode_batch(datas)
    for i in range(9):
        assert np.array_equal(syms[i], O.fec_encode(datas[i]))
    assert np.array_equal(J.fec_encode(G["fec_payload"]), G["fec_symbols"])


def test_fec_decode_roundtrip_errors_and_failure():
    rng = np.random.default_rng(4)
    raws, want = [], []
    for trial, e in enumerate((0, 1, 17, 150, 300, 450, 520, 1500)):
        data = rng.integers(0, 256, 256, dtype=np.uint8)
        soft = np.where(O.fec_encode(data) == 1, 0xC0, 0x40).astype(np.uint8)
        if e:
            soft[rng.choice(5200, e, replace=False)] ^= 0x80
        raws.append(soft)
        want.append(O.fec_decode(soft))
    raws.append(rng.integers(0, 256, 5200, dtype=np.uint8))  # garbage
    want.append(O.fec_decode(raws[-1]))
    soft = np.where(O.fec_encode(np.arange(256, dtype=np.uint8)) == 1, 128 + rng.integers(1, 128, 5200),
                    127 - rng.integers(0, 128, 5200)).astype(np.uint8)  # graded soft values through mettab
    raws.append(soft)
    want.append(O.fec_decode(soft))
    rc, out = J.fec_decode_batch(np.stack(raws))
    for i, (wrc, wout) in enumerate(want):
        assert rc[i] == wrc, i
        if wrc >= 0:
            assert np.array_equal(out[i], wout), i
    assert (rc[:4] == [0, 1, 17, 150]).all() and rc[7] == -1


def test_fec_decode_chain_back_paths_across_the_noise_range():
    """the chain-back runs lane-parallel (40 bits per lane after a warm-up) and is accepted only if the segments chain
    up; blocks from clean to far beyond the correction limit -- and soft garbage -- take both that and the serial path"""
    rng = np.random.default_rng(11)
    raws = []
    for e in range(0, 960, 20):
        data = rng.integers(0, 256, 256, dtype=np.uint8)
        soft = np.where(O.fec_encode(data) == 1, 0xC0, 0x40).astype(np.uint8)
        if e:
            idx = rng.choice(5200, e, replace=False)
            soft[idx] = rng.integers(0, 256, e, dtype=np.uint8)  # erasure-like and inverted symbols of any strength
        raws.append(soft)
    for _ in range(8):
        raws.append(rng.integers(0, 256, 5200, dtype=np.uint8))
    rc, out = J.fec_decode_batch(np.stack(raws))
    ok = fail = 0
    for i, soft in enumerate(raws):
        wrc, wout = O.fec_decode(soft)
        assert rc[i] == wrc, i
        if wrc >= 0:
            assert np.array_equal(out[i], wout), i
            ok += 1
        else:
            fail += 1
    assert ok >= 10 and fail >= 8


def test_fec_decode_byte_errors_exercise_berlekamp_massey():
    """Corrupt whole interleaver rows so that the Viterbi decoder emits byte errors and the RS stage has to
    locate and fix them (count > 0 path of decode_rs_8); must stay bit-identical to the oracle."""
    rng = np.random.default_rng(5)
    raws, want = [], []
    for burst in (40, 80, 120, 160, 240):
        data = rng.integers(0, 256, 256, dtype=np.uint8)
        soft = np.where(O.fec_encode(data) == 1, 0xC0, 0x40).astype(np.uint8)
        start = int(rng.integers(0, 5200 - burst * 8))
        # a burst in de-interleaved order = every 80th raw symbol; hit several adjacent columns hard
        for r in range(burst):
            soft[(start + r * 7) % 5200] ^= 0x80
        soft[rng.choice(5200, 420, replace=False)] ^= 0x80
        raws.append(soft)
        want.append(O.fec_decode(soft))
    rc, out = J.fec_decode_batch(np.stack(raws))
    for i, (wrc, wout) in enumerate(want):
        assert rc[i] == wrc, (i, rc[i], wrc)
        if wrc >= 0:
            assert np.array_equal(out[i], wout)


def test_fec_single_call_keeps_caller_bytes_on_failure():
    rng = np.random.default_rng(6)
    junk = rng.integers(0, 256, 5200, dtype=np.uint8)
    keep = np.arange(256, dtype=np.uint8)
    rc, out = J.fec_decode(junk, out_init=keep)
    assert rc == -1 and np.array_equal(out, keep)  # FECDecoder.java:780 leaves RSdecdata untouched
    rc, out = J.fec_decode(G["fec_soft_200err"])
    assert rc == 200 and np.array_equal(out, G["fec_dec_200err"])


# ------------------------------------------------------------------ synthetic generators
def test_synth_generators_bit_identical_to_oracle():
    seed, nstreams, nframes = 20020109, 3, 2
    pay = J.synth_payloads(seed, 10, nstreams, nframes).to_host(np.uint8).reshape(nstreams, nframes, 256)
    for s in range(nstreams):
        for f in range(nframes):
            assert np.array_equal(pay[s, f], O.synth_payload(seed, 10 + s, f))
    sym = np.stack([np.concatenate([O.fec_encode(pay[s, f]) for f in range(nframes)]) for s in range(nstreams)])
    nsym = sym.shape[1]
    d_sym = J.DeviceBuffer.from_host(sym)
    d_ds = J.DeviceBuffer(sym.size)
    J.synth_diffsign(d_sym, nsym, nstreams, d_ds)
    ds = d_ds.to_host(np.int8).reshape(nstreams, nsym)
    for s in range(nstreams):
        assert np.array_equal(ds[s], O.synth_diffsign(sym[s]))
    ct, st = O.synth_tables(3000)
    keys = np.array([O.mix64(1000 + s) for s in range(nstreams)], np.uint64)
    n, n0 = 100000, 12345
    d_out = J.DeviceBuffer(nstreams * n * 4)
    inc = O.phase_inc_u32(13200.0, 96000)
    J.synth_dbpsk(d_out, 2 * n, nstreams, n0, n, d_ds, nsym, 80, 77, inc, J.DeviceBuffer.from_host(ct),
                  J.DeviceBuffer.from_host(st), 1299, J.DeviceBuffer.from_host(keys))
    got = d_out.to_host(np.int16).reshape(nstreams, 2 * n)
    for s in range(nstreams):
        assert np.array_equal(got[s], O.synth_dbpsk(n0, n, ds[s], 80, 77, inc, ct, st, 1299, int(keys[s])))
    d_t = J.DeviceBuffer(5 * 2048 * 4)
    J.synth_tones(d_t, 3, 5, 2048, J.DeviceBuffer.from_host(ct), 300, O.mix64(42))
    assert np.array_equal(d_t.to_host(np.int16), O.synth_tones(3, 5, 2048, ct, 300, O.mix64(42)))
