"""Known-answer and round-trip tests that pin the CPU oracle.

The reference has no tests and no golden vectors; what pins results is (i) its literal tables
(test_oracle_tables.py), (ii) the behavioural KATs SURVEY.md section 8c derived from the Java text,
(iii) decode(encode(x)) round trips through the restated encoder, (iv) independent numpy
restatements of the float paths written here in the tests.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))


def sine_buf(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    return raw, O.convert_i16(raw)


# ------------------------------------------------------------------ conversion (JavaAudio.java:276-293)
def test_convert_matches_numpy_float_division():
    raw = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    iq = np.stack([raw, raw[::-1]], axis=1).reshape(-1)
    got = O.convert_i16(iq)
    want = (iq.astype(np.float32) / np.float32(32767.0)).astype(np.float32)
    assert np.array_equal(got, want)


def test_convert_dc_correction_wraps_like_java_short():
    iq = np.array([32767, -32768, 100, -100], dtype=np.int16)
    got = O.convert_i16(iq, ic=5, qc=-7)
    wi = np.array([32767 + 5, 100 + 5]).astype(np.int64)
    wq = np.array([-32768 - 7, -100 - 7]).astype(np.int64)
    wrap = lambda v: ((v + 32768) % 65536 - 32768).astype(np.float32) / np.float32(32767)
    assert np.array_equal(got[0::2], wrap(wi))
    assert np.array_equal(got[1::2], wrap(wq))


def test_convert_mono_sets_q_zero():
    got = O.convert_i16(np.array([1, 2, 3], np.int16), chns=1)
    assert np.array_equal(got[1::2], np.zeros(3, np.float32))
    assert np.array_equal(got[0::2], np.array([1, 2, 3], np.float32) / np.float32(32767))


# ------------------------------------------------------------------ fft.java
def test_fft_sine4410_behavioural_kat(golden_dir):
    _, buf = sine_buf(golden_dir)
    for f, want_db in ((0, -2.513), (1, -2.522)):
        psd = O.fft_receive(buf[f * 4096:(f + 1) * 4096], 96000)
        k = int(np.argmax(psd[:2048]))
        assert k in (205, 1843)
        assert abs(psd[2049] - want_db) < 2e-3
        assert psd[2049] == psd[k]
        assert abs(psd[205] - psd[1843]) < 1e-4
        assert psd[2048] == (9609.0 if k == 205 else -9609.0)
        psd44 = O.fft_receive(buf[f * 4096:(f + 1) * 4096], 44100)
        assert psd44[2048] == (4414.0 if k == 205 else -4414.0)
        assert np.array_equal(psd, G["sine_psd96k"][f])


def test_fft_zero_frame_edge_case():
    psd = O.fft_receive(np.zeros(4096, np.float32), 96000)
    assert np.all(np.isneginf(psd[:2048]))
    assert psd[2048] == -23.0  # p=-1 -> -1*96000/4096 truncates toward zero
    assert psd[2049] == -np.finfo(np.float32).max


def test_fft_hz_rule_int_arithmetic():
    # single complex tone at bin k: Hz = 2k*rate/(2n) (trunc) below n/2, (2k-2n)*rate/(2n) above
    n = 2048
    for k in (1, 300, 1023, 1024, 1500, 2047):
        t = np.arange(n)
        x = 0.5 * np.exp(2j * np.pi * k * t / n)
        buf = np.empty(2 * n, np.float32)
        buf[0::2] = x.real
        buf[1::2] = x.imag
        psd = O.fft_receive(buf, 96000)
        assert int(np.argmax(psd[:n])) == k
        p = 2 * k
        want = int(p * 96000 / (2 * n)) if p < n else int((p - 2 * n) * 96000 / (2 * n))
        assert psd[n] == float(want)
        assert abs(psd[n + 1] - 10 * np.log10((0.5 * n) ** 2 * (2 / n) ** 2)) < 1e-3


def test_fft_first_strict_maximum_wins():
    spec = np.zeros(4096, np.float32)
    spec[2 * 7] = 3.0
    spec[2 * 900 + 1] = 3.0  # identical power later: must not replace the first
    psd = O.fft_psd_from_spectrum(spec, 96000)
    assert psd[2048] == float(int(14 * 96000 / 4096))


def test_fft_standin_within_1e5_of_exact_dft():
    rng = np.random.default_rng(3)
    for _ in range(2):
        buf = (rng.standard_normal(4096) * 0.2).astype(np.float32)
        got = O.fft_f32(buf).astype(np.float64)
        want = O.dft_exact(buf)
        peak = np.abs(want[0::2] + 1j * want[1::2]).max()
        err = np.abs((got[0::2] - want[0::2]) + 1j * (got[1::2] - want[1::2])).max()
        assert err <= 1e-5 * peak
        np_ref = np.fft.fft(buf[0::2].astype(np.float64) + 1j * buf[1::2].astype(np.float64))
        assert np.abs(np_ref - (want[0::2] + 1j * want[1::2])).max() < 1e-9 * peak


def test_fft_f64_roundtrip_and_numpy():
    rng = np.random.default_rng(4)
    a = rng.standard_normal(4096)
    f = O.fft_f64(a)
    ref = np.fft.fft(a[0::2] + 1j * a[1::2])
    assert np.abs((f[0::2] + 1j * f[1::2]) - ref).max() < 1e-10 * np.abs(ref).max()
    b = O.fft_f64(f, inverse=True, scale=True)
    assert np.abs(b - a).max() < 1e-12


# ------------------------------------------------------------------ fir.java
def test_fir_weights_and_filter_against_numpy():
    f = O.Fir()
    w = f.weights(500, 1500, 44100.0)
    rate = float(np.float32(44100.0))
    df1, df2 = 500 / rate, 1500 / rate
    want = np.empty(21)
    for n in range(21):
        if n == 10:
            v = 2 * (df2 - df1)
        else:
            v = np.sin(2 * np.pi * df2 * (n - 10)) / (np.pi * (n - 10)) - np.sin(2 * np.pi * df1 * (n - 10)) / (np.pi * (n - 10))
        want[n] = v * (0.54 - 0.46 * np.cos(2 * np.pi * n / 20))
    assert np.allclose(w, want, rtol=0, atol=1e-16)
    assert np.array_equal(w, G["fir_w_500_1500"])
    xs = G["fir_in"]
    got = f.filter_block(xs)
    hist = np.concatenate([np.zeros(20, np.int64), xs.astype(np.int64)])
    ref = np.empty(xs.size, np.int32)
    for t in range(xs.size):
        o = 0.0
        for i in range(21):  # newest first, same order as the ring walk
            o = o + float(hist[20 + t - i]) * w[i]
        ref[t] = int(o)  # Python int() truncates toward zero like Java (int)
    assert np.array_equal(got, ref)
    assert np.array_equal(got, G["fir_out"])


def test_fir_allpass_is_delay_of_10():
    f = O.Fir()
    f.weights(-2 ** 31, -2 ** 31)
    xs = np.arange(1, 60, dtype=np.int32) * 37 - 900
    got = f.filter_block(xs)
    assert np.array_equal(got[10:], xs[:-10])
    assert np.all(got[:10] == 0)


def test_fir_complex_gen_and_mod():
    g = O.fir_complex_gen(1000, 256)
    k = np.arange(256)
    w = (2 * np.pi * 1000 * k) / float(np.float32(44100.0))
    assert np.array_equal(g[:, 0], np.trunc(np.cos(w) * 4096).astype(np.int32))
    assert np.array_equal(g[:, 1], np.trunc(np.sin(w) * 4096).astype(np.int32))
    h = O.fir_complex_gen(500, 256)
    m = O.fir_complex_mod(g, h)
    a = g.astype(np.int64)
    b = h.astype(np.int64)
    assert np.array_equal(m[:, 0], (a[:, 0] * b[:, 0] - a[:, 1] * b[:, 1]).astype(np.int32))
    assert np.array_equal(m[:, 1], (a[:, 0] * b[:, 1] + a[:, 1] * b[:, 0]).astype(np.int32))
    # counter wraps at (int)rate
    g2 = O.fir_complex_gen(1000, 3, start=44099)
    assert np.array_equal(g2[1], g[0])


# ------------------------------------------------------------------ phase.java
def test_phase_reductions(golden_dir):
    _, buf = sine_buf(golden_dir)
    d = buf[:4096]
    assert O.phase_maxabs(d) == float(np.abs(d).max())
    assert O.phase_maxabs(np.zeros(8, np.float32)) == 0.0
    pix, ai, aq = O.phase_columns(d, 300)
    step = np.float32(600) / np.float32(4096)
    pos = np.float32(0)
    lp = 0
    si = sq = np.float32(0)
    cnt = 0
    rp, ri, rq = [], [], []
    for s in range(0, 4096, 2):
        si = np.float32(si + d[s])
        sq = np.float32(sq + d[s + 1])
        cnt += 1
        pos = np.float32(pos + step)
        p = int(pos)
        if p > lp:
            rp.append(p)
            ri.append(np.float32(si / np.float32(cnt)))
            rq.append(np.float32(sq / np.float32(cnt)))
            lp = p
            cnt = 0
            si = sq = np.float32(0)
    assert np.array_equal(pix, np.array(rp, np.int32))
    assert np.array_equal(ai, np.array(ri, np.float32))
    assert np.array_equal(aq, np.array(rq, np.float32))


# ------------------------------------------------------------------ FECDecoder.java
def test_fec_roundtrip_clean_and_with_errors():
    rng = np.random.default_rng(5)
    for trial in range(3):
        data = rng.integers(0, 256, 256, dtype=np.uint8)
        sym = O.fec_encode(data)
        assert set(np.unique(sym)) <= {0, 1}
        soft = np.where(sym == 1, 0xC0, 0x40).astype(np.uint8)
        rc, out = O.fec_decode(soft)
        assert rc == 0 and np.array_equal(out, data)
        for e in (1, 17, 150):
            s2 = soft.copy()
            pos = rng.choice(5200, e, replace=False)
            s2[pos] ^= 0x80
            rc, out = O.fec_decode(s2)
            assert rc == e and np.array_equal(out, data)


def test_fec_uncorrectable_returns_minus_one():
    rng = np.random.default_rng(6)
    data = rng.integers(0, 256, 256, dtype=np.uint8)
    soft = np.where(O.fec_encode(data) == 1, 0xC0, 0x40).astype(np.uint8)
    soft[rng.choice(5200, 1500, replace=False)] ^= 0x80
    rc, _ = O.fec_decode(soft)
    assert rc == -1
    rc, _ = O.fec_decode(rng.integers(0, 256, 5200, dtype=np.uint8))
    assert rc == -1


def test_fec_sync_vector_is_first_interleaver_column():
    sym = O.fec_encode(np.zeros(256, np.uint8))
    sv = O.bpsk_table(2)
    assert np.array_equal(np.where(sym[0::80] == 1, 1, -1), sv.astype(np.int64))


def test_fec_soft_values_other_than_c0_40():
    data = np.arange(256, dtype=np.uint8)
    sym = O.fec_encode(data)
    rng = np.random.default_rng(8)
    soft = np.where(sym == 1, 128 + rng.integers(1, 128, 5200), 127 - rng.integers(0, 128, 5200)).astype(np.uint8)
    rc, out = O.fec_decode(soft)
    assert rc == 0 and np.array_equal(out, data)


def test_fec_golden_vectors():
    assert np.array_equal(O.fec_encode(G["fec_payload"]), G["fec_symbols"])
    rc, dec = O.fec_decode(G["fec_soft_200err"])
    assert rc == int(G["fec_rc_200err"][0]) == 200
    assert np.array_equal(dec, G["fec_dec_200err"]) and np.array_equal(dec, G["fec_payload"])


# ------------------------------------------------------------------ FUNcubeBPSKDemod.java
def test_bpsk_sine4410_behavioural_kat(golden_dir):
    _, buf = sine_buf(golden_dir)
    d = O.Bpsk(do_fft=0)
    d.receive(buf[:4096])
    d.receive(buf[4096:])
    c = d.counters()
    assert (c["cntRaw"], c["cntDS"], c["cntBit"], c["cntFEC"]) == (4096, 409, 50, 0)
    bits = d.bits()
    assert list(bits[:4]) == [1, -1, 1, -1] and np.all(bits[4:] == 1)
    assert np.all(d.decoded() == 0)
    d = O.Bpsk(do_fft=1)
    d.receive(buf[:4096])
    assert d.counters()["centreBin"] == 211
    d.receive(buf[4096:])
    c = d.counters()
    assert c["cntBit"] == 51 and c["centreBin"] == 200 and c["cntFEC"] == 0


def test_bpsk_schedules_are_periodic_at_96k():
    """SURVEY 7 hard part 3: at 12 kHz / 96 kHz the tuner index is an exact 8-cycle, the bit clock
    is exactly periodic (dmBitPos==sample index mod 8)."""
    d = O.Bpsk()
    z = np.zeros(4096, np.float32)
    for _ in range(40):
        d.receive(z)
    ist = d.istate()
    n_ds = d.counters()["cntDS"]
    assert ist[3] == n_ds % 8  # dmBitPos
    assert ist[2] == (64 - n_ds) % 65  # dmPos
    assert ist[0] == (26 - d.counters()["cntRaw"]) % 27  # dsPos


def test_bpsk_dbpsk_end_to_end_decodes_payload():
    iq, pay, _ = O.make_dbpsk_stream(20020109, 3, 458752)
    d = O.Bpsk()
    d.receive_i16(iq)
    fr = d.fec_results()
    assert len(fr) == 1
    rc, bitidx, data = fr[0]
    assert rc >= 0 and np.array_equal(data, pay[0])
    assert np.array_equal(d.bits(), G["dbpsk_bits"])
    assert np.array_equal(np.array([rc], np.int32), G["dbpsk_fec_rc"])


def test_bpsk_frame_chunking_is_irrelevant_to_stream_state():
    """receive() on 2048-sample frames == the same samples in 512-sample frames (state carries)."""
    iq, _, _ = O.make_dbpsk_stream(1, 0, 65536, noise_sigma=800.0)
    a = O.Bpsk(blen=8192)
    a.receive_i16(iq)
    b = O.Bpsk(blen=2048)
    b.receive_i16(iq)
    assert np.array_equal(a.bits(), b.bits())
    assert np.array_equal(a.state(), b.state())


@pytest.mark.parametrize("rate,tuning", [(96000, 12000), (192000, 12000), (96000, 10000)])
def test_bpsk_fir_stages_against_numpy_restatement(rate, tuning):
    """Independent restatement of tuner + 27-tap/decimate + VCO + 65-tap in numpy float64 with the
    reference's summation orders; must be bit-identical to the C oracle."""
    rng = np.random.default_rng(9)
    n = 4096
    raw = rng.integers(-20000, 20000, 2 * n).astype(np.int16)
    buf = O.convert_i16(raw)
    d = O.Bpsk(rate=rate, blen=4 * n, tuning=tuning, trace=n)
    d.receive(buf)
    tr = d.trace()
    ds = d.trace_ds()
    sinT, cosT = O.bpsk_sincos()
    dsF = O.bpsk_table(0)
    dmF = O.bpsk_table(1)
    two_pi = 2.0 * np.pi
    inc = two_pi * float(tuning) / float(rate)
    ph = 0.0
    dsBuf = np.zeros((27, 2))
    dsPos, dsCnt = 26, 0
    dmBuf = np.zeros((65, 2))
    dmPos = 64
    vco = 0.0
    vinc = two_pi * 1200.0 / 9600.0
    outs, dss = [], []
    for t in range(n):
        i = float(buf[2 * t])
        q = float(buf[2 * t + 1])
        ph += inc
        if ph > two_pi:
            ph -= two_pi
        if ph > 0.0:
            k = int(ph * 256.0 / two_pi) % 256
            i, q = i * cosT[k], q * sinT[k]
        dsBuf[dsPos] = (i, q)
        dsCnt += 1
        if dsCnt >= rate // 9600:
            fi = fq = 0.0
            for m in range(27):
                s = (m + dsPos) % 27
                fi += dsBuf[s][0] * dsF[m]
                fq += dsBuf[s][1] * dsF[m]
            dsCnt = 0
            fi *= 0.9 * 32768.0
            fq *= 0.9 * 32768.0
            dss.append((fi, fq))
            vco += vinc
            if vco > two_pi:
                vco -= two_pi
            k = int(vco * 256.0 / two_pi) % 256
            dmBuf[dmPos] = (fi * cosT[k], fq * sinT[k])
            gi = gq = 0.0
            for m in range(65):
                gi += dmBuf[m][0] * dmF[65 - dmPos + m]
                gq += dmBuf[m][1] * dmF[65 - dmPos + m]
            dmPos = dmPos - 1 if dmPos > 0 else 64
            outs.append((gi, gq))
        dsPos = dsPos - 1 if dsPos > 0 else 26
    assert np.array_equal(ds, np.array(dss))
    assert np.array_equal(tr, np.array(outs))


def test_fft_oracle_standin_for_the_default_9600_frame():
    """non power-of-two frames go through the exact DFT; the Hz rule divides by 2n = 19200"""
    n = 9600
    t = np.arange(n)
    for k in (1234, 9000):
        x = 0.5 * np.exp(2j * np.pi * k * t / n)
        buf = np.empty(2 * n, np.float32)
        buf[0::2], buf[1::2] = x.real, x.imag
        psd = O.fft_receive(buf, 96000)
        assert int(np.argmax(psd[:n])) == k
        p = 2 * k
        want = int(p * 96000 / (2 * n)) if p < n else int((p - 2 * n) * 96000 / (2 * n))
        assert psd[n] == float(want)


def test_oracle_mixed_radix_f64_transform_against_numpy():
    """the oracle's definition of the FFT-acquire transform for frames that are not a power of two (the reference's
    default n = 9600 / 4800): Stockham radices 4,..,(2),3..,5.. -- an accurate DFT, and inverse(forward(x)) = x"""
    import ctypes as C
    L = O.lib()
    L.jo_fft_f64.restype = None
    L.jo_fft_f64.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.jo_fft_mixed_radices.argtypes = [C.c_int, C.c_void_p]
    rad = np.zeros(32, np.int32)
    assert L.jo_fft_mixed_radices(9600, rad.ctypes.data) == 7 and list(rad[:7]) == [4, 4, 4, 2, 3, 5, 5]
    assert L.jo_fft_mixed_radices(4800, rad.ctypes.data) == 6 and list(rad[:6]) == [4, 4, 4, 3, 5, 5]
    assert L.jo_fft_mixed_radices(7000, rad.ctypes.data) == 6 and list(rad[:6]) == [4, 2, 5, 5, 5, 7]  # round 4: radix 7
    assert L.jo_fft_mixed_radices(4410, rad.ctypes.data) == 6 and list(rad[:6]) == [2, 3, 3, 5, 7, 7]  # a 44.1 kHz sound card's frame
    # round 5: any other prime factor, ascending, through the r-point DFT as its definition
    assert L.jo_fft_mixed_radices(1100, rad.ctypes.data) == 4 and list(rad[:4]) == [4, 5, 5, 11]
    assert L.jo_fft_mixed_radices(1102, rad.ctypes.data) == 3 and list(rad[:3]) == [2, 19, 29]  # an 11.025 kHz card's frame
    assert L.jo_fft_mixed_radices(1103, rad.ctypes.data) == 1 and list(rad[:1]) == [1103]       # a prime frame: the DFT sum itself
    assert L.jo_fft_mixed_radices(2 * 11 * 11 * 13, rad.ctypes.data) == 4 and list(rad[:4]) == [2, 11, 11, 13]
    # round 6: frames above 9600 samples start with one radix-2 pass; the frames the any-frame passes serve (csrc/bpsk_acqg.hip)
    assert L.jo_fft_mixed_radices(17640, rad.ctypes.data) == 7 and list(rad[:7]) == [2, 4, 3, 3, 5, 7, 7]  # a 176.4 kHz card's frame
    assert L.jo_fft_mixed_radices(38400, rad.ctypes.data) == 8 and list(rad[:8]) == [2, 4, 4, 4, 4, 3, 5, 5]  # a 384 kHz card's
    assert L.jo_fft_mixed_radices(9614, rad.ctypes.data) == 4 and list(rad[:4]) == [2, 11, 19, 23]
    assert L.jo_fft_mixed_radices(11025, rad.ctypes.data) == 6 and list(rad[:6]) == [3, 3, 5, 5, 7, 7]
    rng = np.random.default_rng(12)
    for n in (9600, 4800, 2400, 60, 15, 6, 4410, 7000, 49, 7, 2646, 1102, 1100, 1103, 3146, 11, 800, 638, 17640, 38400, 9614, 11025, 512, 16384):
        x = rng.standard_normal(2 * n)
        a = x.copy()
        L.jo_fft_f64(a.ctypes.data, n, 0, 0)
        want = np.fft.fft(x[0::2] + 1j * x[1::2])
        assert np.abs((a[0::2] + 1j * a[1::2]) - want).max() <= 2e-15 * np.abs(want).max() * np.log2(n)
        L.jo_fft_f64(a.ctypes.data, n, 1, 1)
        assert np.abs(a - x).max() < 1e-13
    # a single tone lands in its bin exactly like the power-of-two network's
    n = 9600
    t = np.arange(n)
    x = np.zeros(2 * n)
    x[0::2], x[1::2] = np.cos(2 * np.pi * 1320 * t / n), np.sin(2 * np.pi * 1320 * t / n)
    L.jo_fft_f64(x.ctypes.data, n, 0, 0)
    mag = np.hypot(x[0::2], x[1::2])
    assert int(np.argmax(mag)) == 1320 and abs(mag[1320] - n) < 1e-8 and np.delete(mag, 1320).max() < 1e-8
