"""GPU parity: sample conversion (JavaAudio.java:276-293) and the fft.java path, through the C ABI."""
import os

import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.npz"))

# Tolerance of BASELINE.json north_star: "FFT/FIR floats within 1e-5 relative".  JTransforms' own float
# rounding is unknowable (source absent), so spectra are compared with the EXACT float64 DFT, error
# normalised by the frame's peak magnitude (hard part 5).
FFT_RTOL = 1e-5


def lin(db):
    return 10.0 ** (np.asarray(db, np.float64) / 20.0)


def check_psd(got, ref, n):
    """psd[n+2] vs oracle: amplitudes within FFT_RTOL of the frame peak; dB within 1e-3 where the bin is
    within 60 dB of the peak; argmax bin identical or an exact-tie mirror (hard part 6)."""
    a, b = lin(got[:n]), lin(ref[:n])
    assert np.abs(a - b).max() <= FFT_RTOL * b.max()
    strong = ref[:n] > ref[:n].max() - 60.0
    assert np.abs(got[:n][strong] - ref[:n][strong]).max() < 1e-3
    if got[n] != ref[n]:
        kg, kr = int(np.argmax(got[:n])), int(np.argmax(ref[:n]))
        assert abs(ref[kg] - ref[kr]) < 1e-3, "argmax moved to a bin that is not a numerical tie"
    assert abs(got[n + 1] - ref[n + 1]) < 1e-3


def prime_factors(n):
    out, q = [], 2
    while q * q <= n:
        while n % q == 0:
            out.append(q)
            n //= q
        q += 1
    if n > 1:
        out.append(n)
    return out


def expected_kernel(n):
    """which kernel serves a frame that is neither a power of two nor 4800 / 9600 / 19200 (fft_rt.hip rt_plan): k_fft_rt for a
    composite n up to 9800 whose prime factors above 7 are small beside it (8 x their sum <= n), k_dft_any for the rest"""
    pf = prime_factors(n)
    big = sum(q for q in pf if q > 7)
    if len(pf) < 2 or n > 9800 or n < 6 or 8 * big > n:
        return "k_dft_any"
    return "k_fft_rt"


def test_convert_all_65536_values_bit_exact():
    raw = np.arange(-32768, 32768, dtype=np.int32).astype(np.int16)
    iq = np.stack([raw, raw[::-1]], axis=1).reshape(-1)
    assert np.array_equal(J.convert_i16(iq), O.convert_i16(iq))


def test_convert_dc_correction_and_mono():
    rng = np.random.default_rng(1)
    iq = rng.integers(-32768, 32768, 4096).astype(np.int16)
    for ic, qc in ((5, -7), (32767, -32768), (70000, -70000)):
        assert np.array_equal(J.convert_i16(iq, ic=ic, qc=qc), O.convert_i16(iq, ic=ic, qc=qc))
    assert np.array_equal(J.convert_i16(iq, chns=1, ic=3), O.convert_i16(iq, chns=1, ic=3))


def test_fft_sine4410_fixture(golden_dir):
    raw = np.fromfile(os.path.join(golden_dir, "sine4410.raw"), dtype="<i2")
    f = J.Fft(2048, 96000)
    f44 = J.Fft(2048, 44100)
    for k in range(2):
        fr = raw[k * 4096:(k + 1) * 4096]
        psd = f.receive_raw(fr)
        check_psd(psd, G["sine_psd96k"][k], 2048)
        assert int(np.argmax(psd[:2048])) in (205, 1843)
        assert abs(psd[2048]) == 9609.0
        assert abs(f44.receive_raw(fr)[2048]) == 4414.0
        # float form of the same frame (IAudioHandler.receive)
        psd_f = f.receive(O.convert_i16(fr))
        assert np.array_equal(psd_f, psd)


def test_fft_zero_frame_edge_case():
    f = J.Fft(2048, 96000)
    psd = f.receive(np.zeros(4096, np.float32))
    assert np.all(np.isneginf(psd[:2048]))
    assert psd[2048] == -23.0 and psd[2049] == -np.finfo(np.float32).max


def test_fft_hz_rule_and_first_maximum():
    n = 2048
    f = J.Fft(n, 96000)
    for k in (0, 1, 300, 1023, 1024, 1500, 2047):
        t = np.arange(n)
        x = 0.5 * np.exp(2j * np.pi * k * t / n)
        buf = np.empty(2 * n, np.float32)
        buf[0::2], buf[1::2] = x.real, x.imag
        psd = f.receive(buf)
        ref = O.fft_receive(buf, 96000)
        assert int(np.argmax(psd[:n])) == k
        assert psd[n] == ref[n]
        assert abs(psd[n + 1] - ref[n + 1]) < 1e-3


def test_fft_int_overflow_of_hz_rule_wraps_like_java():
    # p*rate overflows int32 for n=8192 at rate 384000: p=2k up to 16382 -> 6.3e9
    n = 8192
    f = J.Fft(n, 384000)
    k = 6000
    t = np.arange(n)
    x = 0.5 * np.exp(2j * np.pi * k * t / n)
    buf = np.empty(2 * n, np.float32)
    buf[0::2], buf[1::2] = x.real, x.imag
    assert f.receive(buf)[n] == O.fft_receive(buf, 384000)[n]


@pytest.mark.parametrize("n", [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_fft_spectrum_within_1e5_of_exact_dft(n):
    rng = np.random.default_rng(n)
    frames = 3
    bufs = (rng.standard_normal((frames, 2 * n)) * 0.25).astype(np.float32)
    bufs[1] = 0
    bufs[1, 0::2] = 0.4 * np.cos(2 * np.pi * 7.25 * np.arange(n) / n)  # non-bin-centred real tone
    f = J.Fft(n, 96000)
    spec = f.spectrum(bufs).astype(np.float64)
    for k in range(frames):
        if n <= 2048:
            want = O.dft_exact(bufs[k])
            want = want[0::2] + 1j * want[1::2]
        else:
            want = np.fft.fft(bufs[k, 0::2].astype(np.float64) + 1j * bufs[k, 1::2].astype(np.float64))
        got = spec[k, 0::2] + 1j * spec[k, 1::2]
        assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max()


@pytest.mark.parametrize("n,nframes", [(2048, 1), (2048, 5), (2048, 64), (1024, 7), (256, 33), (64, 100), (4096, 3)])
def test_fft_batch_matches_oracle_ragged_counts(n, nframes):
    ct, _ = O.synth_tables(8000)
    raw = O.synth_tones(3, nframes, n, ct, 300, O.mix64(20020107 + n))
    f = J.Fft(n, 96000)
    psd = f.batch_host_i16(raw)
    buf = O.convert_i16(raw)
    for k in range(nframes):
        check_psd(psd[k], O.fft_receive(buf[k * 2 * n:(k + 1) * 2 * n], 96000), n)


def test_fft_golden_tones():
    f = J.Fft(2048, 96000)
    psd = f.batch_host_i16(G["tones_iq"])
    for k in range(4):
        check_psd(psd[k], G["tones_psd"][k], 2048)


@pytest.mark.parametrize("n,rate", [(9600, 96000), (4800, 48000), (19200, 192000)])
def test_fft_default_non_power_of_two_frames(n, rate):
    """java-sdr's default buffer is rate*size/10 bytes -> n = 9600 @96 kHz (JavaAudio.java:58-59): mixed radix
    2^k.3.5^2.  Spectrum against the exact DFT (1e-5 of the peak), PSD / argmax / Hz against the oracle's rule.
    n = 19200 (192 kHz, FUNcube Dongle Pro+) runs as two 9600-point transforms of the even / odd samples + a
    radix-2 combine pass; the oracle's exact DFT is O(n^2), so fewer frames there."""
    nchk = 3 if n < 19200 else 2
    rng = np.random.default_rng(n)
    t = np.arange(n)
    bufs = np.zeros((3, 2 * n), np.float32)
    x = 0.3 * np.exp(2j * np.pi * 1234 * t / n) + 0.05 * np.exp(-2j * np.pi * 777.5 * t / n)
    bufs[0, 0::2], bufs[0, 1::2] = x.real, x.imag
    bufs[1] = (rng.standard_normal(2 * n) * 0.2).astype(np.float32)
    bufs[2, 0::2] = 0.4 * np.cos(2 * np.pi * 1001.25 * t / n)
    f = J.Fft(n, rate)
    spec = f.spectrum(bufs).astype(np.float64)
    for k in range(3):
        want = np.fft.fft(bufs[k, 0::2].astype(np.float64) + 1j * bufs[k, 1::2].astype(np.float64))
        got = spec[k, 0::2] + 1j * spec[k, 1::2]
        assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max()
    for k in range(nchk):
        check_psd(f.receive(bufs[k]), O.fft_receive(bufs[k], rate), n)
    # raw (IRawHandler) form with DC correction, batched, ragged count
    raw = rng.integers(-20000, 20000, (5, 2 * n)).astype(np.int16)
    psd = f.batch_host_i16(raw, ic=11, qc=-7)
    for k in range(5 if n < 19200 else 1):
        check_psd(psd[k], O.fft_receive(O.convert_i16(raw[k], ic=11, qc=-7), rate), n)
    # every frame of the batch against numpy's FFT of the converted samples (independent of the oracle)
    for k in range(5):
        x = O.convert_i16(raw[k], ic=11, qc=-7).astype(np.float64)
        pw = np.abs(np.fft.fft(x[0::2] + 1j * x[1::2])) ** 2 * (2.0 / n) ** 2
        assert np.abs(lin(psd[k][:n]) - np.sqrt(pw)).max() <= FFT_RTOL * np.sqrt(pw).max()
    z = f.receive(np.zeros(2 * n, np.float32))
    assert np.all(np.isneginf(z[:n])) and z[n] == float(int(-1 * rate / (2 * n))) and z[n + 1] == -np.finfo(np.float32).max


def test_fft_rejects_unsupported_sizes():
    for n in (38400, 20001, 1, 0, -5):
        with pytest.raises(J.JsdrError):
            J.Fft(n, 96000)


@pytest.mark.parametrize("n,rate", [(4410, 44100), (2205, 22050), (3200, 32000), (1102, 11025), (8820, 88200), (17640, 176400),
                                    (9601, 96010), (16384, 96000), (100, 1000), (37, 370),
                                    # round 5: 2^a 3^b 5^c 7^d frames leave the O(n^2) kernel (run-time radix plan, fft_rt.hip)
                                    (800, 8000), (1600, 16000), (2400, 24000), (1200, 12000), (9800, 98000), (12, 120), (343, 3430), (630, 6300),
                                    # ... and every other composite frame: primes above 7 as passes of that radix (11.025 kHz: 1102 = 2.19.29)
                                    (143, 1430), (1100, 11000), (3146, 31460), (551, 5510), (9782, 97820)])
def test_fft_any_frame_size(n, rate):
    """the reference's audio-rate is a free integer and its frame rate / 10 (JavaAudio.java:49,59; JTransforms takes any n,
    fft.java:67,194): 44.1 kHz -> 4410 = 2.3^2.5.7^2 (the reference's own sine4410.wav), 11.025 kHz -> 1102 = 2.19.29,
    primes ...  Frames without a Stockham kernel take the DFT itself (fft_any.hip).  Same bar as the other sizes: spectrum
    within 1e-5 of the peak of the exact DFT, PSD / first maximum / Hz rule against the oracle, int16 form with DC
    correction, ragged batch, the all-zero frame."""
    rng = np.random.default_rng(n)
    t = np.arange(n)
    bufs = np.zeros((3, 2 * n), np.float32)
    x = 0.3 * np.exp(2j * np.pi * (n // 7) * t / n) + 0.05 * np.exp(-2j * np.pi * (n / 9.0 + 0.5) * t / n)
    bufs[0, 0::2], bufs[0, 1::2] = x.real, x.imag
    bufs[1] = (rng.standard_normal(2 * n) * 0.2).astype(np.float32)
    bufs[2, 0::2] = 0.4 * np.cos(2 * np.pi * (n / 5.0 + 0.25) * t / n)
    f = J.Fft(n, rate)
    # k_fft_rt: any composite n up to 9800 (primes above 7 through a pass that is the DFT's definition: 1102 = 2.19.29); k_dft_any: primes
    # and frames above 9800 samples
    assert f.kernel_name() == expected_kernel(n), f.kernel_name()
    spec = f.spectrum(bufs).astype(np.float64)
    for k in range(3):
        want = np.fft.fft(bufs[k, 0::2].astype(np.float64) + 1j * bufs[k, 1::2].astype(np.float64))
        got = spec[k, 0::2] + 1j * spec[k, 1::2]
        assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max()
    nchk = 3 if n <= 4410 else 1  # (the oracle's exact DFT is O(n^2) in long double)
    for k in range(nchk):
        check_psd(f.receive(bufs[k]), O.fft_receive(bufs[k], rate), n)
    raw = rng.integers(-20000, 20000, (5, 2 * n)).astype(np.int16)
    psd = f.batch_host_i16(raw, ic=11, qc=-7)
    check_psd(psd[4], O.fft_receive(O.convert_i16(raw[4], ic=11, qc=-7), rate), n)
    for k in range(5):
        xx = O.convert_i16(raw[k], ic=11, qc=-7).astype(np.float64)
        pw = np.abs(np.fft.fft(xx[0::2] + 1j * xx[1::2])) ** 2 * (2.0 / n) ** 2
        assert np.abs(lin(psd[k][:n]) - np.sqrt(pw)).max() <= FFT_RTOL * np.sqrt(pw).max()
        assert np.array_equal(f.receive_raw(raw[k], ic=11, qc=-7), psd[k])  # one frame spread over the chip == the batch form
    z = f.receive(np.zeros(2 * n, np.float32))
    assert np.all(np.isneginf(z[:n])) and z[n] == float(int(-1 * rate / (2 * n))) and z[n + 1] == -np.finfo(np.float32).max


def test_fft_sine4410_wav_at_its_own_rate(golden_dir):
    """the reference's sine4410.wav is a 44.1 kHz recording: its own frame is 4410 samples (audio-buflen = rate*size/10)"""
    import wave
    with wave.open(os.path.join(golden_dir, "sine4410.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (2, 2, 44100)
        raw = np.frombuffer(w.readframes(2 * 4410), dtype="<i2")
    n = 4410
    f = J.Fft(n, 44100)
    for k in range(2):
        psd = f.receive_raw(raw[2 * n * k:2 * n * (k + 1)])
        check_psd(psd, O.fft_receive(O.convert_i16(raw[2 * n * k:2 * n * (k + 1)]), 44100), n)
        assert abs(abs(psd[n]) - 4410.0) <= 44100 / n + 1  # the tone the file is named after


def test_fft_linearity_property_at_full_batch_size():
    """size-independent property at BASELINE's batch shape: psd of frame k does not depend on its
    neighbours -> a 4096-frame batch equals per-frame calls on a sample of frames."""
    n, nframes = 2048, 4096
    ct, _ = O.synth_tables(6000)
    d_ct = J.DeviceBuffer.from_host(ct)
    d_raw = J.DeviceBuffer(nframes * n * 4)
    J.synth_tones(d_raw, 0, nframes, n, d_ct, 250, O.mix64(99))
    d_psd = J.DeviceBuffer(nframes * (n + 2) * 4)
    f = J.Fft(n, 96000)
    f.batch_i16(d_raw, nframes, d_psd)
    psd = d_psd.to_host(np.float32).reshape(nframes, n + 2)
    raw = d_raw.to_host(np.int16).reshape(nframes, 2 * n)
    for k in (0, 1, 777, 4095):
        assert np.array_equal(f.receive_raw(raw[k]), psd[k])
        check_psd(psd[k], O.fft_receive(O.convert_i16(raw[k]), 96000), n)


def test_fft_run_time_plan_at_random_smooth_sizes():
    """k_fft_rt takes ANY n = 2^a 3^b 5^c 7^d up to 9800 (audio-rate is a free integer, JavaAudio.java:49,58-59): forty sizes drawn
    from all of them -- every radix mix, plans of two to eight passes, the composite radices from 4000 samples -- spectrum within
    1e-5 of the peak of the exact DFT, batch of three frames each"""
    smooth = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(14) for b in range(9) for c in range(6) for d in range(5)
                     if 6 <= 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 9800})
    rng = np.random.default_rng(2024)
    picks = [n for n in rng.choice(smooth, 48, replace=False) if (n & (n - 1)) != 0 and n not in (9600, 4800)][:40]
    picks += [9800, 9604, 8750, 4000, 4116, 3969]  # the largest, 2^2 7^4, 2 5^4 7, the merge threshold, 2^2 3 7^3, 3^4 7^2
    for n in picks:
        f = J.Fft(int(n), 10 * int(n))
        if f.kernel_name() != "k_fft_rt":
            assert int(n) in (7, 9, 14, 15, 21, 25, 35, 49, 6, 10) or n < 12, (n, f.kernel_name())  # one-pass plans go to k_dft_any
            continue
        bufs = (rng.standard_normal((3, 2 * int(n))) * 0.25).astype(np.float32)
        bufs[0, 0::2] += np.float32(0.5) * np.cos(2 * np.pi * (int(n) // 5) * np.arange(int(n)) / int(n)).astype(np.float32)
        spec = f.spectrum(bufs).astype(np.float64)
        for k in range(3):
            want = np.fft.fft(bufs[k, 0::2].astype(np.float64) + 1j * bufs[k, 1::2].astype(np.float64))
            got = spec[k, 0::2] + 1j * spec[k, 1::2]
            assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max(), n


def test_fft_run_time_plan_at_random_sizes_with_large_prime_factors():
    """... and any OTHER composite frame up to 9800 samples: a prime factor above 7 becomes a pass of that radix by the DFT's
    definition (k_fft_rt's rt_pass_generic, summed in double) while such factors are small beside n; the rest stays with
    k_dft_any.  Sizes drawn from all composites that are not 7-smooth, plus the corners: a huge prime radix (2 x 4099), two
    large primes (97 x 101), a prime squared (97^2), a prime cubed (19^3), the largest such n, the smallest."""
    rng = np.random.default_rng(77)
    cand = [n for n in range(22, 9801) if len(prime_factors(n)) >= 2 and max(prime_factors(n)) > 7]
    picks = [int(n) for n in rng.choice(cand, 30, replace=False)] + [2 * 4099, 97 * 101, 97 * 97, 19 ** 3, 9799, 9798, 22, 26, 4 * 2447, 8 * 11, 16 * 11]
    seen = set()
    for n in picks:
        f = J.Fft(n, 10 * n)
        assert f.kernel_name() == expected_kernel(n), (n, prime_factors(n), f.kernel_name())
        seen.add(f.kernel_name())
        bufs = (rng.standard_normal((3, 2 * n)) * 0.25).astype(np.float32)
        bufs[0, 0::2] += np.float32(0.5) * np.cos(2 * np.pi * (n // 5) * np.arange(n) / n).astype(np.float32)
        spec = f.spectrum(bufs).astype(np.float64)
        psd = np.stack([f.receive(bufs[k]) for k in range(3)])
        for k in range(3):
            want = np.fft.fft(bufs[k, 0::2].astype(np.float64) + 1j * bufs[k, 1::2].astype(np.float64))
            got = spec[k, 0::2] + 1j * spec[k, 1::2]
            assert np.abs(got - want).max() <= FFT_RTOL * np.abs(want).max(), (n, prime_factors(n))
            check_psd(psd[k], O.fft_receive(bufs[k], 10 * n), n)
    assert seen == {"k_fft_rt", "k_dft_any"}
