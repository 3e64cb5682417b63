"""How robust are FFT-acquire mode's data-dependent decisions to the last bits of the FFT?  

`doBufferFFT` (FUNcubeBPSKDemod.java:406-464) crosses JTransforms' `DoubleFFT_1D`, whose source is absent: the oracle's
own double FFT IS the definition the kernels are held to, and agreement with JTransforms' last bit is unpinned.  What that
costs depends on how often a decision taken on the spectrum sits within an FFT's rounding error of its alternative:
  (1) the first maximum of the 100-bin boxcar sums over the searched quarter band (:433-443) -- the margin is the gap to the
      runner-up position's sum;
  (2) the rule `maxBin > avePeakPower/4*5` (:447) -- the margin is the distance of maxBin to that threshold.
Both are measured here, on the CPU, in units of the error an FFT can put into a boxcar sum: a floating-point FFT has the
forward error |X'_k - X_k| <= c u log2(n) ||x||_2 per bin (Higham, ASNA, thm 24.2; c a small constant), so a sum of
100 magnitudes moves by at most 100 x that.  And the decisions are re-taken with the spectrum PERTURBED by that bound
(x 1 and x 16, several seeds; the perturbed centre bin feeds the averages of the following frames, as a different FFT's
would): the number of frames whose `centreBin` differs from the unperturbed run is counted.

Round 5 takes the same question one stage further, to the BITS: the inverse transform's output
(:459-463) feeds RxDownSample, both FIRs and RxDemodulate's three data-dependent decisions -- `energy2 > 100`, `di < 0`
(:544-545) and the 8-way dmNewPeak argmax (:586-592).  The oracle carries a bound on what two correct double FFTs can differ
by in one real sample (the inverse's own rounding plus the forward error in the 204 gathered bins) through the filters' tap
sums and the energy IIRs, and records each decision's smallest margin in units of the error that bound allows in the
compared quantity; and the inverse output is perturbed by its bound as well (x 1, x 16), the whole bit stream, the threshold
outcomes and the peak sequence compared with the unperturbed run.

Corpus: the six streams of the FFT-mode fixtures' recipe (clean, noisy, noise-only; both band halves) at 2048-sample frames,
DBPSK + noise at the application's default frames 9600 and 19200 (192 kHz), and the reference's sine4410.raw."""
import os

import numpy as np

import oracle_lib as O

U = 2.0 ** -53
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def corpus():
    out = []
    n = 2048 * 160
    for do_up, carrier in ((0, 13200.0), (0, 6000.0), (1, 30000.0)):
        for s, sigma in ((0, 600.0), (1, 2400.0)):
            out.append(("2048 up=%d carrier=%d sigma=%d" % (do_up, carrier, sigma), 96000, 2048, do_up,
                        O.make_dbpsk_stream(41, s, n, carrier_hz=carrier, noise_sigma=sigma)[0]))
    rng = np.random.default_rng(5)
    out.append(("2048 noise only", 96000, 2048, 0, rng.integers(-12000, 12000, 2 * n).astype(np.int16)))
    out.append(("9600 default frame", 96000, 9600, 0, O.make_dbpsk_stream(43, 2, 9600 * 60, noise_sigma=900.0)[0]))
    out.append(("19200 FCD Pro+ frame", 192000, 19200, 0, O.make_dbpsk_stream(44, 3, 19200 * 30, rate=192000, noise_sigma=900.0)[0]))
    out.append(("4410 (44.1 kHz sound card)", 44100, 4410, 0,
                O.make_dbpsk_stream(45, 4, 4410 * 60, rate=44100, carrier_hz=6000.0, noise_sigma=900.0)[0]))
    raw = np.fromfile(os.path.join(GOLD, "sine4410.raw"), dtype="<i2")
    out.append(("sine4410.raw", 96000, 2048, 0, raw[:(raw.size // 4096) * 4096]))
    return out


def run(rate, frame, do_up, iq, perturb=0.0, seed=0):
    o = O.Bpsk(rate=rate, blen=4 * frame, do_fft=1, do_up=do_up)
    o.fft_probe_enable(iq.size // (2 * frame) + 1)
    if perturb:
        o.fft_perturb(perturb, seed)
    o.receive_i16(iq)
    return o.fft_probe(), o.bits().copy(), o.decision_margins()


def test_decisions_against_an_ffts_own_rounding_error():
    report = []
    total = changed1 = changed16 = 0
    worst1 = worst2 = np.inf
    bits_total = 0
    bitflips = {1.0: 0, 16.0: 0}
    seqflips = {1.0: 0, 16.0: 0}
    worst_dec = {"di": np.inf, "energy2": np.inf, "argmax": np.inf}
    ndec = {"di": 0, "energy2": 0, "argmax": 0}
    for name, rate, frame, do_up, iq in corpus():
        base, base_bits, base_m = run(rate, frame, do_up, iq)
        nfr = len(base)
        bits_total += len(base_bits)
        for k in worst_dec:
            if base_m["n_" + k] > 0:
                worst_dec[k] = min(worst_dec[k], base_m[k])
                ndec[k] += base_m["n_" + k]
        # the error an FFT can put into a boxcar sum of 100 magnitudes (c = 1; |X| moves by at most |dX| <= sqrt(2) x the
        # per-component figure)
        esum = 100.0 * np.sqrt(2.0) * U * np.log2(base[:, 7]) * base[:, 6]
        live = (base[:, 0] >= 0) & (esum > 0)          # frames in which a maximum was found and the input is not all zero
        m1 = (base[live, 1] - base[live, 2]) / esum[live]       # argmax margin, in units of that error
        m2 = np.abs(base[live, 1] - base[live, 3]) / esum[live]  # rule margin
        worst1, worst2 = min(worst1, m1.min(initial=np.inf)), min(worst2, m2.min(initial=np.inf))
        flips = {}
        for scale in (1.0, 16.0):
            worst = 0
            for seed in (1, 2, 3):
                p, pbits, pm = run(rate, frame, do_up, iq, perturb=scale, seed=seed)
                worst = max(worst, int(np.count_nonzero(p[:, 5] != base[:, 5])))
                nb = min(len(pbits), len(base_bits))
                bitflips[scale] = max(bitflips[scale], int(np.count_nonzero(pbits[:nb] != base_bits[:nb])) + abs(len(pbits) - len(base_bits)))
                seqflips[scale] += int(pm["hash_energy2"] != base_m["hash_energy2"]) + int(pm["hash_peak"] != base_m["hash_peak"])
            flips[scale] = worst
        total += nfr
        changed1 += flips[1.0]
        changed16 += flips[16.0]
        report.append("%-34s frames %4d  argmax margin min %.3g (x error)  rule margin min %.3g  centreBin changed: %d / %d (x1 / x16)"
                      "  | bits %5d  margins di %.3g  energy2 %.3g  peak argmax %.3g"
                      % (name, nfr, m1.min(initial=np.inf), m2.min(initial=np.inf), flips[1.0], flips[16.0], len(base_bits),
                         base_m["di"] if base_m["n_di"] else np.inf, base_m["energy2"] if base_m["n_energy2"] else np.inf,
                         base_m["argmax"] if base_m["n_argmax"] else np.inf))
    text = "\n".join(report) + "\ntotal frames %d; centreBin decisions changed by a perturbation of 1 x / 16 x the bound: %d / %d; " \
        "smallest margins: argmax %.3g, rule %.3g (in units of the error an FFT can put into a boxcar sum)" % (
            total, changed1, changed16, worst1, worst2)
    text += ("\nRxDemodulate's decisions behind the inverse transform (:544-545, :586-592): %d slicer decisions (di < 0), %d threshold "
             "decisions (energy2 > 100), %d peak argmaxes; smallest margins in units of the error two correct double FFTs allow in the "
             "compared quantity: di %.3g, energy2 %.3g, argmax %.3g; with the spectrum AND the inverse's output perturbed by 1 x / 16 x "
             "their bounds: bits that differ (worst seed, of %d) %d / %d, threshold-outcome or peak sequences that differ (of %d runs each) %d / %d"
             % (ndec["di"], ndec["energy2"], ndec["argmax"], worst_dec["di"], worst_dec["energy2"], worst_dec["argmax"], bits_total,
                bitflips[1.0], bitflips[16.0], 3 * 2 * len(report), seqflips[1.0], seqflips[16.0]))
    print(text)
    with open(os.path.join(os.path.dirname(GOLD), "..", "profiles", "r05_fft_decision_robustness.txt"), "w") as f:
        f.write("tests/test_fft_decision_robustness.py (CPU, the C oracle)\n" + text + "\n")
    # what the parity claim rests on: no decision of the corpus sits within an FFT's rounding error of its alternative
    assert changed1 == 0 and changed16 == 0, text
    assert worst1 > 100.0 and worst2 > 100.0, text
    assert bitflips[1.0] == 0 and bitflips[16.0] == 0 and seqflips[1.0] == 0 and seqflips[16.0] == 0, text
