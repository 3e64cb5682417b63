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
    return o.fft_probe()


def test_decisions_against_an_ffts_own_rounding_error():
    report = []
    total = changed1 = changed16 = 0
    worst1 = worst2 = np.inf
    for name, rate, frame, do_up, iq in corpus():
        base = run(rate, frame, do_up, iq)
        nfr = len(base)
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
                p = run(rate, frame, do_up, iq, perturb=scale, seed=seed)
                worst = max(worst, int(np.count_nonzero(p[:, 5] != base[:, 5])))
            flips[scale] = worst
        total += nfr
        changed1 += flips[1.0]
        changed16 += flips[16.0]
        report.append("%-34s frames %4d  argmax margin min %.3g (x error)  rule margin min %.3g  centreBin changed: %d / %d (x1 / x16)"
                      % (name, nfr, m1.min(initial=np.inf), m2.min(initial=np.inf), flips[1.0], flips[16.0]))
    text = "\n".join(report) + "\ntotal frames %d; centreBin decisions changed by a perturbation of 1 x / 16 x the bound: %d / %d; " \
        "smallest margins: argmax %.3g, rule %.3g (in units of the error an FFT can put into a boxcar sum)" % (
            total, changed1, changed16, worst1, worst2)
    print(text)
    with open(os.path.join(os.path.dirname(GOLD), "..", "profiles", "r04_fft_decision_robustness.txt"), "w") as f:
        f.write("tests/test_fft_decision_robustness.py (CPU, the C oracle)\n" + text + "\n")
    # what the parity claim rests on: no decision of the corpus sits within an FFT's rounding error of its alternative
    assert changed1 == 0 and changed16 == 0, text
    assert worst1 > 100.0 and worst2 > 100.0, text
