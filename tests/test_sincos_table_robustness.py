"""What can the one ulp that Java's Math.sin / Math.cos are allowed change?  (SURVEY.md 7 hard part 7; CPU only.)

The demodulator's 256-entry sin / cos tables are `Math.sin/cos(n*2.0*Math.PI/256)` (FUNcubeBPSKDemod.java:159-162).  Those functions
are specified to within 1 ulp, not correctly rounded, and no JVM exists here to read the tables off; oracle and product ship the
CORRECTLY ROUNDED values (oracle/o_bpsk.c jo_bpsk_sincos; tests/test_reference_fixtures.py pins them to an exact-rational
evaluation).  So the real tables may differ from ours by one ulp in any entry that is not exact.  The tables feed the tuner
(:384-390) and the VCO (:511-516), i.e. EVERY mode including the headline's, through products that are then filtered and
compared: `energy2 > 100`, `di < 0` (:544-545) and the 8-way dmNewPeak argmax (:586-592).

Measured here: every entry that is not exactly 0 or +-1 moved by one ulp -- all up, all down, and random signs (three seeds); both
tables (the entries near zero included: cos(64 * 2 pi / 256) is 6.1e-17, not 0, because the argument is a double; its ulp is
1.2e-32).  Over the FFT-mode corpus of test_fft_decision_robustness.py in both modes, plus the tune-mode fixture streams:
  * bits, FECDecode return codes and bytes, counters: compared with the unperturbed run;
  * every decision's margin in units of what the perturbation DID to the compared quantity (the largest change of that
    decision's di / energy2 / argmax gap over the five perturbed runs).
The FFT decisions in front (boxcar argmax, centre-bin rule) do not see the tables at all."""
import os
import sys

import numpy as np

import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from fixture_cases import STREAMS, stream_input  # noqa: E402
from test_fft_decision_robustness import corpus  # noqa: E402


def perturbed_tables(kind, seed=0):
    s, c = O.bpsk_sincos()
    rng = np.random.default_rng(seed)

    def move(t):
        t = t.copy()
        exact = (t == 0.0) | (np.abs(t) == 1.0)
        if kind == "up":
            d = np.ones(t.size)
        elif kind == "down":
            d = -np.ones(t.size)
        else:
            d = rng.choice([-1.0, 1.0], t.size)
        out = np.where(d > 0, np.nextafter(t, np.inf), np.nextafter(t, -np.inf))
        return np.where(exact, t, out), int(np.count_nonzero(~exact))

    (s2, ns), (c2, nc) = move(s), move(c)
    return s2, c2, ns + nc


PATTERNS = [("up", 0), ("down", 0), ("random", 1), ("random", 2), ("random", 3)]


def run(rate, frame, do_fft, do_up, iq, tuning=12000, ic=0, qc=0, tables=None):
    o = O.Bpsk(rate=rate, blen=4 * frame, tuning=tuning, do_fft=do_fft, do_up=do_up)
    o.declog_enable(iq.size // (2 * max(1, rate // 9600)) + 64)
    if tables is not None:
        o.set_sincos(*tables)
    o.receive_i16(iq, ic, qc)
    c = o.counters()
    return dict(bits=o.bits().copy(), fec=[(r[0], r[1], r[2].tobytes()) for r in o.fec_results()],
                counters=[c[k] for k in ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits")],
                det=o.declog(0), peak=o.declog(1), centre=c.get("centreBin", 0))


def cases():
    out = []
    for name, rate, frame, do_up, iq in corpus():
        out.append(("FFT mode  " + name, rate, frame, 1, do_up, iq, 12000, 0, 0))
        if frame == 2048 and rate == 96000:  # the same streams through the tuner (tune mode takes 2048-sample frames)
            out.append(("tune mode " + name, rate, frame, 0, 0, iq, 12000, 0, 0))
    for name, p in STREAMS.items():  # the tune-mode fixture streams (incl. DC correction, other tunings / rates)
        raw = stream_input(name)
        nfr = raw.size // 4096
        out.append(("fixture   " + name, p["rate"], 2048, 0, 0, raw[:nfr * 4096], p["tuning"], p["ic"], p["qc"]))
    return out


def test_one_ulp_in_every_table_entry_changes_no_decision():
    report = []
    worst = {"di": np.inf, "energy2": np.inf, "argmax": np.inf}
    ndec = {"di": 0, "energy2": 0, "argmax": 0}
    changed = {"bits": 0, "fec": 0, "counters": 0, "threshold": 0, "peak": 0, "centre": 0}
    nbits = nfec = 0
    moved = 0
    for name, rate, frame, do_fft, do_up, iq, tuning, ic, qc in cases():
        base = run(rate, frame, do_fft, do_up, iq, tuning, ic, qc)
        nbits += len(base["bits"])
        nfec += len(base["fec"])
        ddi = np.zeros(len(base["det"]))
        de2 = np.zeros(len(base["det"]))
        dgap = np.zeros(len(base["peak"]))
        for kind, seed in PATTERNS:
            s2, c2, moved = perturbed_tables(kind, seed)
            p = run(rate, frame, do_fft, do_up, iq, tuning, ic, qc, tables=(s2, c2))
            changed["bits"] += int(len(p["bits"]) != len(base["bits"]) or np.count_nonzero(p["bits"] != base["bits"]) > 0)
            changed["fec"] += int(p["fec"] != base["fec"])
            changed["counters"] += int(p["counters"] != base["counters"])
            changed["centre"] += int(p["centre"] != base["centre"])
            same_len = len(p["det"]) == len(base["det"]) and len(p["peak"]) == len(base["peak"])
            if not same_len:
                changed["threshold"] += 1
                continue
            changed["threshold"] += int(np.count_nonzero((p["det"][:, 1] > 100.0) != (base["det"][:, 1] > 100.0)) > 0)
            changed["peak"] += int(np.count_nonzero(p["peak"][:, 1] != base["peak"][:, 1]) > 0)
            ddi = np.maximum(ddi, np.abs(p["det"][:, 0] - base["det"][:, 0]))
            de2 = np.maximum(de2, np.abs(p["det"][:, 1] - base["det"][:, 1]))
            dgap = np.maximum(dgap, np.abs(p["peak"][:, 0] - base["peak"][:, 0]))
        # margins in units of what the perturbation did to the quantity (decisions it did not move at all have no finite ratio)
        taken = base["det"][:, 1] > 100.0
        m_e2 = np.abs(base["det"][:, 1] - 100.0)[de2 > 0] / de2[de2 > 0]
        m_di = np.abs(base["det"][:, 0])[taken & (ddi > 0)] / ddi[taken & (ddi > 0)]
        m_gap = base["peak"][:, 0][dgap > 0] / dgap[dgap > 0]
        for k, m, n in (("di", m_di, int(np.count_nonzero(taken))), ("energy2", m_e2, len(base["det"])), ("argmax", m_gap, len(base["peak"]))):
            ndec[k] += n
            if m.size:
                worst[k] = min(worst[k], float(m.min()))
        report.append("%-44s bits %5d  FECDecode calls %d  largest change of di %.3g  energy2 %.3g  argmax gap %.3g | smallest margin / change: "
                      "di %.3g  energy2 %.3g  argmax %.3g"
                      % (name, len(base["bits"]), len(base["fec"]), ddi.max(initial=0.0), de2.max(initial=0.0), dgap.max(initial=0.0),
                         m_di.min(initial=np.inf), m_e2.min(initial=np.inf), m_gap.min(initial=np.inf)))
    runs = len(report) * len(PATTERNS)
    text = "\n".join(report)
    text += ("\n%d of the 512 table entries are not exactly 0 or +-1 and were moved by one ulp (all up / all down / random signs x 3 seeds = "
             "%d perturbed runs over %d streams, %d bits, %d FECDecode calls).\nRuns in which anything differed from the correctly rounded "
             "tables' run: bits %d, FECDecode results %d, counters %d, threshold outcomes %d, peak sequences %d, final centre bin %d.\n"
             "%d slicer decisions (di < 0), %d threshold decisions (energy2 > 100), %d peak argmaxes; smallest margin in units of the largest "
             "change the perturbation made to that decision's own quantity: di %.3g, energy2 %.3g, argmax %.3g."
             % (moved, runs, len(report), nbits, nfec, changed["bits"], changed["fec"], changed["counters"], changed["threshold"],
                changed["peak"], changed["centre"], ndec["di"], ndec["energy2"], ndec["argmax"], worst["di"], worst["energy2"], worst["argmax"]))
    print(text)
    with open(os.path.join(os.path.dirname(GOLD), "..", "profiles", "r06_sincos_ulp_robustness.txt"), "w") as f:
        f.write("tests/test_sincos_table_robustness.py (CPU, the C oracle)\n" + text + "\n")
    assert all(v == 0 for v in changed.values()), text
    assert min(worst.values()) > 1000.0, text
