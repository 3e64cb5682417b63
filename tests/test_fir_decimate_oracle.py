"""CPU: oracle/o_fir_phase.c:jo_fir_decimate (RxDownSample's ring-buffer state machine as an operator,
FUNcubeBPSKDemod.java:466-492) against (a) the sum's definition in Python floats at small n, for the tap counts and
decimations of BASELINE config 3, and (b) the down-sampler trace of the oracle's full demodulator, which the
independent Java restatement pins (tests/test_reference_fixtures.py)."""
import numpy as np
import pytest

import oracle_lib as O
from fir_defs import HOWARD, py_fir


def taps_for(ntaps, rng):
    if ntaps == 65:
        return O.bpsk_table(1)[:65]  # dmFilter (FUNcubeBPSKDemod.java:58-77)
    if ntaps == 27:
        return O.bpsk_table(0)  # dsFilter (:27-55)
    if ntaps == 21:
        return O.Fir().weights(500, 1500, 44100.0)  # fir.java:169-195
    return rng.standard_normal(ntaps)


@pytest.mark.parametrize("ntaps,decim", [(65, 1), (65, 10), (65, 20), (21, 1), (21, 10), (27, 1), (27, 10), (33, 3),
                                         (128, 7), (1, 1), (26, 101), (20, 110)])
def test_fir_decimate_equals_the_definition(ntaps, decim):
    rng = np.random.default_rng(ntaps * 1000 + decim)
    n = 700 + 3 * decim
    iq = rng.integers(-32768, 32768, 2 * n).astype(np.int16)
    taps = taps_for(ntaps, rng)
    got = O.fir_decimate(iq, taps, decim, 1.25)
    assert got.tobytes() == py_fir(iq, taps, decim, 1.25).tobytes()


@pytest.mark.parametrize("rate,decim", [(96000, 10), (192000, 20), (48000, 5), (44100, 4)])
def test_fir_decimate_equals_the_demodulators_downsampler(rate, decim):
    n = 20000 + 3 * decim + 1
    iq = O.make_dbpsk_stream(3, 1, n, rate=rate, noise_sigma=900.0)[0]
    o = O.Bpsk(rate=rate, blen=4, size=4, tuning=-1, trace=n // decim + 8)  # tuning <= 0: RxMixTuner passes through (:395)
    o.receive_i16(iq)
    assert O.fir_decimate(iq, O.bpsk_table(0), decim, HOWARD).tobytes() == o.trace_ds().tobytes()


def test_fir_decimate_rejects_bad_shapes():
    out = np.empty(4)
    raw = np.zeros(8, np.int16)
    t = np.zeros(200)
    assert O.lib().jo_fir_decimate(O.ptr(raw), 4, O.ptr(t), 129, 1, 1.0, O.ptr(out)) == -1
    assert O.lib().jo_fir_decimate(O.ptr(raw), 4, O.ptr(t), 27, 0, 1.0, O.ptr(out)) == -1
