"""GPU parity: jsdr_fir_batch_decimate_i16 (BASELINE config 3) -- exact-order FP64, bit-identical to
(a) the oracle's RxDownSample trace (FUNcubeBPSKDemod.java:466-492; tuner off so that the stage stands alone) and
(b) a plain numpy/Python statement of the sum, newest sample first, for the other tap counts and decimations."""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

from fir_defs import HOWARD, py_fir

pytestmark = pytest.mark.gpu


def gpu_fir(iq_streams, n, taps, decim, scale):
    S = len(iq_streams)
    d_iq = J.DeviceBuffer.from_host(np.concatenate(iq_streams))
    no = n // decim
    d_out = J.DeviceBuffer(S * max(no, 1) * 16)
    got = J.fir_batch_decimate_i16(d_iq, S, 2 * n, n, taps, decim, scale, d_out, max(no, 1))
    assert got == no
    return d_out.to_host(np.float64).reshape(S, max(no, 1), 2)[:, :no]


@pytest.mark.parametrize("rate,decim", [(96000, 10), (192000, 20), (48000, 5), (44100, 4)])
def test_fir_batch_equals_the_oracles_downsampler(rate, decim):
    n = 65536 + 7 * decim + 3
    rng = np.random.default_rng(rate)
    streams = [O.make_dbpsk_stream(3, s, n, rate=rate, noise_sigma=900.0)[0] for s in range(2)]
    streams.append(rng.integers(-32768, 32768, 2 * n).astype(np.int16))
    taps = O.bpsk_table(0)
    got = gpu_fir(streams, n, taps, decim, HOWARD)
    for s, iq in enumerate(streams):
        o = O.Bpsk(rate=rate, blen=4, size=4, tuning=-1, trace=n // decim + 8)  # tuning <= 0: RxMixTuner passes through (:395)
        o.receive_i16(iq)
        want = o.trace_ds()
        assert want.shape[0] == n // decim
        assert got[s].tobytes() == want.tobytes(), (rate, s)


@pytest.mark.parametrize("ntaps,decim", [(65, 1), (65, 10), (65, 20), (21, 1), (21, 10), (27, 1), (33, 3), (128, 7), (1, 1),
                                         (26, 101), (20, 110), (20, 120), (64, 110)])  # the last four: pairs whose
# ntaps * 100 + decim collides with a register-blocked kernel's key -- they must take the generic kernel
def test_fir_batch_other_taps_and_decimations_against_the_definition(ntaps, decim):
    n = 2500 + 3 * decim
    rng = np.random.default_rng(ntaps * 100 + decim)
    iq = rng.integers(-32768, 32768, 2 * n).astype(np.int16)
    if ntaps == 65:
        taps = O.bpsk_table(1)[:65]  # dmFilter
    elif ntaps == 21:
        taps = O.Fir().weights(500, 1500, 44100.0)  # fir.java's window (:169-195)
    else:
        taps = rng.standard_normal(ntaps)
    got = gpu_fir([iq, iq[::-1].copy()], n, taps, decim, 1.25)
    assert got[0].tobytes() == py_fir(iq, taps, decim, 1.25).tobytes()
    assert got[1].tobytes() == py_fir(iq[::-1].copy(), taps, decim, 1.25).tobytes()


@pytest.mark.parametrize("ntaps,decim", [(65, 10), (65, 1), (21, 1), (21, 10), (27, 10), (65, 20)])
def test_fir_batch_config3_shape_one_million_samples_by_four_streams(ntaps, decim):
    """BASELINE config 3: "65-tap low-pass + decimate over 1 M-sample IQ batch" -- the register-blocked kernels across
    every workgroup boundary of a 2^20-sample stream, four streams (DBPSK, noise, full-scale square wave, a ramp),
    every output bit-identical to the oracle's RxDownSample operator (jo_fir_decimate, pinned on the CPU by
    tests/test_fir_decimate_oracle.py)"""
    n = 1 << 20
    rng = np.random.default_rng(ntaps * 7 + decim)
    sq = np.where((np.arange(2 * n) // 14) % 2 == 0, 32767, -32768).astype(np.int16)
    ramp = (np.arange(2 * n) * 37 % 65536 - 32768).astype(np.int16)
    streams = [O.make_dbpsk_stream(5, 2, n)[0], rng.integers(-32768, 32768, 2 * n).astype(np.int16), sq, ramp]
    taps = O.bpsk_table(1)[:65] if ntaps == 65 else (O.bpsk_table(0) if ntaps == 27 else O.Fir().weights(500, 1500, 44100.0))
    got = gpu_fir(streams, n, taps, decim, HOWARD)
    for s, iq in enumerate(streams):
        want = O.fir_decimate(iq, taps, decim, HOWARD)
        assert got[s].shape == want.shape
        assert got[s].tobytes() == want.tobytes(), (ntaps, decim, s, int(np.argmax((got[s] != want).any(axis=1))))


def test_fir_batch_api_errors_and_empty():
    buf = J.DeviceBuffer(4096)
    out = J.DeviceBuffer(4096)
    with pytest.raises(J.JsdrError):
        J.fir_batch_decimate_i16(buf, 1, 2048, 1024, np.zeros(129), 1, 1.0, out, 256)
    with pytest.raises(J.JsdrError):
        J.fir_batch_decimate_i16(buf, 1, 2048, 1024, np.zeros(27), 0, 1.0, out, 256)
    with pytest.raises(J.JsdrError):
        J.fir_batch_decimate_i16(buf, 2, 100, 1024, np.zeros(27), 10, 1.0, out, 256)  # stride < samples
    assert J.fir_batch_decimate_i16(buf, 1, 2048, 7, np.ones(27), 10, 1.0, out, 256) == 0
