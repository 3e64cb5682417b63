"""GPU parity: jsdr_fir_batch_decimate_i16 (BASELINE config 3) -- exact-order FP64, bit-identical to
(a) the oracle's RxDownSample trace (FUNcubeBPSKDemod.java:466-492; tuner off so that the stage stands alone) and
(b) a plain numpy/Python statement of the sum, newest sample first, for the other tap counts and decimations."""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
HOWARD = 0.9 * 32768.0


def gpu_fir(iq_streams, n, taps, decim, scale):
    S = len(iq_streams)
    d_iq = J.DeviceBuffer.from_host(np.concatenate(iq_streams))
    no = n // decim
    d_out = J.DeviceBuffer(S * max(no, 1) * 16)
    got = J.fir_batch_decimate_i16(d_iq, S, 2 * n, n, taps, decim, scale, d_out, max(no, 1))
    assert got == no
    return d_out.to_host(np.float64).reshape(S, max(no, 1), 2)[:, :no]


def py_fir(iq, taps, decim, scale):
    """the definition, in Python floats (IEEE doubles, no FMA): newest sample first"""
    x = O.convert_i16(iq).astype(np.float64).reshape(-1, 2)
    n = x.shape[0]
    out = []
    for j in range(n // decim):
        newest = decim * (j + 1) - 1
        fi = fq = 0.0
        for a, t in enumerate(taps):
            k = newest - a
            if k >= 0:
                fi += float(x[k, 0]) * float(t)
                fq += float(x[k, 1]) * float(t)
            else:
                fi += 0.0 * float(t)
                fq += 0.0 * float(t)
        out.append((fi * scale, fq * scale))
    return np.array(out, np.float64).reshape(-1, 2)


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


@pytest.mark.parametrize("ntaps,decim", [(65, 1), (65, 10), (65, 20), (21, 1), (21, 10), (27, 1), (33, 3), (128, 7), (1, 1)])
def test_fir_batch_other_taps_and_decimations_against_the_definition(ntaps, decim):
    n = 2500
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
