"""GPU: jsdr_fft_set_cu_share / jsdr_bpsk_set_cu_share -- the PSD kernel and the demodulator's front-end kernel as a fixed
number of persistent workgroups per CU, so that a caller can run the two plugins jsdr.java:476,482 feeds from one audio
buffer SIDE BY SIDE on two streams.  The shares change where the work runs, never a result."""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
N = 2048


def batch(S, L, seed):
    rows = [O.make_dbpsk_stream(seed + s, s, L, carrier_hz=13200.0 + 41.0 * s, noise_sigma=400.0 + 150 * s)[0][:2 * L] for s in range(S)]
    return np.ascontiguousarray(np.stack(rows))


def run(raw, S, chunks, shares, two_streams):
    fft = J.Fft(N, 96000)
    dem = J.Bpsk(nstreams=S, max_batch_samples=max(chunks))
    if shares:
        fft.set_cu_share(shares[0])
        dem.set_cu_share(shares[1])
    s1 = J.Stream()
    s2 = J.Stream() if two_streams else s1
    out = []
    pos = 0
    for c in chunks:
        d_iq = J.DeviceBuffer.from_host(np.ascontiguousarray(raw[:, 2 * pos:2 * (pos + c)]))
        d_psd = J.DeviceBuffer(S * (c // N) * (N + 2) * 4)
        dem.batch_i16(d_iq, 2 * c, c, stream=s1.ptr)
        fft.batch_i16(d_iq, S * (c // N), d_psd, stream=s2.ptr)
        s2.sync()
        s1.sync()
        dem.sync()
        out.append((d_psd.to_host(np.float32), [dem.bits(s).copy() for s in range(S)], [dem.trace(s).copy() for s in range(S)]))
        pos += c
    return dem, out


@pytest.mark.parametrize("shares", [(2, 1), (1, 1), (4, 2)])
def test_cu_shares_change_no_result(shares):
    S, L = 6, 2048 * 120
    raw = batch(S, L, 31 + shares[0])
    chunks = [2048 * 70, 2048 * 50]
    d0, o0 = run(raw, S, chunks, None, False)
    d1, o1 = run(raw, S, chunks, shares, True)
    for (p0, b0, t0), (p1, b1, t1) in zip(o0, o1):
        assert p0.tobytes() == p1.tobytes()
        for s in range(S):
            assert np.array_equal(b0[s], b1[s]) and t0[s].tobytes() == t1[s].tobytes()
    for s in range(S):
        assert d0.counters(s) == d1.counters(s) and d0.state(s).tobytes() == d1.state(s).tobytes()
        assert np.array_equal(d0.decoded(s), d1.decoded(s))


def test_cu_share_arguments():
    fft = J.Fft(N, 96000)
    dem = J.Bpsk(nstreams=2, max_batch_samples=4096)
    with pytest.raises(J.JsdrError):
        fft.set_cu_share(-1)
    with pytest.raises(J.JsdrError):
        dem.set_cu_share(100)
    fft.set_cu_share(0)
    dem.set_cu_share(0)
