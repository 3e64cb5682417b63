"""GPU: the ORACLE on the execution modes the headline number comes from.

`python bench.py` runs fft.receive (fft.java:190-228) and FUNcubeBPSKDemod.receive (FUNcubeBPSKDemod.java:466-595) over
8192 streams x 2^20 samples with k_fft and k_fm as PERSISTENT workgroups that stride over their work items
(jsdr_*_set_cu_share), with k_tail8 (eight streams per wave) and the batch FEC decoder (k_vitq) at their native width.
The parity suites elsewhere use a handful of streams, where no workgroup ever takes a second work item and the small
handles take k_tail / the one-wave FEC.  Here:

 (a) a batch with several times more work items than workgroups under the shares, on two streams -- sampled streams'
     (fi,fq) trace, bits, every state double, counters and FEC bytes against O.Bpsk, sampled PSD rows against
     O.fft_receive, and the launch getters prove that both grids really were capped;
 (b) 2048 streams (k_tail8's and k_vitq's native width, no environment override) x 2^17 samples over two ragged calls:
     8 sampled streams against the oracle, every stream's payload round trip.
"""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
N = 2048
CN = ["cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK"]
STATE_IDX = (0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17)


def synth_batch(S, L, nfr, seed):
    """the bench's own device-side generator (bench.py make_streams): FEC-carrying DBPSK at 13 200 Hz, one payload set per stream"""
    pay = J.synth_payloads(seed, 0, S, nfr)
    d_sym = J.DeviceBuffer(S * nfr * 5200)
    J.fec_encode_dev(pay, S * nfr, d_sym)
    d_ds = J.DeviceBuffer(S * nfr * 5200)
    J.synth_diffsign(d_sym, nfr * 5200, S, d_ds)
    ct, st = O.synth_tables(3000)
    keys = np.array([O.mix64((seed * 0x9E3779B1 + s) ^ 0xA5A5A5A5) for s in range(S)], np.uint64)
    d_iq = J.DeviceBuffer(S * L * 4)
    gain = int(round(1500.0 / 37837.0 * 32768.0))
    J.synth_dbpsk(d_iq, 2 * L, S, 0, L, d_ds, nfr * 5200, 80, 0, O.phase_inc_u32(13200.0, 96000),
                  J.DeviceBuffer.from_host(ct), J.DeviceBuffer.from_host(st), gain, J.DeviceBuffer.from_host(keys))
    return d_iq, pay.to_host(np.uint8).reshape(S, nfr, 256)


def check_stream_against_oracle(d, d_iq, L, s, bits, trace, fec):
    iq = d_iq.to_host(np.int16, count=2 * L, offset_bytes=4 * L * s)
    o = O.Bpsk(trace=L // 10 + 8)
    o.receive_i16(iq)
    assert np.array_equal(np.concatenate(bits), o.bits()), f"stream {s}: bits differ"
    if trace is not None:
        assert np.concatenate(trace).tobytes() == o.trace().tobytes(), f"stream {s}: (fi,fq) differ"
    fo = o.fec_results()
    assert len(fec) == len(fo), (s, len(fec), len(fo))
    for (rc, _, data), (orc, _, odata) in zip(fec, fo):
        assert rc == orc and np.array_equal(data, odata), s
    c, oc = d.counters(s), o.counters()
    assert [c[n] for n in CN] == [oc[n] for n in CN], (s, c, oc)
    gs, os_ = d.state(s), o.state()
    for i in STATE_IDX:
        assert gs[i] == os_[i], (s, i, gs[i], os_[i])
    assert np.array_equal(d.decoded(s), o.decoded()), s
    return iq


def test_persistent_stride_under_cu_shares_against_the_oracle():
    """112 streams x 2^20 samples in two calls with the shipped split (2 + 1 workgroups per CU) on two streams: k_fm takes
    ~6 tiles per workgroup, k_fft ~28 groups of frames per workgroup -- the headline's mode, checked against the oracle."""
    S, L = 112, 1048576
    chunks = [2048 * 300, 2048 * 212]
    d_iq, payloads = synth_batch(S, L, 3, 20020109)
    fft = J.Fft(N, 96000)
    dem = J.Bpsk(nstreams=S, max_batch_samples=max(chunks))
    fft.set_cu_share(2)
    dem.set_cu_share(1)
    s1, s2 = J.Stream(), J.Stream()
    sampled = (0, 1, 17, 40, 63, 64, 110, 111)
    bits = {s: [] for s in sampled}
    trace = {s: [] for s in sampled}
    fec = {s: [] for s in sampled}
    psd_rows = []
    pos = 0
    for c in chunks:
        nfr = c // N
        d_psd = J.DeviceBuffer(S * nfr * (N + 2) * 4)
        # the bench's launch shape: one batch call over [S][stride] for the demodulator, one PSD call per stream-major
        # block of frames -- frames of a stream are contiguous inside the call's window only, so the PSD goes stream by
        # stream here except for the capped launch below, which takes stream 0's whole remaining row of the buffer
        dem.batch_i16(d_iq.ptr + 4 * pos, 2 * L, c, stream=s1.ptr)
        wi, wg = dem.last_launch()
        assert wi >= 4 * wg, f"k_fm: {wi} tiles on {wg} workgroups -- the stride path must take >= 4 tiles per workgroup"
        for s in range(S):
            fft.batch_i16(d_iq.ptr + 4 * (L * s + pos), nfr, d_psd.ptr + 4 * (N + 2) * nfr * s, stream=s2.ptr)
        s2.sync()
        s1.sync()
        dem.sync()
        for s in sampled:
            bits[s].append(dem.bits(s).copy())
            trace[s].append(dem.trace(s).copy())
            fec[s].extend(dem.fec_results(s))
        psd_rows.append((pos, nfr, d_psd))
        pos += c
    assert pos == L
    # one PSD launch that is capped by the share: all frames of the whole buffer as one batch (S * 512 frames)
    d_all = J.DeviceBuffer(S * (L // N) * (N + 2) * 4)
    fft.batch_i16(d_iq, S * (L // N), d_all, stream=s2.ptr)
    s2.sync()
    wi, wg = fft.last_launch()
    assert wg == 512 and wi >= 4 * wg, f"k_fft: {wi} groups of frames on {wg} workgroups"
    allpsd = d_all.to_host(np.float32).reshape(S, L // N, N + 2)
    # the same frames through an unshared handle: identical bits
    fft0 = J.Fft(N, 96000)
    d_ref = J.DeviceBuffer(S * (L // N) * (N + 2) * 4)
    fft0.batch_i16(d_iq, S * (L // N), d_ref)
    J.binding.stream_sync()
    wi0, wg0 = fft0.last_launch()
    assert wg0 > wg
    assert np.array_equal(allpsd.view(np.int32).ravel(), d_ref.to_host(np.float32).view(np.int32))
    # the per-call PSD rows equal the one-batch rows
    for pos0, nfr, d_psd in psd_rows:
        rows = d_psd.to_host(np.float32).reshape(S, nfr, N + 2)
        assert np.array_equal(rows.view(np.int32), allpsd[:, pos0 // N:pos0 // N + nfr].view(np.int32))
    for s in sampled:
        iq = check_stream_against_oracle(dem, d_iq, L, s, bits[s], trace[s], fec[s])
        assert len(fec[s]) == 2 and all(rc >= 0 for rc, _, _ in fec[s])
        assert np.array_equal(fec[s][0][2], payloads[s, 0]) and np.array_equal(fec[s][1][2], payloads[s, 1])
        # PSD rows of this stream against fft.receive's restatement: frames early, at the call seam, last
        buf = O.convert_i16(iq)
        for k in (0, 1, 299, 300, 301, 511):
            ref = O.fft_receive(buf[2 * N * k:2 * N * (k + 1)], 96000)
            got = allpsd[s, k]
            lin_g = 10.0 ** (got[:N].astype(np.float64) / 20)
            lin_r = 10.0 ** (ref[:N].astype(np.float64) / 20)
            assert np.abs(lin_g - lin_r).max() <= 1e-5 * lin_r.max(), (s, k)
            assert got[N] == ref[N] or abs(got[N + 1] - ref[N + 1]) < 1e-4, (s, k, got[N], ref[N])


def test_native_width_tail8_and_batch_fec_against_the_oracle():
    """2048 streams: the handle takes k_tail8 and the batch FEC decoder by itself (no JSDR_TAIL8 override).  Two ragged
    calls carrying one full FEC frame; 8 sampled streams against the oracle, every stream by payload."""
    S, L = 2048, 2048 * 224 + 2048 * 40  # 540 672 samples: 5200 symbols x 80 = 416 000 + acquisition + a call seam after the frame
    chunks = [268365, L - 268365]
    d_iq, payloads = synth_batch(S, L, 2, 20020111)
    dem = J.Bpsk(nstreams=S, max_batch_samples=max(chunks))
    sampled = (0, 7, 8, 1023, 1024, 1500, 2040, 2047)
    bits = {s: [] for s in sampled}
    trace = {s: [] for s in sampled}
    fec = {s: [] for s in sampled}
    nfec = np.zeros(S, np.int64)
    pos = 0
    for c in chunks:
        dem.batch_i16(d_iq.ptr + 4 * pos, 2 * L, c)
        assert dem.tail_kernel_name() == "k_tail8", dem.tail_kernel_name()
        assert "k_vitq" in dem.fec_kernel_name(), dem.fec_kernel_name()
        for s in sampled:
            bits[s].append(dem.bits(s).copy())
            trace[s].append(dem.trace(s).copy())
            fec[s].extend(dem.fec_results(s))
        info = dem.slot_info()
        slots = J.DeviceBuffer(S * info["slot_bytes"])
        dem.pack_slots(slots)
        blob = slots.to_host(np.uint8).reshape(S, info["slot_bytes"])
        for s in range(S):
            for rc, _, data in J.sharding.unpack_slot(blob[s], info)["fec"]:
                if rc >= 0:  # a sync hit whose decode fails (rc -1) is the reference's behaviour too: the oracle check below covers those
                    assert np.array_equal(data, payloads[s, nfec[s]]), (s, rc)
                    nfec[s] += 1
        pos += c
    assert (nfec == 1).all(), np.flatnonzero(nfec != 1)[:10]
    for s in sampled:
        check_stream_against_oracle(dem, d_iq, L, s, bits[s], trace[s], fec[s])


def test_library_is_deaf_to_its_knobs_without_the_master_switch():
    """JSDR_TAIL8=2 forces k_tail8 on small handles -- but only when JSDR_KNOBS=1 is set as well (the tests' conftest sets it):
    a stray tuning variable in a production environment must change nothing"""
    import os
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, java_sdr_amd as J\n"
            "d = J.Bpsk(nstreams=2, max_batch_samples=4096); b = J.DeviceBuffer(2 * 4096 * 4); b.zero()\n"
            "d.batch_i16(b, 2 * 4096, 4096); print(d.tail_kernel_name())" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, JSDR_TAIL8="2")
    env.pop("JSDR_KNOBS", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("k_tail"), r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, JSDR_KNOBS="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("k_tail8"), r.stdout + r.stderr
