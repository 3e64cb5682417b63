"""GPU: jsdr_group_* -- one process, one host thread per device, the result slots gathered to every device after each call
(the reference hosts its demodulators in one JVM: jsdr.java:479-483).

What a one-GPU box can run: the RCCL path with ONE device (ncclCommInitAll over one rank, ncclAllGather on the gather
stream), and the whole multi-rank machinery -- threads, contiguous shards, rendezvous, slot order -- with several group
members on device 0 and device-to-device copies in RCCL's place (RCCL refuses a device twice).  Every gathered byte is
compared with a plain single-handle run over the same streams and with sharding.pack_slot's numpy statement."""
import numpy as np
import pytest

import java_sdr_amd as J
import oracle_lib as O

pytestmark = pytest.mark.gpu
SH = J.sharding
CN = ["cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr", "decodeOK"]


def streams(S, n, seed=20020109):
    return [O.make_dbpsk_stream(seed, s, n, noise_sigma=1200.0 + 90 * s)[0] for s in range(S)]


def reference_slots(iq, n, chunks):
    """a plain S-stream handle over the same calls: its packed slots after each call"""
    S = len(iq)
    d = J.Bpsk(nstreams=S, max_batch_samples=max(chunks))
    d_iq = J.DeviceBuffer.from_host(np.concatenate(iq))
    info = d.slot_info()
    out, pos = [], 0
    for c in chunks:
        d.batch_i16(d_iq.ptr + 4 * pos, 2 * n, c)
        slots = J.DeviceBuffer(S * info["slot_bytes"])
        d.pack_slots(slots)
        out.append(slots.to_host(np.uint8).reshape(S, info["slot_bytes"]).copy())
        pos += c
    return d, info, out


def run_group(iq, n, chunks, ndev, **kw):
    S = len(iq)
    per = S // ndev
    g = J.Group(ndev, S, max(chunks), devices=[0] * ndev if kw.get("gather_copy") else None, **kw)
    assert g.streams_per_device == per
    bufs = [J.DeviceBuffer.from_host(np.concatenate(iq[r * per:(r + 1) * per])) for r in range(ndev)]
    out, pos = [], 0
    for c in chunks:
        g.batch_i16([b.ptr + 4 * pos for b in bufs], 2 * n, c)
        out.append([g.gathered(r).copy() for r in range(ndev)])
        pos += c
    return g, out


def test_group_of_one_device_gathers_through_rccl():
    n = 458752
    chunks = [2048 * 100, n - 2048 * 100]
    iq = streams(6, n)
    d, info, want = reference_slots(iq, n, chunks)
    g, got = run_group(iq, n, chunks, 1)
    assert g.rccl_version > 0 and g.slot_bytes == info["slot_bytes"]
    for k in range(len(chunks)):
        assert np.array_equal(got[k][0], want[k])
    # the numpy statement of the slot, from the getters
    for s in range(6):
        u = SH.unpack_slot(got[-1][0][s], info)
        c = d.counters(s)
        slot = SH.pack_slot(info, [c[k] for k in CN], d.bits(s), d.fec_results(s))
        nb = min(len(d.bits(s)), info["slot_bits"])
        assert np.array_equal(u["bits"][:nb], d.bits(s)[:nb])
        assert np.array_equal(slot[:8], got[-1][0][s][:8])  # nbits, nfec
        assert [int(v) for v in u["header"][2:11]] == [c[k] for k in CN]
        assert len(u["fec"]) == 1 and u["fec"][0][0] >= 0
    assert np.array_equal(g.read_slot(0, 3), got[-1][0][3])


@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_group_rehearsal_several_members_on_one_device(ndev):
    """threads, shards, rendezvous and slot order of an N-device group, on one device with copies in RCCL's place"""
    n = 2048 * 60
    chunks = [2048 * 25 + 77, 2048 * 20 - 77, 2048 * 15]
    iq = streams(16, n, seed=77)
    _, info, want = reference_slots(iq, n, chunks)
    g, got = run_group(iq, n, chunks, ndev, gather_copy=True)
    assert g.rccl_version == 0
    for k in range(len(chunks)):
        for r in range(ndev):
            assert np.array_equal(got[k][r], want[k]), (k, r)
    dev, view = g.device(ndev - 1)
    assert dev == 0 and view.counters(0)["cntRaw"] == n


def test_group_with_psd_and_error_paths():
    n = 2048 * 40
    iq = streams(4, n, seed=5)
    g = J.Group(2, 4, n, devices=[0, 0], gather_copy=True, with_psd=True)
    bufs = [J.DeviceBuffer.from_host(np.concatenate(iq[2 * r:2 * r + 2])) for r in range(2)]
    psds = [J.DeviceBuffer(2 * 40 * 2050 * 4) for _ in range(2)]
    g.batch_i16(bufs, 2 * n, n, psd_devs=psds)
    g.sync()
    f = J.Fft(2048, 96000)
    for r in range(2):
        ref = J.DeviceBuffer(2 * 40 * 2050 * 4)
        f.batch_i16(bufs[r], 80, ref)
        J.binding.stream_sync()
        assert psds[r].to_host(np.float32).tobytes() == ref.to_host(np.float32).tobytes()
    # a PSD call over PART of the buffers (stride != 2 nsamples: the chunked calls of a longer buffer, the JNI's 2 max_batch
    # stride): every stream's frames are taken from ITS row, not from one contiguous run
    half = n // 2
    psds2 = [J.DeviceBuffer(2 * 20 * 2050 * 4) for _ in range(2)]
    g.batch_i16([b.ptr + 4 * half for b in bufs], 2 * n, half, psd_devs=psds2)
    g.sync()
    for r in range(2):
        got = psds2[r].to_host(np.float32).reshape(2, 20, 2050)
        for s in range(2):
            ref = J.DeviceBuffer(20 * 2050 * 4)
            f.batch_i16(bufs[r].ptr + 4 * (s * n + half), 20, ref)
            J.binding.stream_sync()
            assert got[s].tobytes() == ref.to_host(np.float32).tobytes(), (r, s)
    # a call every member refuses (more samples than the handles were sized for): reported, nobody hangs, the group lives on
    with pytest.raises(J.JsdrError, match="rank"):
        g.batch_i16(bufs, 2 * n, 2 * n)
    g.sync()
    with pytest.raises(J.JsdrError):
        g.batch_i16(bufs, 2 * n, 1000, psd_devs=psds)  # the PSD needs whole frames
    g.batch_i16(bufs, 2 * n, 2048)
    a = g.gathered(0)
    b = g.gathered(1)
    assert np.array_equal(a, b)
    # RCCL takes a device once
    with pytest.raises(J.JsdrError, match="twice"):
        J.Group(2, 4, n, devices=[0, 0])


def test_group_block_of_a_recording_shorter_than_the_block(tmp_path):
    """HipDemodGroup.decodeBlock's last block: jsdr_recordings_load pads a short recording with zeros on the
    NULL stream, the group's device threads demodulate on non-blocking streams of their own -- the padding must be in
    place when the load returns, or the kernels read the previous block's samples."""
    n = 2048 * 24
    iq = streams(2, n, seed=11)
    short = n // 2 + 123
    paths = []
    for s in range(2):
        p = tmp_path / f"s{s}.raw"
        iq[s][:2 * short].astype("<i2").tofile(p)
        paths.append(str(p))
    g = J.Group(1, 2, n, devices=[0], gather_copy=True)
    stale = np.concatenate(streams(2, n, seed=12))  # what the previous block left in the buffer
    for trial in range(3):
        buf = J.DeviceBuffer.from_host(stale)
        got = J.recordings_load(paths, 2, 96000, 0, n, buf, 2 * n)
        assert got == [short, short]
        g.batch_i16([buf], 2 * n, n)
    g.sync()
    padded = [np.concatenate([iq[s][:2 * short], np.zeros(2 * (n - short), np.int16)]) for s in range(2)]
    d = J.Bpsk(nstreams=2, max_batch_samples=n)  # a plain handle fed the zero-padded block three times
    d_iq = J.DeviceBuffer.from_host(np.concatenate(padded))
    for trial in range(3):
        d.batch_i16(d_iq, 2 * n, n)
    info = d.slot_info()
    slots = J.DeviceBuffer(2 * info["slot_bytes"])
    d.pack_slots(slots)
    assert np.array_equal(g.gathered(0).reshape(2, info["slot_bytes"]), slots.to_host(np.uint8).reshape(2, info["slot_bytes"]))


# ------------------------------------------------------------------ the C++ harness's --gpus mode: config 5 with no Python in the process
import json
import os
import subprocess

HARNESS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "java-sdr_amd", "host", "jsdr_harness")


@pytest.mark.parametrize("args", [
    ["--gpus", "1", "--streams", "256", "--psd"],                                         # RCCL, one rank
    ["--gpus", "4", "--streams", "512", "--same-device", "--copy-gather"],                # 4 host threads, shards of 128
    ["--gpus", "8", "--streams", "512", "--same-device", "--copy-gather", "--psd"],       # the driver's N = 8 shape, rehearsed
])
def test_harness_gpus_mode(args):
    assert os.path.exists(HARNESS), "build with __graft_entry__.build() (make -C java-sdr_amd/host)"
    r = subprocess.run([HARNESS] + args + ["--samples", str(2048 * 448), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["validated"] is True and j["every_device_holds_the_same_slots"] is True
    assert j["streams_without_decoded_frame"] == 0 and j["decoded_frames_matching_no_sent_payload"] == 0
    assert j["decoded_frames"] >= j["total_streams"]
    assert (j["rccl_version"] > 0) == (j["gather"] == "ncclAllGather")


def test_group_in_fft_acquire_mode_equals_a_single_handle():
    """the group carries whatever demodulator configuration it is given: FFT-acquire mode (bpsk-dofft) at the default 2048-sample
    frame, three members on one device -- every gathered byte equals a plain 6-stream handle's slots"""
    n = 2048 * 36
    chunks = [2048 * 20, 2048 * 16]
    iq = [O.make_dbpsk_stream(61, s, n, carrier_hz=13200.0 + 90.0 * s, noise_sigma=500.0 + 200 * s)[0] for s in range(6)]
    d = J.Bpsk(nstreams=6, max_batch_samples=max(chunks), do_fft=1)
    d_iq = J.DeviceBuffer.from_host(np.concatenate(iq))
    info = d.slot_info()
    g = J.Group(3, 6, max(chunks), devices=[0, 0, 0], do_fft=1, gather_copy=True)
    bufs = [J.DeviceBuffer.from_host(np.concatenate(iq[2 * r:2 * r + 2])) for r in range(3)]
    pos = 0
    for c in chunks:
        d.batch_i16(d_iq.ptr + 4 * pos, 2 * n, c)
        slots = J.DeviceBuffer(6 * info["slot_bytes"])
        d.pack_slots(slots)
        want = slots.to_host(np.uint8).reshape(6, info["slot_bytes"])
        g.batch_i16([b.ptr + 4 * pos for b in bufs], 2 * n, c)
        for r in range(3):
            assert np.array_equal(g.gathered(r), want), (pos, r)
        pos += c
    _, view = g.device(1)
    assert view.counters(0)["centreBin"] == d.counters(2)["centreBin"]
