"""N>1 path on CPU: 2 ranks over gloo.  Each rank owns a contiguous shard of streams, produces one fixed-size
result slot per stream (here from the oracle: tests may use it), and ONE all-gather assembles them in global
stream order -- the same sharding.py code path bench.py drives with RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_streams, nsamples, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from java_sdr_amd import sharding as SH

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = SH.shard_streams(total_streams, world, rank)
    layout = SH.slot_layout(slot_bits=8192, nfec_max=8)
    local = np.zeros(count * layout["slot_bytes"], np.uint8)
    for i in range(count):
        iq, _, _ = O.make_dbpsk_stream(20020109, first + i, nsamples)
        d = O.Bpsk()
        d.receive_i16(iq)
        c = d.counters()
        counters = [c[k] for k in ("cntRaw", "cntDS", "cntBit", "cntFEC", "cntDec", "dmErrBits", "dmCorr", "dmMaxCorr",
                                   "decodeOK")]
        fec = [(rc, bi, data) for rc, bi, data in d.fec_results()]
        local[i * layout["slot_bytes"]:(i + 1) * layout["slot_bytes"]] = SH.pack_slot(layout, counters, d.bits(), fec)
    gathered = SH.all_gather_slots(dist, torch.from_numpy(local), world).numpy()
    if rank == 0:
        q.put((gathered.copy(), layout))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_of_result_slots():
    import torch.multiprocessing as mp

    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from java_sdr_amd import sharding as SH

    world, total, nsamples = 2, 4, 458752
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, nsamples, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, layout = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert gathered.size == total * layout["slot_bytes"]
    for s in range(total):
        u = SH.unpack_slot(gathered[s * layout["slot_bytes"]:(s + 1) * layout["slot_bytes"]], layout)
        iq, pay, _ = O.make_dbpsk_stream(20020109, s, nsamples)
        d = O.Bpsk()
        d.receive_i16(iq)
        assert u["header"][0] == len(d.bits())
        assert np.array_equal(u["bits"], d.bits()[:layout["slot_bits"]])
        assert len(u["fec"]) == 1 and u["fec"][0][0] >= 0
        assert np.array_equal(u["fec"][0][2], pay[0])  # global stream order == rank order


def test_shard_streams_and_slot_roundtrip():
    sys.path.insert(0, ROOT)
    from java_sdr_amd import sharding as SH

    assert SH.shard_streams(8192, 8, 3) == (3072, 1024)
    with pytest.raises(ValueError):
        SH.shard_streams(10, 4, 0)
    lay = SH.slot_layout(96, 2)
    bits = np.array([1, -1, 1, 1, -1], np.int8)
    fec = [(3, 5202, np.arange(256, dtype=np.uint8))]
    u = SH.unpack_slot(SH.pack_slot(lay, list(range(9)), bits, fec), lay)
    assert np.array_equal(u["bits"], bits) and u["fec"][0][0] == 3 and u["fec"][0][1] == 5202
    assert list(u["header"][2:11]) == list(range(9))
