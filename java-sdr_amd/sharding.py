"""Multi-GPU plumbing of the batched demodulator: independent streams are the shard, so rank r of
W owns a contiguous block of streams and no data-path collective is needed; the only exchange is ONE all-gather
per batch of fixed-size per-stream result slots (equal counts on every rank).

Slot layout (must match csrc/bpsk.hip k_pack_slots):
    int32 header[16] = {nbits, nfec, cntRaw, cntDS, cntBit, cntFEC, cntDec, dmErrBits, dmCorr, dmMaxCorr, decodeOK, 0..}
    int8  bits[slot_bits]
    nfec_max x { int32 rc; int32 bit_index; uint8 data[256] }
"""
import numpy as np

HEADER_BYTES = 64
FEC_ENTRY_BYTES = 264


def shard_streams(total_streams, world, rank):
    """contiguous, equal-sized shards (the gather needs equal counts): returns (first_stream, count)"""
    if total_streams % world:
        raise ValueError(f"{total_streams} streams do not split evenly over {world} ranks")
    per = total_streams // world
    return rank * per, per


def slot_layout(slot_bits, nfec_max):
    bits_offset = HEADER_BYTES
    fec_offset = HEADER_BYTES + slot_bits
    return dict(slot_bytes=fec_offset + nfec_max * FEC_ENTRY_BYTES, bits_offset=bits_offset, fec_offset=fec_offset,
                slot_bits=slot_bits, nfec_max=nfec_max)


def pack_slot(layout, counters, bits, fec):
    """numpy statement of k_pack_slots for one stream.  counters: the 9 ints after nbits/nfec; fec: [(rc, bit_index, data)]"""
    slot = np.zeros(layout["slot_bytes"], np.uint8)
    hdr = np.zeros(16, np.int32)
    hdr[0] = len(bits)
    hdr[1] = len(fec)
    hdr[2:2 + len(counters)] = counters
    slot[:HEADER_BYTES] = hdr.view(np.uint8)
    nb = min(len(bits), layout["slot_bits"])
    slot[layout["bits_offset"]:layout["bits_offset"] + nb] = np.asarray(bits[:nb], np.int8).view(np.uint8)
    for t, (rc, bi, data) in enumerate(fec[:layout["nfec_max"]]):
        o = layout["fec_offset"] + t * FEC_ENTRY_BYTES
        slot[o:o + 8] = np.array([rc, bi], np.int32).view(np.uint8)
        slot[o + 8:o + FEC_ENTRY_BYTES] = data
    return slot


def unpack_slot(slot, layout):
    slot = np.asarray(slot, np.uint8)
    hdr = slot[:HEADER_BYTES].view(np.int32)
    nb, nt = int(hdr[0]), int(hdr[1])
    nb = min(nb, layout["slot_bits"])
    bits = slot[layout["bits_offset"]:layout["bits_offset"] + nb].view(np.int8)
    fec = []
    for t in range(nt):
        o = layout["fec_offset"] + t * FEC_ENTRY_BYTES
        rc, bi = slot[o:o + 8].view(np.int32)
        fec.append((int(rc), int(bi), slot[o + 8:o + FEC_ENTRY_BYTES].copy()))
    return dict(header=hdr.copy(), bits=bits.copy(), fec=fec)


def all_gather_slots(dist, local_slots, world):
    """local_slots: torch uint8 tensor [S_local * slot_bytes]; returns the [world * S_local * slot_bytes] gather,
    ordered by rank == ordered by global stream id (contiguous shards)."""
    import torch
    out = torch.empty(world * local_slots.numel(), dtype=torch.uint8, device=local_slots.device)
    dist.all_gather_into_tensor(out, local_slots)
    return out
