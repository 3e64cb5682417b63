#!/usr/bin/env python3
"""Generate tests/golden/reference_fixtures.npz with the pure-Python restatement of the Java text.

java_restatement.py (tables parsed from /root/reference as data, arithmetic written from the .java text,
independent of oracle/*.c) is run on a handful of inputs; its outputs -- slicer bits, counters, every scalar
state double, the FECDecode log (rc, bit index, bytes), decoded[] and a (fi,fq) trace -- are committed.
tests/test_reference_fixtures.py holds the C oracle to them on the CPU, tests/test_gpu_fixtures.py the HIP path.

Inputs are data: the reference's own sine4410.raw, and synthetic DBPSK streams from the repo's integer
generator (oracle/o_synth.c == csrc/synth.hip); only their parameters and sha256 are stored, the tests
regenerate them and check the digest.  Runs in the build container only (needs /root/reference):

    python tests/golden/make_reference_fixtures.py        (about ten seconds)
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import java_restatement as J  # noqa: E402

from fixture_cases import FEC_CASES, STREAMS, TRACE, stream_input  # noqa: E402


def fec_blocks():
    """name -> uint8[5200] soft symbols; built with the Python encoder and numpy's seeded generator"""
    rng = np.random.default_rng(20020112)
    pay = rng.integers(0, 256, 256, dtype=np.uint8)
    sym = np.array(J.Encoder().encode_FEC40([int(v) for v in pay]), np.uint8)
    hard = np.where(sym == 1, 0xC0, 0x40).astype(np.uint8)
    out = {"clean": hard.copy()}
    b = hard.copy()
    b[rng.choice(5200, 200, replace=False)] ^= 0x80
    out["flips200"] = b
    b = hard.copy()
    b[1000:1400] ^= 0x80
    out["burst400"] = b
    soft = np.where(sym == 1, 0xA0, 0x60).astype(np.int32) + rng.integers(-40, 41, 5200)
    out["soft"] = np.clip(soft, 0, 255).astype(np.uint8)
    b = hard.copy()
    b[rng.choice(5200, 350, replace=False)] ^= 0x80
    out["flips350"] = b
    b = hard.copy()
    b[rng.choice(5200, 520, replace=False)] ^= 0x80
    out["flips520"] = b
    b = hard.copy()
    b[rng.choice(5200, 700, replace=False)] ^= 0x80
    out["flips700"] = b
    out["garbage"] = rng.integers(0, 256, 5200, dtype=np.uint8)
    # byte errors placed BEHIND the convolutional code: the Viterbi decoder returns exactly these bytes wrong and the RS
    # stage has to correct them -- 8 and 16 (the limit) per code word decode, 17 in one word does not
    for name, (n0, n1) in (("rs8", (8, 8)), ("rs16", (16, 16)), ("rs16_0", (16, 0)), ("rs17", (17, 3))):
        out[name] = np.where(np.array(encode_with_byte_errors(pay, n0, n1, rng), np.uint8) == 1, 0xC0, 0x40).astype(np.uint8)
    return pay, sym, out


class _ByteErrorEncoder(J.Encoder):
    """encode_FEC40 (:677-688) whose byte stream into the scrambler / convolutional encoder carries errors the RS parity
    does not know about (the parity is computed from the clean bytes)"""

    def __init__(self, errs):
        super().__init__()
        self.errs = errs  # position in the 320-byte stream -> xor mask

    def scramble_and_encode(self, c):
        super().scramble_and_encode(c ^ self.errs.get(self.Nbytes, 0))


def encode_with_byte_errors(pay, n0, n1, rng):
    errs = {}
    for blk, n in ((0, n0), (1, n1)):  # byte i of the stream belongs to RS word i & 1 (:617)
        for p in rng.choice(160, n, replace=False):
            errs[2 * int(p) + blk] = int(rng.integers(1, 256))
    return _ByteErrorEncoder(errs).encode_FEC40([int(v) for v in pay])


def main():
    t0 = time.time()
    out = {}
    sin_tab, cos_tab = J.sincos_tables()
    out["sin_tab"] = np.array(sin_tab)
    out["cos_tab"] = np.array(cos_tab)
    for name, p in STREAMS.items():
        raw = stream_input(name)
        d = J.Demod(rate=p["rate"], tuning=p["tuning"], trace_cap=TRACE)
        d.receive(J.convert_i16(raw, p["ic"], p["qc"]))
        k = "s_" + name + "_"
        out[k + "sha256"] = np.frombuffer(hashlib.sha256(raw.tobytes()).digest(), np.uint8)
        out[k + "counters"] = np.array(d.counters(), np.int32)
        out[k + "state"] = np.array(d.state(), np.float64)
        out[k + "istate"] = np.array(d.istate(), np.int32)
        out[k + "bits"] = np.array(d.bits, np.int8)
        out[k + "trace"] = np.array(d.trace, np.float64).reshape(-1, 2)
        out[k + "decoded"] = np.array(d.decoded, np.uint8)
        out[k + "fec_rc"] = np.array([f[0] for f in d.fec_log], np.int32)
        out[k + "fec_bitidx"] = np.array([f[1] for f in d.fec_log], np.int64)
        out[k + "fec_data"] = np.array([f[2] for f in d.fec_log], np.uint8).reshape(-1, 256)
        print(f"{name:8s} counters={d.counters()[:8]} fec={[(f[0], f[1]) for f in d.fec_log]}  [{time.time() - t0:.0f} s]",
              flush=True)
    pay, sym, blocks = fec_blocks()
    assert list(blocks) == FEC_CASES
    out["f_payload"] = pay
    out["f_symbols"] = sym
    for name, blk in blocks.items():
        dec = [0xEE] * 256  # RSdecdata stays untouched when RS fails (:780)
        rc = J.FECDecode([int(v) for v in blk], dec)
        out["f_" + name + "_in"] = blk
        out["f_" + name + "_rc"] = np.array([rc], np.int32)
        out["f_" + name + "_out"] = np.array(dec, np.uint8)
        print(f"fec {name:9s} rc={rc}  [{time.time() - t0:.0f} s]", flush=True)
    path = os.path.join(HERE, "reference_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
