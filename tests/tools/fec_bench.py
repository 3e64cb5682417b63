#!/usr/bin/env python3
"""GPU-side timing of the FEC kernels at different batch sizes (decode latency vs concurrency)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import java_sdr_amd as J
import oracle_lib as O

rng = np.random.default_rng(0)
data = rng.integers(0, 256, (64, 256), dtype=np.uint8)
sym = np.stack([O.fec_encode(d) for d in data])
soft_clean = np.where(sym == 1, 0xC0, 0x40).astype(np.uint8)
soft_err = soft_clean.copy()
for i in range(64):
    soft_err[i, rng.choice(5200, 400, replace=False)] ^= 0x80
# bursts the interleaver spreads over both RS code words: the Viterbi decoder leaves byte errors, the RS stage has to correct
soft_rs = soft_clean.copy()
for i in range(64):
    for b0 in (700, 2100, 3900):
        soft_rs[i, b0:b0 + 150] ^= 0x80
# blocks whose two RS code words need 16 corrections each (errors placed behind the convolutional code by the fixture generator)
FX = np.load(os.path.join(ROOT, "tests", "golden", "reference_fixtures.npz"))
soft_rs16 = np.tile(FX["f_rs16_in"], (64, 1))
for name, soft in (("clean", soft_clean), ("400 flips", soft_err), ("3 bursts", soft_rs), ("RS 16+16", soft_rs16)):
    for nb in (64, 256, 768, 1536, 2560, 5120):
        raws = np.tile(soft, (nb // 64, 1))
        d_raw = J.DeviceBuffer.from_host(raws)
        d_out = J.DeviceBuffer(256 * nb); d_rc = J.DeviceBuffer(4 * nb)
        t = J.Timer()
        J.fec_decode_dev(d_raw, nb, d_out, d_rc)
        J.binding.stream_sync(None)
        t.start(); J.fec_decode_dev(d_raw, nb, d_out, d_rc); t.stop()
        ms = t.elapsed_ms()
        rc = d_rc.to_host(np.int32)
        print(f"decode {name:10s} nblocks={nb:5d}  {ms:8.3f} ms  {ms / nb * 1e3:8.2f} us/block  rc[:4]={rc[:4]}")
nb = 4096
d_in = J.DeviceBuffer.from_host(np.tile(data, (nb // 64, 1)))
d_sym = J.DeviceBuffer(5200 * nb)
J.fec_encode_dev(d_in, nb, d_sym); J.binding.stream_sync(None)
t = J.Timer(); t.start(); J.fec_encode_dev(d_in, nb, d_sym); t.stop()
print(f"encode nblocks={nb}: {t.elapsed_ms():.3f} ms")
