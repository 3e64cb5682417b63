"""per-kernel times of the BPSK chain at another sample rate: python tools/rate_bench.py [rate=192000] [S=1024]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import java_sdr_amd as J

rate = int(sys.argv[1]) if len(sys.argv) > 1 else 192000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = 1 << 20
d = J.Bpsk(rate=rate, blen=8192, tuning=12000, nstreams=S, max_batch_samples=L)
one = np.random.default_rng(1).integers(-3000, 3000, 2 * L).astype(np.int16)
buf = J.DeviceBuffer(S * L * 4)
for s in range(S):
    J.lib().jsdr_memcpy_h2d(J.binding.C.c_void_p(buf.ptr + s * L * 4), J.binding._addr(one), J.binding.C.c_size_t(L * 4))
for _ in range(2):
    d.batch_i16(buf, 2 * L, L)
d.sync()
d.profile_read()
d.profile_enable(True)
for _ in range(3):
    d.batch_i16(buf, 2 * L, L)
d.sync()
print(rate, {k: round(v[0] / max(v[1], 1), 3) for k, v in d.profile_read().items() if v[1]})
