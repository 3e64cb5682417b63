"""FFT-acquire mode at the reference's default frame (n = 9600): time per call of 1024 streams x 109 frames."""
import faulthandler
import os
import sys
import time

faulthandler.enable()

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import java_sdr_amd as J

S, n, nfr = int(os.environ.get('FM_S', '1024')), 9600, int(os.environ.get('FM_NFR', '109'))
L = n * nfr
d = J.Bpsk(rate=96000, blen=4 * n, tuning=12000, do_fft=1, nstreams=S, max_batch_samples=L)
rng = np.random.default_rng(1)
one = rng.integers(-3000, 3000, 2 * L).astype(np.int16)
buf = J.DeviceBuffer(S * L * 4)
for s in range(S):  # same noise in every stream: timing only
    J.lib().jsdr_memcpy_h2d(J.binding.C.c_void_p(buf.ptr + s * L * 4), J.binding._addr(one), J.binding.C.c_size_t(L * 4))
print('inputs up', flush=True)
for _ in range(2):
    d.batch_i16(buf, 2 * L, L)
    d.sync()
    print('warm call done', flush=True)
d.profile_read()
d.profile_enable(True)
t0 = time.perf_counter()
for _ in range(3):
    d.batch_i16(buf, 2 * L, L)
d.sync()
dt = (time.perf_counter() - t0) / 3
print("ms per call %.2f  Gsamples/s %.1f" % (dt * 1e3, S * L / dt / 1e9), {k: round(v[0] / max(v[1], 1), 3) for k, v in d.profile_read().items() if v[1]})
