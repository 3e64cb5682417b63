"""time jsdr_fft_batch_i16 at a given frame size: python tools/fft_n_bench.py [n=9600] [Msamples=1024]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import java_sdr_amd as J

n = int(sys.argv[1]) if len(sys.argv) > 1 else 9600
total = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 20
nframes = total // n
rate = {9600: 96000, 4800: 48000, 19200: 192000}.get(n, 96000)
f = J.Fft(n, rate)
d_raw = J.DeviceBuffer(nframes * n * 4)
d_raw.zero()
one = np.random.default_rng(0).integers(-20000, 20000, 2 * n * 64).astype(np.int16)
J.lib().jsdr_memcpy_h2d(J.binding.C.c_void_p(d_raw.ptr), J.binding._addr(one), J.binding.C.c_size_t(one.nbytes))
d_psd = J.DeviceBuffer(nframes * (n + 2) * 4)
t = J.Timer()
for _ in range(2):
    f.batch_i16(d_raw, nframes, d_psd)
J.binding.stream_sync(None)
t.start()
for _ in range(5):
    f.batch_i16(d_raw, nframes, d_psd)
t.stop()
ms = t.elapsed_ms() / 5
print(f"n={n} ({f.kernel_name()}): {nframes} frames in {ms:.3f} ms = {nframes * n / ms / 1e6:.1f} Gsamples/s, {nframes * n * 8.0 / ms / 1e9:.2f} TB/s algorithmic")
