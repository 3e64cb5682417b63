import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import java_sdr_amd as J
import oracle_lib as O
from test_gpu_demod import fm_am_signal
n, S, rate = 2048, 3, 96000
rng = np.random.default_rng(100 * 2 + 7 * n)
raws = [fm_am_signal(rng, 6 * n, rate, fc=5000.0 + 900.0 * s, seed_shift=31.0 * s)[:2 * n] for s in range(S)]
d = J.Demod(rate=rate, n=n, nstreams=S, max_batch_samples=n)
d.configure(2, 0, 0, 0)
d.weights(3000, 11000)
got = d.batch_host_i16(np.stack(raws), n)
for s in range(S):
    o = O.Demod(rate); o.configure(2, 0, 0, 0); o.weights(3000, 11000)
    want = o.receive(O.convert_i16(raws[s]))
    bad = np.flatnonzero(got[s] != want)
    buf = O.convert_i16(raws[s])
    amp = np.sqrt((buf[0::2] * buf[0::2] + buf[1::2] * buf[1::2]).astype(np.float64)).astype(np.float32)
    print(s, "stats gpu", d.frame_stats(s), "oracle", o.max, o.avg, "nbad", bad.size, bad[:6], got[s][bad[:6]], want[bad[:6]])
