import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import java_sdr_amd as J
n = 2048
rng = np.random.default_rng(1)
raw = rng.integers(-8000, 8000, 2 * n).astype(np.int16)
d2 = J.Bpsk(nstreams=1)
for _ in range(60):
    d2.receive_raw(raw)
