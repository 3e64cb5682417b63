#!/usr/bin/env python3
"""diagnostic: per-step sync-hit counts and FEC return codes for the bench workload (64 streams)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import java_sdr_amd as J
import oracle_lib as O
import bench
S, L = 64, 1048576
d_iq, pay, nfr = bench.make_inputs(J, O, S, L, 0)
dem = J.Bpsk(nstreams=S, max_batch_samples=L)
for step in range(4):
    dem.batch_i16(d_iq, 2 * L, L)
    cnt = [len(dem.fec_results(s)) for s in range(S)]
    rcs = [r[0] for s in range(4) for r in dem.fec_results(s)]
    print("step", step, "hits per stream:", np.bincount(cnt), "rc sample:", rcs, "cntBit0", dem.counters(0)["cntBit"])
