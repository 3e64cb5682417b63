#!/bin/bash
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
export TMPDIR=/tmp
for F in 9600 4800 19200; do
for OV in 0 1; do
  JSDR_NO_OVERLAP=$OV timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $F --streams 1024 --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ov.log 2>&1
  python3 - $F $OV <<'PY'
import json, sys
for l in open("gpurun_out/ov.log"):
    if l.startswith("{"):
        d = json.loads(l); print("frame", sys.argv[1], "no_overlap", sys.argv[2], "step", d["ms_per_step"], "validated", d.get("validated"))
PY
done
done
JSDR_NO_OVERLAP=0 timeout -k 10 600 python bench.py --workload bpsk --fft-acquire --bpsk-frame 9600 --streams 8192 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/ov.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/ov.log
JSDR_NO_OVERLAP=1 timeout -k 10 600 python bench.py --workload bpsk --fft-acquire --bpsk-frame 9600 --streams 8192 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/ov.log 2>&1; grep -o '"ms_per_step": [0-9.]*' gpurun_out/ov.log
