#!/bin/bash
# tools/r06_acq_ab.sh [frames...] -- FFT-acquire front end: round 5's fused kernel (JSDR_ACQ3=0) against the three-phase form, one session
export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0 JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1
mkdir -p gpurun_out
for FR in "${@:-2048}"; do
F=${FR%@*}; RATE=96000; case $FR in *@*) RATE=${FR#*@};; esac
for rep in 1 2; do
for M in 0 1; do
  JSDR_ACQ3=$M timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $F --rate $RATE --streams ${STREAMS:-1024} --no-cpu-baseline --no-validate --steps 8 --warmup 2 > gpurun_out/ab_acq3_$M.log 2>&1 || { tail -5 gpurun_out/ab_acq3_$M.log; exit 1; }
  python3 - $M $F <<'PY'
import json, sys
for l in open(f"gpurun_out/ab_acq3_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l); k = d["roofline"]["kernels_ms_per_step"]
        print("frame", sys.argv[2], "ACQ3=" + sys.argv[1], "step", d["ms_per_step"], {a: round(b, 3) for a, b in k.items() if a.startswith("k_front") or a.startswith("k_acq")})
PY
done
done
done
