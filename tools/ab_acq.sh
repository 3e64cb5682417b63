#!/bin/bash
# tools/ab_acq.sh <frame> <lib...> -- the same FFT-acquire bench line with several builds of the library, in ONE session
# (boxes differ by several percent; only numbers of one session compare), each twice, interleaved
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
F=$1; shift
for rep in 1 2; do
for L in "$@"; do
  JSDR_LIB=$PWD/java-sdr_amd/$L timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $F --streams 1024 --no-cpu-baseline --no-validate --steps 8 --warmup 2 > gpurun_out/ab_$L.log 2>&1
  python3 - "$L" <<'PY'
import json, sys
for l in open(f"gpurun_out/ab_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l); k = d["roofline"]["kernels_ms_per_step"]
        print(sys.argv[1], "step", d["ms_per_step"], {a: b for a, b in k.items() if a.startswith("k_front") or a.startswith("k_acq")})
PY
done
done
