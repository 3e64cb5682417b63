#!/bin/bash
# tools/prof_wl.sh <tag> "<bench args>" [VAR=val ...] -- rocprofv3 --kernel-trace --stats of one bench line; prints the per-kernel table
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
T=$1; A=$2; shift 2
rm -rf gpurun_out/prof_$T
env "$@" timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -- python3 bench.py $A --no-cpu-baseline > gpurun_out/prof_$T.log 2>&1
f=$(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_${T}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"].split("(")[0][-40:]
    print(f'{n:42s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:9.4f} ms  min {float(r["MinNs"])/1e6:9.4f}')
PY
