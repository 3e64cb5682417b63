#!/bin/bash
# tools/ab_args.sh "<common bench args>" "<args A>" "<args B>" ... -- one bench line under several argument sets in ONE session, each twice, interleaved
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
C=$1; shift
for rep in 1 2; do
i=0
for A in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py $C $A --no-cpu-baseline > "gpurun_out/abargs_$i.log" 2>&1
  echo -n "[$A] "; python3 tools/kms.py "gpurun_out/abargs_$i.log"
done
done
