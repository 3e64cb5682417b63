#!/bin/bash
# tools/ab_env.sh "<bench args>" VAR=a VAR=b ... -- one bench line under several environment settings in ONE session, each twice, interleaved
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
A=$1; shift
for rep in 1 2; do
for KV in "$@"; do
  env $KV timeout -k 10 300 python bench.py $A --no-cpu-baseline > "gpurun_out/abenv_${KV//[^A-Za-z0-9_=]/_}.log" 2>&1
  echo -n "$KV: "; python3 tools/kms.py "gpurun_out/abenv_${KV//[^A-Za-z0-9_=]/_}.log"
done
done
