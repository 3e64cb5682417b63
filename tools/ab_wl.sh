#!/bin/bash
# tools/ab_wl.sh "<bench args>" <lib...> -- one bench line with several builds of the library in ONE session, each twice, interleaved
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
A=$1; shift
for rep in 1 2; do
for L in "$@"; do
  JSDR_LIB=$PWD/java-sdr_amd/$L timeout -k 10 300 python bench.py $A --no-cpu-baseline > gpurun_out/ab_$L.log 2>&1
  python3 tools/kms.py gpurun_out/ab_$L.log
done
done
