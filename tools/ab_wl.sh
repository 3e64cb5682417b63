#!/bin/bash
# tools/ab_wl.sh "<bench args>" <lib...> -- one bench line with several builds of the library in ONE session, each twice, interleaved
mkdir -p gpurun_out; export TMPDIR=/tmp
A=$1; shift
for rep in 1 2; do
for L in "$@"; do
  JSDR_LIB=$PWD/java-sdr_amd/$L timeout -k 10 300 python bench.py $A --no-cpu-baseline > gpurun_out/ab_$L.log 2>&1
  python3 tools/kms.py gpurun_out/ab_$L.log
done
done
