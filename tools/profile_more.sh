#!/bin/bash
# tools/profile_more.sh <tag> -- rocprofv3 --kernel-trace --stats of the other workloads' bench lines (fft, fir, bpsk, the three
# FFT-acquire frame sizes) and a 200-step soak of the default pipeline; summaries under gpurun_out/<tag>_*
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
set -u
T=${1:-rXX}
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$PWD/gpurun_out
line() { grep '^{"metric' "$1" | tail -1; }
stats() { # name args...
  local name=$1; shift
  rm -rf $O/prof_${T}_$name
  echo "=== $name" | tee -a $O/session.log
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_$name -- python3 bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $O/${T}_prof_$name.log 2>&1
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit 99; fi
  cp "$(find $O/prof_${T}_$name -name '*kernel_stats.csv' | head -1)" $O/${T}_kernel_stats_$name.csv
  line $O/${T}_prof_$name.log > $O/${T}_under_rocprof_$name.json
  rm -rf $O/prof_${T}_$name
}
stats fft --workload fft
stats fir --workload fir
stats bpsk --workload bpsk
stats acq2048 --workload bpsk --fft-acquire --streams 1024
stats acq9600 --workload bpsk --fft-acquire --bpsk-frame 9600 --streams 1024
stats acq19200 --workload bpsk --fft-acquire --bpsk-frame 19200 --streams 1024
echo "=== soak" | tee -a $O/session.log
timeout -k 10 600 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > $O/${T}_soak_200steps.log 2>&1
line $O/${T}_soak_200steps.log > $O/${T}_soak_exact_200steps.json
echo "profile_more $T done"
