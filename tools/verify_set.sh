#!/bin/bash
# tools/verify_set.sh <tag> -- GPU tests, then the bench lines whose fields changed this round (default pipeline, config 2 on
# tone frames, config 3 validated).  Everything under gpurun_out/<tag>_*.
set -u
T=${1:-rXX}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out
line() { grep '^{"metric' "$1" | tail -1; }
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a $O/session.log
  timeout -k 10 "$to" "$@" > "$O/${T}_$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $O/session.log
  tail -n 6 "$O/${T}_$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/session.log; exit 99; fi
  if grep -q "Memory access fault" "$O/${T}_$name.log"; then echo "GPU FAULT in $name: stopping" | tee -a $O/session.log; exit 98; fi
}
step tests 1000 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider --timeout 600
B="--no-cpu-baseline"
step b_fft 300 python bench.py --workload fft $B;   line $O/${T}_b_fft.log > $O/${T}_b_fft.json
step b_fir 300 python bench.py --workload fir $B;   line $O/${T}_b_fir.log > $O/${T}_b_fir.json
step b_fir65 300 python bench.py --workload fir --fir-taps 65 --streams 1024 $B;   line $O/${T}_b_fir65.log > $O/${T}_b_fir65.json
step bench 400 python bench.py --steps 20 --warmup 5; line $O/${T}_bench.log > $O/${T}_bench.json
echo "verify set $T done" | tee -a $O/session.log
