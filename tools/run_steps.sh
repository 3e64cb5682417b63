#!/bin/bash
# tools/run_steps.sh -- ad-hoc GPU session: each argument is "name|timeout|command" run in order under gpurun_out/;
# a step that times out or faults ends the session (no GPU work after a kill).
#   gpurun -- 'bash tools/run_steps.sh "t|600|python -m pytest tests -m gpu -x -q" "b|300|python bench.py"'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for spec in "$@"; do
  IFS='|' read -r name to cmd <<< "$spec"
  echo "=== $name: $cmd" | tee -a gpurun_out/session.log
  t0=$(date +%s)
  timeout -k 10 "$to" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a gpurun_out/session.log
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/session.log; exit 99; fi
  if grep -q "Memory access fault" "gpurun_out/$name.log"; then echo "GPU FAULT in $name: stopping" | tee -a gpurun_out/session.log; exit 98; fi
done
echo "session done"
