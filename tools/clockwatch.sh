#!/bin/bash
# tools/clockwatch.sh <log> -- cmd...: sample the GPU's shader clock and power (rocm-smi, read-only) while cmd runs
log=$1; shift
"$@" > "$log" 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed 's/  */ /g'; echo
  sleep 0.5
done
wait $pid
