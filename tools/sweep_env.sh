#!/bin/bash
# tools/sweep_env.sh VAR "v1 v2 ..." -- bench args...: one bench.py line per value of an environment knob
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
var=$1; vals=$2; shift 3
for v in $vals; do
  env $var=$v python bench.py "$@" > gpurun_out/sweep_${var}_$v.log 2>&1 || { tail -3 gpurun_out/sweep_${var}_$v.log; exit 1; }
  echo "$var=$v: $(python -c "
import json
for l in open('gpurun_out/sweep_${var}_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'])
")"
done
