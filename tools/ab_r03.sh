#!/bin/bash
# tools/ab_r03.sh <tag> -- the default bench line of the round-3 tree (.r03tree, a scratch export of that round's last commit with
# its own built library) and of this tree, three times each, interleaved, on ONE box: boxes differ by ~1 ms on this line
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
set -u
T=${1:-rXX}; O=$PWD/gpurun_out; mkdir -p $O
line() { grep '^{"metric' "$1" | tail -1; }
: > $O/${T}_ab_r03.txt
for i in 1 2 3; do
  for w in r03 new; do
    d=$PWD; [ $w = r03 ] && d=$PWD/.r03tree
    ( cd $d && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${T}_ab_${w}_$i.log 2>&1 )
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT $w $i"; exit 99; fi
    python3 - "$w" "$i" "$O/${T}_ab_${w}_$i.log" >> $O/${T}_ab_r03.txt <<'PY'
import json,sys
w,i,p=sys.argv[1:4]
l=[x for x in open(p) if x.startswith('{"metric')]
if not l: print(w,i,"NO LINE"); sys.exit(0)
o=json.loads(l[-1]); pk=o["roofline"]["per_kernel"]
print(w,i,"ms_per_step=%.3f"%o["ms_per_step"],"validated=%s"%o.get("validated")," ".join("%s=%.2f"%(k,v.get("avg_launch_ms",0)) for k,v in pk.items() if isinstance(v,dict)))
PY
  done
done
cat $O/${T}_ab_r03.txt
