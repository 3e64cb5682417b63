#!/bin/bash
# tools/timeline.sh <tag> "<bench args>" [VAR=val ...] -- rocprofv3 --kernel-trace of one bench line; prints when every kernel of the
# last two steps started and ended (which kernels really run beside which)
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
T=$1; A=$2; shift 2
rm -rf gpurun_out/tl_$T
env "$@" timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_$T -- python3 bench.py $A --no-cpu-baseline --no-validate > gpurun_out/tl_$T.log 2>&1
f=$(find gpurun_out/tl_$T -name "*kernel_trace.csv" | head -1)
python3 - "$f" > gpurun_out/${T}_timeline.txt <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "synth" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
ffts=[i for i,r in enumerate(rows) if "k_fft" in r["Kernel_Name"]]
i0=ffts[-2] if len(ffts)>=2 else 0
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%9.3f ms .. %9.3f ms  (%8.3f)  %s"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,r["Kernel_Name"].split("(")[0][-44:]))
PY
rm -rf gpurun_out/tl_$T
cat gpurun_out/${T}_timeline.txt
