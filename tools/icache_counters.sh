#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQ_INST_CYCLES|SQ_WAIT_IFETCH|SQC_" | head -40 > $O/icache_counters.txt
for F in 9600 19200; do
PA="--workload bpsk --fft-acquire --bpsk-frame $F --streams 1024 --no-cpu-baseline --steps 2 --warmup 1 --no-validate"
rm -rf $O/ic_$F
timeout -k 10 300 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/ic_$F -- python3 bench.py $PA > $O/ic_$F.log 2>&1
python tools/pmc_summary.py $O/ic_$F > $O/ic_summary_$F.txt 2>&1
rm -rf $O/ic_$F
done
echo done
