#!/bin/bash
# tools/fm_counters.sh <tag> -- SQ counters of k_fm ALONE (--workload bpsk, no side stream), exact and fast variant, on
# the current code: three passes each (issue, waits, instruction mix).  Summaries land in gpurun_out/<tag>_sq_counters_k_fm_*.
#   gpurun --timeout 1100 -- 'bash tools/fm_counters.sh r05_a'
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
set -u
T=${1:-rXX}
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$PWD/gpurun_out
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a $O/session.log
  timeout -k 10 "$to" "$@" > "$O/${T}_$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a $O/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name" | tee -a $O/session.log; exit 99; fi
  if grep -q "Memory access fault" "$O/${T}_$name.log"; then echo "GPU FAULT in $name" | tee -a $O/session.log; exit 98; fi; }
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ3="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
SQ4="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_CYCLES"
export JSDR_NO_OVERLAP=1
for V in exact fast; do
  A="--workload bpsk --variant $V --no-cpu-baseline"
  step bpsk_$V 300 python bench.py $A --steps 5 --warmup 2
  grep '^{"metric' $O/${T}_bpsk_$V.log | tail -1 > $O/${T}_b_bpsk_alone_$V.json
  i=1
  for SQ in "$SQ1" "$SQ2" "$SQ3" "$SQ4"; do
    rm -rf $O/${T}_sq_$V$i
    step sq${i}_$V 300 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}_sq_$V$i -- python3 bench.py $A --steps 2 --warmup 1 --no-validate
    python tools/pmc_summary.py $O/${T}_sq_$V$i > $O/${T}_sq_counters_k_fm_${V}_$i.txt
    rm -rf $O/${T}_sq_$V$i
    i=$((i+1))
  done
done
echo "fm counters $T done" | tee -a $O/session.log
