#!/bin/bash
# tools/acq_counters.sh <tag> [frames...] -- fresh evidence for the FFT-acquire front ends (k_front_fft / k_front_fftm /
# k_front_fft2x) at 1024 streams x 2^20 samples: the bench line, two SQ counter passes, the FETCH_SIZE / WRITE_SIZE
# passes (separate runs, as the guide prescribes) and the per-phase clocks.  Everything lands under gpurun_out/<tag>_*.
#   gpurun --timeout 1100 -- 'bash tools/acq_counters.sh r03_a 2048 9600 19200'
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
set -u
T=${1:-rXX}; shift
FRAMES=${*:-2048 9600 19200}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out
line() { grep '^{"metric' "$1" | tail -1; }
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a $O/session.log
  timeout -k 10 "$to" "$@" > "$O/${T}_$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $O/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/session.log; exit 99; fi
  if grep -q "Memory access fault" "$O/${T}_$name.log"; then echo "GPU FAULT in $name: stopping" | tee -a $O/session.log; exit 98; fi
}
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
for FR in $FRAMES; do  # a frame size, or size@rate (4410@44100)
  F=${FR%@*}; RATE=96000; case $FR in *@*) RATE=${FR#*@};; esac
  ARGS="--workload bpsk --fft-acquire --bpsk-frame $F --rate $RATE --streams 1024 --no-cpu-baseline"
  step acq${F} 300 python bench.py $ARGS --steps 5 --warmup 2
  line $O/${T}_acq${F}.log > $O/${T}_acq${F}.json
  PA="$ARGS --steps 2 --warmup 1 --no-validate"
  JSDR_FFT_PHASECLK=1 step clk_$F 300 python bench.py $PA
  rm -rf $O/${T}_sq1_$F $O/${T}_sq2_$F $O/${T}_rd_$F $O/${T}_wr_$F
  step sq1_$F 300 rocprofv3 --pmc $SQ1 --output-format csv -d $O/${T}_sq1_$F -- python3 bench.py $PA
  python tools/pmc_summary.py $O/${T}_sq1_$F > $O/${T}_sq_counters_acq${F}_1.txt
  step sq2_$F 300 rocprofv3 --pmc $SQ2 --output-format csv -d $O/${T}_sq2_$F -- python3 bench.py $PA
  python tools/pmc_summary.py $O/${T}_sq2_$F > $O/${T}_sq_counters_acq${F}_2.txt
  step rd_$F 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_rd_$F -- python3 bench.py $PA
  step wr_$F 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_wr_$F -- python3 bench.py $PA
  python tools/pmc_traffic.py $O/${T}_rd_$F $O/${T}_wr_$F $O/${T}_pmc_traffic_acq${F}.json $((1024 * (1048576 / F) * F)) > /dev/null
  rm -rf $O/${T}_sq1_$F $O/${T}_sq2_$F $O/${T}_rd_$F $O/${T}_wr_$F
done
echo "acq counters $T done" | tee -a $O/session.log
