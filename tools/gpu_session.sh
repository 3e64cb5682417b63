#!/bin/bash
# tools/gpu_session.sh -- what one gpurun call runs on the MI355X box.  Every step writes under
# gpurun_out/; a step that TIMES OUT (124/137) ends the session (no further GPU work after a kill),
# an ordinary failure (assert, non-zero exit) is logged and the session goes on.
#   usage: tools/gpu_session.sh step [step ...]   steps: tests micro bench_small bench bench_fft bench_bpsk prof pmc
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND:-r01}
run() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/session.log
  local t0=$(date +%s)
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a gpurun_out/session.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping session" | tee -a gpurun_out/session.log; exit 99; fi
  if grep -q "Memory access fault" "gpurun_out/$name.log"; then echo "GPU FAULT in $name: stopping session" | tee -a gpurun_out/session.log; exit 98; fi
  return 0
}
for step in "$@"; do
  case $step in
    tests)       run tests 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --timeout 600 ;;
    tests_x)     run tests 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider --timeout 600 ;;
    smoke)       run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    micro)       run micro_build 120 hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mb_fp64 tools/microbench_fp64.hip
                 run micro 120 /tmp/mb_fp64 ;;
    bench_small) run bench_small 300 python bench.py --steps 3 --warmup 1 --streams 128 --samples 1048576 --cpu-seconds 3 ;;
    bench)       JSDR_BENCH_LIVE_TRAFFIC=1 run bench 900 python bench.py ;;
    tests_demod) run tests_demod 600 python -m pytest tests/test_gpu_demod.py -m gpu -q -x -p no:cacheprovider --timeout 600 ;;
    tests_fft)   run tests_fft 900 python -m pytest tests/test_gpu_fft.py -m gpu -q -x -p no:cacheprovider --timeout 800 ;;
    tests_misc)  run tests_misc 600 python -m pytest tests/test_gpu_fir_phase_fec.py tests/test_gpu_host.py -m gpu -q -x -p no:cacheprovider --timeout 600 ;;
    tests_fmt)   run tests_fmt 600 python -m pytest tests/test_gpu_formats.py -m gpu -q -x -p no:cacheprovider --timeout 600 ;;
    tests_host)  run tests_host 600 python -m pytest tests/test_gpu_host.py -m gpu -q -x -p no:cacheprovider --timeout 600 ;;
    tests_bpsk)  run tests_bpsk 600 python -m pytest tests/test_gpu_bpsk.py -m gpu -q -x -p no:cacheprovider --timeout 600 ;;
    bench_quick) run bench_quick 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_acq)   run bench_acq 400 python bench.py --workload bpsk --fft-acquire --steps 3 --warmup 1 --no-cpu-baseline ;;
    bench_acq9600) run bench_acq9600 400 python bench.py --workload bpsk --fft-acquire --bpsk-frame 9600 --steps 3 --warmup 1 --no-cpu-baseline ;;
    bench_acq_clk) JSDR_FFT_PHASECLK=1 run bench_acq_clk 400 python bench.py --workload bpsk --fft-acquire --steps 2 --warmup 1 --no-cpu-baseline ;;
    bench_n2)    JSDR_BENCH_SAME_DEVICE=1 JSDR_BENCH_BACKEND=gloo run bench_n2 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --streams 256 ;;
    bench_reg_alone) JSDR_NO_OVERLAP=1 run bench_reg_alone 300 python bench.py --workload bpsk --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_reg) run bench_reg 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_m64) JSDR_FFT_GRID_MULT=64 run bench_m64 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_m256) JSDR_FFT_GRID_MULT=256 run bench_m256 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_m16) JSDR_FFT_GRID_MULT=16 run bench_m16 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_alone_b) JSDR_LIB=$PWD/java-sdr_amd/libjsdr_hip_b.so JSDR_NO_OVERLAP=1 run bench_alone_b 300 python bench.py --workload bpsk --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_alone) JSDR_NO_OVERLAP=1 run bench_alone 300 python bench.py --workload bpsk --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_noov)  JSDR_NO_OVERLAP=1 run bench_noov 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    fec_bench)   run fec_bench 300 python tests/tools/fec_bench.py ;;
    hbm)         run hbm_build 120 hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_hbm tools/microbench_hbm.hip
                 run hbm 120 /tmp/mb_hbm ;;
    counters)    run counters 120 rocprofv3 -L ;;
    pmc_sq1)     rm -rf gpurun_out/pmc_sq1_$R
                 run pmc_sq1 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq1_$R -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate ;;
    pmc_sq2)     rm -rf gpurun_out/pmc_sq2_$R
                 run pmc_sq2 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_sq2_$R -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate ;;
    pmc_fft1)    rm -rf gpurun_out/pmc_fft1_$R
                 run pmc_fft1 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_fft1_$R -- python3 bench.py --workload fft --steps 2 --warmup 1 --no-cpu-baseline --no-validate
                 python tools/pmc_summary.py gpurun_out/pmc_fft1_$R | tee gpurun_out/pmc_fft1_summary.txt ;;
    pmc_fft2)    rm -rf gpurun_out/pmc_fft2_$R
                 run pmc_fft2 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_fft2_$R -- python3 bench.py --workload fft --steps 2 --warmup 1 --no-cpu-baseline --no-validate
                 python tools/pmc_summary.py gpurun_out/pmc_fft2_$R | tee gpurun_out/pmc_fft2_summary.txt ;;
    pmc_dm1)     rm -rf gpurun_out/pmc_dm1_$R
                 run pmc_dm1 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_dm1_$R -- python3 bench.py --workload demod --demod-mode nfm --steps 2 --warmup 1 --no-cpu-baseline
                 python tools/pmc_summary.py gpurun_out/pmc_dm1_$R | tee gpurun_out/pmc_dm1_summary.txt ;;
    pmc_dm2)     rm -rf gpurun_out/pmc_dm2_$R
                 run pmc_dm2 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_dm2_$R -- python3 bench.py --workload demod --demod-mode nfm --steps 2 --warmup 1 --no-cpu-baseline
                 python tools/pmc_summary.py gpurun_out/pmc_dm2_$R | tee gpurun_out/pmc_dm2_summary.txt ;;
    pmc_acq1)    rm -rf gpurun_out/pmc_acq1_$R
                 run pmc_acq1 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_acq1_$R -- python3 bench.py --workload bpsk --fft-acquire --steps 2 --warmup 1 --no-cpu-baseline --no-validate
                 python tools/pmc_summary.py gpurun_out/pmc_acq1_$R | tee gpurun_out/pmc_acq1_summary.txt ;;
    pmc_acq2)    rm -rf gpurun_out/pmc_acq2_$R
                 run pmc_acq2 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_acq2_$R -- python3 bench.py --workload bpsk --fft-acquire --steps 2 --warmup 1 --no-cpu-baseline --no-validate
                 python tools/pmc_summary.py gpurun_out/pmc_acq2_$R | tee gpurun_out/pmc_acq2_summary.txt ;;
    pmc_fm1)     rm -rf gpurun_out/pmc_fm1_$R
                 run pmc_fm1 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_fm1_$R -- python3 tools/fftm_bench.py
                 python tools/pmc_summary.py gpurun_out/pmc_fm1_$R | tee gpurun_out/pmc_fm1_summary.txt ;;
    pmc_fm2)     rm -rf gpurun_out/pmc_fm2_$R
                 run pmc_fm2 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_fm2_$R -- python3 tools/fftm_bench.py
                 python tools/pmc_summary.py gpurun_out/pmc_fm2_$R | tee gpurun_out/pmc_fm2_summary.txt ;;
    dbg)         run dbg 300 python tests/tools/dbg_fftmode.py ;;
    fftm_bench)  run fftm_bench 300 python tools/fftm_bench.py ;;
    rate192)     JSDR_NO_OVERLAP=1 run rate192 300 python tools/rate_bench.py 192000 ;;
    rate48)      JSDR_NO_OVERLAP=1 run rate48 300 python tools/rate_bench.py 48000 ;;
    rate44)      JSDR_NO_OVERLAP=1 run rate44 300 python tools/rate_bench.py 44100 ;;
    fft9600)     run fft9600 300 python tools/fft_n_bench.py 9600 ;;
    fft19200)    run fft19200 300 python tools/fft_n_bench.py 19200 ;;
    fft4800)     run fft4800 300 python tools/fft_n_bench.py 4800 ;;
    fftm_small)  FM_S=8 FM_NFR=20 run fftm_small 300 python tools/fftm_bench.py ;;
    dbg_demod)   run dbg_demod 300 python tests/tools/dbg_demod.py ;;
    trig)        run trig 300 python tests/tools/trig_stats.py ;;
    bench_fft)   run bench_fft 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_fft_m16) JSDR_FFT_GRID_MULT=16 run bench_fft_m16 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_fft_m64) JSDR_FFT_GRID_MULT=64 run bench_fft_m64 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_fft_m1k) JSDR_FFT_GRID_MULT=1024 run bench_fft_m1k 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_fft_m1) JSDR_FFT_GRID_MULT=1 run bench_fft_m1 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_demod) run bench_demod 400 python bench.py --workload demod --steps 5 --warmup 2 --cpu-seconds 4 ;;
    bench_demod_fm) run bench_demod_fm 400 python bench.py --workload demod --demod-mode nfm --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_fft_wf) run bench_fft_wf 300 python bench.py --workload fft --waterfall-width 1024 --no-cpu-baseline ;;
    bench_fft_b) JSDR_LIB=$PWD/java-sdr_amd/libjsdr_hip_b.so run bench_fft_b 300 python bench.py --workload fft --no-cpu-baseline ;;
    bench_quick_b) JSDR_LIB=$PWD/java-sdr_amd/libjsdr_hip_b.so run bench_quick_b 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    bench_bpsk)  run bench_bpsk 400 python bench.py --workload bpsk --no-cpu-baseline ;;
    prof)        rm -rf gpurun_out/prof_$R
                 run prof 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline ;;
    pmc_rd)      rm -rf gpurun_out/pmc_rd_$R
                 run pmc_rd 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_rd_$R -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate ;;
    pmc_wr)      rm -rf gpurun_out/pmc_wr_$R
                 run pmc_wr 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_wr_$R -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate ;;
    *) echo "unknown step $step" ;;
  esac
done
echo "session done" | tee -a gpurun_out/session.log
