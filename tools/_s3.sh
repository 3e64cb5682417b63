export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0 JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1
mkdir -p gpurun_out
JSDR_FFT_PHASECLK=1 timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 2048 --streams 1024 --no-cpu-baseline --no-validate --steps 4 --warmup 1 > gpurun_out/r06_d_clk.log 2>&1
grep "phase\|ms_per_step" gpurun_out/r06_d_clk.log | cut -c1-200 | grep -v metric
