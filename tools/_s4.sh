export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0 JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1
mkdir -p gpurun_out
JSDR_LIB=$PWD/java-sdr_amd/libx_FMLDS.so timeout -k 10 600 python -m pytest tests/test_gpu_bpsk.py -m gpu -x -q -k "fft" > gpurun_out/r06_e_tests.log 2>&1; tail -3 gpurun_out/r06_e_tests.log
for F in 9600 4800; do bash tools/ab_acq.sh $F libjsdr_hip.so libx_FMLDS.so; done
