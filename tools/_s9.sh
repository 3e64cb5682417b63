export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_headline_mode.py tests/test_gpu_cu_share.py -m gpu -x -q > gpurun_out/r06_j_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r06_j_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_wl.sh "--workload fft --steps 10 --warmup 3 --no-validate" libjsdr_hip.so libx_FFTSYNC.so
bash tools/ab_wl.sh "--steps 10 --warmup 3 --no-validate --no-autotune --no-compare-serial" libjsdr_hip.so libx_FFTSYNC.so
bash tools/ab_wl.sh "--steps 10 --warmup 3 --no-validate --serial" libjsdr_hip.so libx_FFTSYNC.so
