export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_bpsk.py -m gpu -x -q -k "front_end_choice" > gpurun_out/r06_g_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r06_g_tests.log
exit $rc
