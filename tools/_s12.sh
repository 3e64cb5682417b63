export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_bpsk.py -m gpu -x -q -k "any_frame or rejects or api_errors" > gpurun_out/r06_l_tests.log 2>&1; rc=$?
tail -30 gpurun_out/r06_l_tests.log
exit $rc
