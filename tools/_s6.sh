export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fixtures.py -m gpu -x -q -k "fast" > gpurun_out/r06_g_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r06_g_tests.log
exit $rc
