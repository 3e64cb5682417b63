export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_bpsk.py -m gpu -x -q -k "either_front_end or front_end_choice" > gpurun_out/r06_g_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r06_g_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/r06_acq_ab.sh 2048 4096
