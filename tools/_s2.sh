export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bpsk.py tests/test_gpu_fixtures.py -m gpu -x -q -k "fft and not either_front_end" > gpurun_out/r06_b_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r06_b_tests.log
[ $rc -eq 0 ] || exit $rc
bash tools/_s3.sh
bash tools/r06_acq_ab.sh 2048
