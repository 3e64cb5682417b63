export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r06_f_alltests.log 2>&1; rc=$?
tail -8 gpurun_out/r06_f_alltests.log
exit $rc
