set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_headline_mode.py tests/test_gpu_fixtures.py tests/test_gpu_bench_launcher.py tests/test_gpu_bpsk.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/r05_i_tests.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r05_i_tests.log
