set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_group.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/r05_c_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_c_tests.log
timeout -k 10 300 java-sdr_amd/host/jsdr_harness --gpus 1 --streams 8192 --psd --steps 10 --warmup 3 > gpurun_out/r05_c_harness_n1.json 2>gpurun_out/r05_c_harness_n1.err; echo "harness rc=$?"; cat gpurun_out/r05_c_harness_n1.json; tail -3 gpurun_out/r05_c_harness_n1.err
