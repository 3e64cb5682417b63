set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_bpsk.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "other_decimations or consumer_sound or rejects_partial or api_errors or any_audio_rate or other_mixed" > gpurun_out/r05_d_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_d_tests.log
