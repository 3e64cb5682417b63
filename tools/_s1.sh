set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_fft.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "acquire_mode or random_smooth" 2>&1 | tail -15
