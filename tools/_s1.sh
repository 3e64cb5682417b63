set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fft.py -m gpu -q -x -p no:cacheprovider --timeout 600 > gpurun_out/r05_e_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_e_tests.log
for n in 9600 4410 2205 3200 800 8820 1102; do timeout -k 10 120 python tools/fft_n_bench.py $n 1024 2>&1 | tail -1; done | tee gpurun_out/r05_e_fft_n_bench.txt
python tools/latency_bench.py 2>&1 | tail -12
