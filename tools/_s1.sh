set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_m_bench.log 2>&1; echo "bench rc=$?"; python3 tools/kms.py gpurun_out/r05_m_bench.log; grep -o '"one_after_the_other_ms_per_step": [0-9.]*\|"kernel": "[^"]*"' gpurun_out/r05_m_bench.log
timeout -k 10 300 python bench.py --streams 1024 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_m_b1k.log 2>&1; python3 tools/kms.py gpurun_out/r05_m_b1k.log; grep -o '"kernel": "[^"]*"' gpurun_out/r05_m_b1k.log
timeout -k 10 300 java-sdr_amd/host/jsdr_harness --gpus 1 --streams 8192 --psd --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_cu_share.py tests/test_gpu_headline_mode.py -m gpu -q -p no:cacheprovider --timeout 600 2>&1 | tail -2
