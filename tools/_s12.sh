export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_bpsk.py -m gpu -x -q -k "any_frame" > gpurun_out/r06_l_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r06_l_tests.log
[ $rc -eq 0 ] || exit $rc
export JSDR_BENCH_LIVE_TRAFFIC=0
timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 16384 --rate 96000 --streams 256 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r06_m_acqg_16384.json 2> gpurun_out/r06_m_acqg_16384.err || { tail -5 gpurun_out/r06_m_acqg_16384.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_m_acqg_16384.json') if l.startswith('{')][-1])
print('frame 16384', d['ms_per_step'], d['validated'], d['roofline'].get('kernels_ms_per_step'))
PY
