export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r06_m_alltests.log 2>&1; rc=$?
tail -18 gpurun_out/r06_m_alltests.log
[ $rc -eq 0 ] || exit $rc
export JSDR_BENCH_LIVE_TRAFFIC=0
for CASE in "17640 176400 256" "38400 384000 256" "16384 96000 256"; do
set -- $CASE
timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $1 --rate $2 --streams $3 --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/r06_m_acqg_$1.json 2> gpurun_out/r06_m_acqg_$1.err || { tail -5 gpurun_out/r06_m_acqg_$1.err; exit 1; }
python3 - $1 <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r06_m_acqg_{sys.argv[1]}.json') if l.startswith('{')][-1])
print('frame', sys.argv[1], d['ms_per_step'], d['validated'], d['config'].get('streams'), d['roofline'].get('kernels_ms_per_step'))
PY
done
