export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0
mkdir -p gpurun_out
for W in pipeline bpsk; do
timeout -k 10 600 python bench.py --workload $W --variant fast --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_i_fast_$W.json 2> gpurun_out/r06_i_fast_$W.err; echo rc=$?
tail -2 gpurun_out/r06_i_fast_$W.err
python3 - $W <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r06_i_fast_{sys.argv[1]}.json') if l.startswith('{')][-1])
print(sys.argv[1], d['ms_per_step'], d['validated'], d.get('certification'), {k:v for k,v in (d.get('validation') or {}).items() if 'oracle' in k or 'differ' in k or 'recovered' in k})
PY
done
timeout -k 10 900 python bench.py --variant fast --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r06_i_fast_soak200.json 2> gpurun_out/r06_i_fast_soak200.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_i_fast_soak200.json') if l.startswith('{')][-1])
print('soak', d['ms_per_step'], d['validated'], d.get('certification'), {k:v for k,v in (d.get('validation') or {}).items() if 'oracle' in k or 'differ' in k or 'recovered' in k})
PY
