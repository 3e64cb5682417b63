export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_h_bench.json 2> gpurun_out/r06_h_bench.err; echo rc=$?
tail -3 gpurun_out/r06_h_bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_h_bench.json') if l.startswith('{')][-1])
r=d['roofline']
print(d['ms_per_step'], d['validated'], r.get('valu_issue_frac'), r.get('autotune',{}).get('chosen'), r.get('autotune',{}).get('side_by_side_ms'), r.get('autotune',{}).get('one_after_the_other_ms'), r.get('one_after_the_other_ms_per_step'), r.get('side_by_side_ms_per_step'))
print(json.dumps(r.get('valu_issue'))[:900])
print(r.get('traffic'), r.get('traffic_source','')[:80])
PY
