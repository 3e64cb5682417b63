#!/bin/bash
# tools/r06_baseline.sh -- round 6, first session: the ADVICE-fix tests, then the acquire front ends as round 5 left them (the "before" of the three-phase rewrite)
export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_group.py tests/test_gpu_formats.py -m gpu -x -q > gpurun_out/r06_a_tests.log 2>&1 || { tail -30 gpurun_out/r06_a_tests.log; exit 1; }
tail -3 gpurun_out/r06_a_tests.log
for F in 2048 9600 4800; do
  timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $F --streams 1024 --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r06_a_acq$F.json 2> gpurun_out/r06_a_acq$F.err || { tail -5 gpurun_out/r06_a_acq$F.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r06_a_acq$F.json') if l.startswith('{')][-1])
print($F, d['ms_per_step'], d.get('validated'), {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 4410 --rate 44100 --streams 1024 --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/r06_a_acq4410.json 2> gpurun_out/r06_a_acq4410.err && python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r06_a_acq4410.json') if l.startswith('{')][-1])
print(4410, d['ms_per_step'], d.get('validated'), {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
