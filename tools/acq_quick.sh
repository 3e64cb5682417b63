#!/bin/bash
# tools/acq_quick.sh [frames...] -- the FFT-acquire GPU tests, then one short bench line per frame size with the phase clocks
export JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1 JSDR_BENCH_LIVE_TRAFFIC=0  # the library listens to its tuning knobs only with JSDR_KNOBS=1; bench.py measures with them only when told
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_bpsk.py tests/test_gpu_fixtures.py -m gpu -q -x -p no:cacheprovider --timeout 600 -k "fft or acq or mixed or 9600 or 4800 or 19200" > gpurun_out/t_acq.log 2>&1
tail -3 gpurun_out/t_acq.log
for F in ${*:-9600}; do
  JSDR_FFT_PHASECLK=1 timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $F --streams 1024 --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/acq_$F.log 2>&1
  grep -h "phase" gpurun_out/acq_$F.log
  python3 - "$F" <<'PY'
import json, sys
for l in open(f"gpurun_out/acq_{sys.argv[1]}.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernels_ms_per_step"], d["validated"])
PY
done
