# tools/ab_acq_sidestream.sh -- the FFT-acquire step with the tail / sync / FEC section on the side stream (default where the library chooses it) against JSDR_NO_OVERLAP=1, one session (profiles/r06_experiments.md 1c)
export TMPDIR=/tmp JSDR_BENCH_LIVE_TRAFFIC=0 JSDR_KNOBS=1 JSDR_BENCH_ALLOW_KNOBS=1
mkdir -p gpurun_out
for CASE in "2048 1024" "4096 1024" "2048 64" "1024 1024" "2048 8192"; do
set -- $CASE
for OV in default 1; do
  if [ $OV = default ]; then unset JSDR_NO_OVERLAP; else export JSDR_NO_OVERLAP=1; fi
  timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $1 --streams $2 --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/ab_ov.log 2>&1
  python3 - $OV $1 $2 <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/ab_ov.log') if l.startswith('{')][-1])
print('frame', sys.argv[2], 'streams', sys.argv[3], 'NO_OVERLAP', sys.argv[1], d['ms_per_step'], d['validated'])
PY
done
done
