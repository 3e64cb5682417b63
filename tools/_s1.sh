set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
for F in "4410 44100" "9600 96000" "4800 48000"; do set -- $F
JSDR_FFT_PHASECLK=1 timeout -k 10 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame $1 --rate $2 --streams 1024 --no-cpu-baseline --steps 2 --warmup 1 --no-validate 2>&1 | grep "phase\|metric" | cut -c1-120 | sed "s/^/$1: /"
done | tee gpurun_out/r05_h_phase_clocks.txt
