set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== fft alone"; bash tools/ab_env.sh "--workload fft --steps 10 --warmup 3" JSDR_FFT_FORM=0 JSDR_FFT_FORM=1 JSDR_FFT_FORM=2 JSDR_FFT_FORM=3 2>&1 | sed 's/validated.*//' 
for f in 0 1 2 3; do grep -o '"ms_per_step": [0-9.]*' gpurun_out/abenv_JSDR_FFT_FORM=$f.log | tail -1 | sed "s/^/form $f /"; done
echo "== pipeline serial"; bash tools/ab_env.sh "--serial --steps 8 --warmup 3" JSDR_FFT_FORM=0 JSDR_FFT_FORM=2
