set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/sqp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/sqp -- python3 bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-validate > gpurun_out/sqp.log 2>&1
python tools/pmc_summary.py gpurun_out/sqp > gpurun_out/r05_g_sq_counters_pipeline_serial.txt; rm -rf gpurun_out/sqp
grep -A8 "^jsdr\|^void" gpurun_out/r05_g_sq_counters_pipeline_serial.txt | grep "^jsdr\|^void\|INSTS_VALU\|BUSY_CYCLES\|SQ_WAVES " 
