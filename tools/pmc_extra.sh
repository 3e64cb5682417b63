#!/bin/bash
# tools/pmc_extra.sh <tag> -- FETCH_SIZE / WRITE_SIZE passes (separate runs) for the workloads whose dominant kernel the
# default pipeline does not launch: --workload fir (k_fir_batch) -> gpurun_out/<tag>_pmc_traffic_fir.json
set -u
T=${1:-rXX}
mkdir -p gpurun_out; export TMPDIR=/tmp
O=$PWD/gpurun_out
A="--workload fir --steps 2 --warmup 1 --no-cpu-baseline --no-validate"
rm -rf $O/${T}_rd_fir $O/${T}_wr_fir
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_rd_fir -- python3 bench.py $A > $O/${T}_rd_fir.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_wr_fir -- python3 bench.py $A > $O/${T}_wr_fir.log 2>&1 || exit 1
python tools/pmc_traffic.py $O/${T}_rd_fir $O/${T}_wr_fir $O/${T}_pmc_traffic_fir.json 8589934592
rm -rf $O/${T}_rd_fir $O/${T}_wr_fir
