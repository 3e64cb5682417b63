#!/bin/bash
# tools/profile_set.sh <tag> -- one gpurun call that takes the judged evidence set on the current code and leaves
# ready-to-commit summaries under gpurun_out/<tag>_*: the default bench line, rocprofv3 kernel stats of the same
# command, the two separate PMC passes (FETCH_SIZE, WRITE_SIZE -> pmc_traffic.json), and the other workloads' lines.
#   gpurun --timeout 1100 -- 'bash tools/profile_set.sh r02_j'       then:  cp gpurun_out/r02_j_* profiles/
set -u
T=${1:-rXX}
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out
line() { grep '^{"metric' "$1" | tail -1; }
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a $O/session.log
  timeout -k 10 "$to" "$@" > "$O/${T}_$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a $O/session.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/session.log; exit 99; fi
  if grep -q "Memory access fault" "$O/${T}_$name.log"; then echo "GPU FAULT in $name: stopping" | tee -a $O/session.log; exit 98; fi
}
step bench 500 python bench.py --steps 20 --warmup 5
line $O/${T}_bench.log > $O/${T}_bench.json
rm -rf $O/prof_$T $O/pmc_rd_$T $O/pmc_wr_$T
step prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-compare-serial --no-autotune
line $O/${T}_prof.log > $O/${T}_bench_under_rocprof.json
cp "$(find $O/prof_$T -name '*kernel_stats.csv' | head -1)" $O/${T}_kernel_stats.csv
step pmc_rd 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_rd_$T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate --no-compare-serial --no-autotune
step pmc_wr 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_wr_$T -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-validate --no-compare-serial --no-autotune
python tools/pmc_traffic.py $O/pmc_rd_$T $O/pmc_wr_$T $O/${T}_pmc_traffic.json 8589934592 > /dev/null
python - "$O" "$T" <<'PY'
import collections, csv, glob, sys
O, T = sys.argv[1], sys.argv[2]
for tag, ctr in (("rd", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_{tag}_{T}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    with open(f"{O}/{T}_pmc_{tag}.csv", "w") as o:
        o.write("kernel,counter,launches,mean_value_KiB\n")
        for k in sorted(acc):
            o.write(f'"{k}",{ctr},{len(acc[k])},{sum(acc[k]) / len(acc[k]):.1f}\n')
PY
B="--no-cpu-baseline"
# the pipeline the old way: PSD then demodulator on one stream, each kernel with the whole chip (per-kernel times not stretched by the other)
step b_serial 300 python bench.py --serial --steps 20 --warmup 5 $B;     line $O/${T}_b_serial.log > $O/${T}_b_serial.json
step b_bpsk 300 python bench.py --workload bpsk $B;                      line $O/${T}_b_bpsk.log > $O/${T}_b_bpsk.json
step b_bpsk_fast 300 python bench.py --workload bpsk --variant fast $B;  line $O/${T}_b_bpsk_fast.log > $O/${T}_b_bpsk_fast.json
step b_pipe_fast 300 python bench.py --variant fast $B;                  line $O/${T}_b_pipe_fast.log > $O/${T}_b_pipe_fast.json
step b_fft 300 python bench.py --workload fft $B;                        line $O/${T}_b_fft.log > $O/${T}_b_fft.json
step b_fir 300 python bench.py --workload fir $B;                        line $O/${T}_b_fir.log > $O/${T}_b_fir.json
step b_1k 300 python bench.py --streams 1024 $B;                         line $O/${T}_b_1k.log > $O/${T}_b_1k.json
step b_acq 300 python bench.py --workload bpsk --fft-acquire --streams 1024 $B;                    line $O/${T}_b_acq.log > $O/${T}_b_acq.json
step b_acq9600 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 9600 --streams 1024 $B;  line $O/${T}_b_acq9600.log > $O/${T}_b_acq9600.json
step b_acq19200 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 19200 --streams 1024 $B;  line $O/${T}_b_acq19200.log > $O/${T}_b_acq19200.json
step b_acq4800 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 4800 --streams 1024 $B;  line $O/${T}_b_acq4800.log > $O/${T}_b_acq4800.json
# the FUNcube Dongle Pro+ default: 192 kHz, frames of 19200 (JavaAudio.java:59)
step b_acq19200_192k 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 19200 --rate 192000 --streams 1024 $B;  line $O/${T}_b_acq19200_192k.log > $O/${T}_b_acq19200_192k.json
# a 44.1 kHz sound card: frames of 4410 = 2 3^2 5 7^2 samples (radix-7 passes; checked against the oracle, not by payload)
step b_acq4410_44k 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 4410 --rate 44100 --streams 1024 $B;  line $O/${T}_b_acq4410_44k.log > $O/${T}_b_acq4410_44k.json
# frames no LDS front end takes (round 6: the any-frame passes in global memory): a 176.4 kHz and a 384 kHz card's, 256 streams
step b_acqg17640 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 17640 --rate 176400 --streams 256 --steps 4 --warmup 1 $B;  line $O/${T}_b_acqg17640.log > $O/${T}_b_acqg17640.json
step b_acqg38400 300 python bench.py --workload bpsk --fft-acquire --bpsk-frame 38400 --rate 384000 --streams 256 --steps 4 --warmup 1 $B;  line $O/${T}_b_acqg38400.log > $O/${T}_b_acqg38400.json
# FFT-acquire at the reference's default frame, all 8192 streams, every stream validated by payload
step b_acq9600_8k 500 python bench.py --workload bpsk --fft-acquire --bpsk-frame 9600 $B;  line $O/${T}_b_acq9600_8k.log > $O/${T}_b_acq9600_8k.json
step b_fir65 300 python bench.py --workload fir --fir-taps 65 --streams 1024 $B;  line $O/${T}_b_fir65.log > $O/${T}_b_fir65.json
step b_fir65_1 300 python bench.py --workload fir --fir-taps 65 --fir-decim 1 --streams 1024 $B;  line $O/${T}_b_fir65_1.log > $O/${T}_b_fir65_1.json
# the N > 1 path from the bare command on this one-GPU box: two ranks on device 0, gloo in RCCL's place (rehearsal knobs)
JSDR_BENCH_SAME_DEVICE=1 JSDR_BENCH_BACKEND=gloo step b_n2_rehearsal 400 python bench.py --gpus 2 --streams 512 --steps 5 --warmup 2 $B
line $O/${T}_b_n2_rehearsal.log > $O/${T}_b_n2_rehearsal.json
step latency 200 python tools/latency_bench.py; cp $O/${T}_latency.log $O/${T}_single_stream_latency.txt
# fft.receive at the frame sizes other audio rates give (k_fft_rt / k_dft_any) against the 9600-sample default
for n in 9600 4800 19200 4410 2205 3200 800 8820 1102; do timeout -k 10 120 python tools/fft_n_bench.py $n 1024 2>&1 | tail -1; done > $O/${T}_fft_n_bench.txt
# config 5 behind the C ABI alone: the C++ harness, one process, jsdr_group_* (RCCL with one rank here; 8 members on one device with copies)
step harness_n1 300 java-sdr_amd/host/jsdr_harness --gpus 1 --streams 8192 --psd --steps 10 --warmup 3; grep '^{"harness' $O/${T}_harness_n1.log > $O/${T}_harness_n1.json
step harness_n8 300 java-sdr_amd/host/jsdr_harness --gpus 8 --streams 8192 --same-device --copy-gather --steps 5 --warmup 2; grep '^{"harness' $O/${T}_harness_n8.log > $O/${T}_harness_n8_rehearsal.json
rm -f $O/${T}_*.log.tmp
echo "profile set $T done" | tee -a $O/session.log
