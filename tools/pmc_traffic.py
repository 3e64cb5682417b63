#!/usr/bin/env python3
"""tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> -- per-kernel HBM bytes per launch from the two separate
rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE).  Units and the gfx950 correction as MI355X_MICROARCH.md
prescribes: counters are KiB; FETCH_SIZE reports half the bytes of coalesced streaming reads => hbm = 2*FETCH + WRITE."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            m = re.search(r"(k_[a-z0-9_]+)", name)
            if not m:
                continue
            key = m.group(1)  # the real kernel: k_front_reg / k_front_fft / k_fm ... (bench.py looks the variant up by name)
            acc[key].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def table(fetch_dir, write_dir, samples):
    """per-kernel {fetch raw, write, hbm bytes per launch} from the two passes' output directories"""
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    out = {"_samples_per_launch": samples, "_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/gpu_session.sh pmc_rd pmc_wr), "
                    "bench.py --steps 2 --warmup 1, per launch of _samples_per_launch IQ samples.  Counter unit = KiB.  Per MI355X_MICROARCH.md the gfx950 "
                    "FETCH_SIZE reports half the bytes of coalesced streaming reads: hbm_bytes = 2*FETCH + WRITE "
                    "(calibration: k_synth_dbpsk writes exactly 4.295e9 B; k_fft reads 4.295e9 B of int16 IQ)."}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith('_'):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        out[k] = {"fetch_size_bytes_raw": int(f), "write_size_bytes": int(w), "hbm_bytes_per_launch": int(2 * f + w)}
    return out


def valu_table(d, nsteps, nsimd=1024, nse=32):
    """per kernel and PER STEP (the pass ran nsteps steps): VALU issue cycles per SIMD (SQ_ACTIVE_INST_VALU counts quad-cycles
    over all SIMDs), busy cycles per shader engine (SQ_BUSY_CYCLES is summed over the SEs), wave instructions, and the kernels'
    time under the profiler (dispatch timestamps) -- what bench.py's roofline.valu_issue_frac is made of"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(1)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            did = (f, r["Dispatch_Id"])
            if did not in seen:
                seen.add(did)
                acc[k]["_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                acc[k]["_launches"] += 1
    out = {}
    for k, v in acc.items():
        if "SQ_ACTIVE_INST_VALU" not in v or "SQ_BUSY_CYCLES" not in v:
            continue
        out[k] = {"valu_cycles_per_simd": v["SQ_ACTIVE_INST_VALU"] * 4.0 / nsimd / nsteps,
                  "busy_cycles_per_se": v["SQ_BUSY_CYCLES"] / nse / nsteps,
                  "valu_wave_instructions": v.get("SQ_INSTS_VALU", 0.0) / nsteps,
                  "ms_under_profiler": v["_ns"] * 1e-6 / nsteps, "launches_per_step": v["_launches"] / nsteps}
    return out


if __name__ == "__main__":
    samples = int(sys.argv[4]) if len(sys.argv) > 4 else 1024 * 1048576  # IQ samples one launch of the batch kernels covers
    out = table(sys.argv[1], sys.argv[2], samples)
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if not k.startswith("_")}))
