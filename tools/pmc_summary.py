#!/usr/bin/env python3
"""tools/pmc_summary.py <rocprofv3 output dir> -- per-kernel mean of every counter in *_counter_collection.csv"""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(rows.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v) / len(v):16.1f}   n={len(v)}")
