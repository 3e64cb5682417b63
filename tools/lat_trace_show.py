#!/usr/bin/env python3
"""tools/lat_trace_show.py <kernel_trace.csv> -- the kernel timeline of the last receive() calls of tools/lat_trace.py"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-22:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%6.1f us  %s" % ((s - t0) / 1000.0, (e - s) / 1000.0, r["Kernel_Name"][:60]))
