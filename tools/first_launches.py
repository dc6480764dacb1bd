#!/usr/bin/env python3
"""Print the durations of the first launches of each kernel (name filter) in a rocprofv3 kernel trace.
Usage: first_launches.py <dir> <filter> [n]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
flt = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seen = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    if flt in k and seen.setdefault(k, 0) < n:
        seen[k] += 1
        print("%-40s %9.1f us" % (k[-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
