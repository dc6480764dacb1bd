#!/usr/bin/env python3
"""Tabulate rocprofv3 --pmc output (counter_collection.csv): one row per dispatch of the
kernels whose name contains the filter, one column per counter.
Usage: pmc_table.py <dir-or-csv> [name-filter] [min-value-of-first-counter]"""
import csv, glob, os, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = {}
names = []
for r in rows:
    if flt not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:24], "vgpr": r.get("VGPR_Count"), "scr": r.get("Scratch_Size")})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    if r["Counter_Name"] not in names:
        names.append(r["Counter_Name"])
print("dispatch", *names, sep="\t")
for k in sorted(disp):
    d = disp[k]
    print(k, *["%.4g" % d.get(n, 0) for n in names], sep="\t")
