#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters (counter_collection.csv) per kernel name over its dispatches and
print derived ratios: lanes = THREAD_CYCLES_VALU / INSTS_VALU, VALU-busy and wait shares of
WAVE_CYCLES.  Usage: pmc_sum.py <dir-or-csv> [name-filter]"""
import csv, glob, os, sys, collections
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = path if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if flt not in n:
        continue
    tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[n] = (r.get("VGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size"))
for n, c in tot.items():
    line = "%s: vgpr/scratch/lds %s;" % (n, "/".join(str(x) for x in meta[n]))
    line += " " + " ".join("%s=%.3g" % (k, v) for k, v in sorted(c.items()))
    if c.get("SQ_INSTS_VALU"):
        line += " lanes=%.1f" % (c.get("SQ_THREAD_CYCLES_VALU", 0) / c["SQ_INSTS_VALU"])
    if c.get("SQ_WAVE_CYCLES"):
        line += " valu_busy=%.0f%% wait_any=%.0f%%" % (100 * c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"], 100 * c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"])
    print(line)
