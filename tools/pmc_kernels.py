#!/usr/bin/env python3
"""Per-kernel totals of one rocprofv3 --pmc counter over a run.
Usage: pmc_kernels.py <dir with *counter_collection.csv> <COUNTER> [scale]
FETCH_SIZE / WRITE_SIZE are in KiB; give scale 2048 for bytes read (the counter sees half of them on this
part, see the calibration in profiles/*_pmc_hbm.txt) and 1024 for bytes written."""
import collections, csv, glob, os, sys

src, counter = sys.argv[1], sys.argv[2]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
total, calls = collections.defaultdict(float), collections.Counter()
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0][:56]
            total[k] += float(r["Counter_Value"]) * scale
            calls[k] += 1
for k in sorted(total, key=lambda k: -total[k])[:14]:
    print("%-58s calls %6d  %s %14.4g" % (k, calls[k], counter, total[k]))
print("%-58s calls %6d  %s %14.4g" % ("all kernels", sum(calls.values()), counter, sum(total.values())))
