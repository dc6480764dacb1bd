#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name the number of dispatches, the sum of
durations and the time during which at least one dispatch of that name ran; for the whole trace the
union of all dispatches and the idle gaps between them.  Usage: trace_timeline.py <dir-or-csv> [skip-first-n-frames-marker]"""
import csv, glob, os, sys, collections
path = sys.argv[1]
f = path if path.endswith(".csv") else sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
# frames are separated by k_encode dispatches; keep the last complete frame
enc = [i for i, r in enumerate(rows) if r[0].startswith("k_encode")]
if len(enc) >= 2:
    rows = rows[enc[-2] + 1: enc[-1] + 1]
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
by = collections.defaultdict(list)
for n, s, e in rows: by[n].append((s, e))
t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
print(f"frame span {1e-6 * (t1 - t0):.2f} ms, {len(rows)} dispatches, busy (union) {1e-6 * union([(s, e) for _, s, e in rows]):.2f} ms")
for n, iv in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    print(f"{n[:60]:60s} n={len(iv):4d} sum {1e-6 * sum(e - s for s, e in iv):8.2f} ms  union {1e-6 * union(iv):8.2f} ms  max {1e-6 * max(e - s for s, e in iv):6.2f}")
if "-v" in sys.argv:
    for n, s, e in rows: print(f"{1e-6 * (s - t0):9.3f} {1e-6 * (e - s):8.3f} {n[:50]}")
