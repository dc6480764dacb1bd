#!/usr/bin/env python3
"""Count the memory instructions by kind in the kernels of a gfx950 assembly listing
(hipcc --cuda-device-only -S).  Usage: isa_mem_ops.py <file.s> [name-filter]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; cnt = None
kinds = ["flat_load", "flat_store", "flat_atomic", "global_load", "global_store", "global_atomic", "ds_read", "ds_write", "ds_add", "scratch_load", "scratch_store", "s_waitcnt", "s_load", "v_"]
def flush():
    if cur and flt in cur:
        print(cur[:70], " ".join("%s=%d" % (k, cnt[k]) for k in kinds if cnt[k]))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush(); cur = m.group(1); cnt = collections.Counter(); continue
    if l.startswith(".Lfunc_end"):
        flush(); cur = None; continue
    if cur:
        t = l.strip()
        for k in kinds:
            if t.startswith(k):
                cnt[k] += 1; break
