#!/usr/bin/env python3
"""Times every rank's share of an n-way split of one frame, one after the other on one GPU:
the slowest share is what an n-GPU run of the frame would take, the gather aside.
Usage: prof_split.py [scene] [width] [height] [nranks ...]   (TILE_ROWS in the environment: rows per block, default 16)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
scn = importlib.import_module("harness.scenes")
rnd = importlib.import_module("rust-raytracer_amd.render")
name = sys.argv[1] if len(sys.argv) > 1 else "glass_monkey"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
splits = [int(x) for x in sys.argv[4:]] or [1, 2, 4, 8]
s = scn.Scene(name)
dim, cam, cfg = s.view(width=w, height=h, anti_alias=True, seed=1)
ds = rnd.DeviceScene(s)
for n in splits:
    shares = []
    for r in range(n):
        c = cfg.copy()
        c.rank, c.nranks, c.tile_rows = r, n, int(os.environ.get("TILE_ROWS", "16"))
        best = 1e9
        for _ in range(3):
            t = time.time()
            res = ds.render(dim, cam, c, want_aux=False)
            best = min(best, res["ms"]["pass1"] + res["ms"]["aa"])
        shares.append(best)
    print(f"{n} ranks: shares {['%.1f' % x for x in shares]} ms, slowest {max(shares):.1f} ms, sum {sum(shares):.1f} ms", flush=True)
