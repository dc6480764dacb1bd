#!/usr/bin/env python3
"""Render one rank's share of an n-way split a few times (for rocprofv3).
Usage: prof_rank.py scene width height rank nranks [frames]"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
scn = importlib.import_module("harness.scenes")
rnd = importlib.import_module("rust-raytracer_amd.render")
name, w, h, rank, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
frames = int(sys.argv[6]) if len(sys.argv) > 6 else 2
s = scn.Scene(name)
dim, cam, cfg = s.view(width=w, height=h, anti_alias=True, seed=1)
cfg.rank, cfg.nranks, cfg.tile_rows = rank, n, 16
ds = rnd.DeviceScene(s)
for i in range(frames):
    r = ds.render(dim, cam, cfg, want_aux=False)
    print(f"frame {i}: pass1 {r['ms']['pass1']:.2f} ms aa {r['ms']['aa']:.2f} ms", flush=True)
