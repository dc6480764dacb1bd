#!/usr/bin/env python3
"""Tie-heavy scenes at a large frame: is the frame the same from run to run, and does it still agree with the CPU
oracle in every primary hit id (i.e. did every noted tie decision fit its sub-queue's list)?
Usage: tie_scale_check.py [width] [scene ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
scenes = importlib.import_module("harness.scenes")
render = importlib.import_module("rust-raytracer_amd.render")
from oracle import binding as oracle

width = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
for name in sys.argv[2:] or ["scene_light_1_05", "refraction_03", "child_transform_test_1", "octahedron_basic_02", "polygon_basic"]:
    s = scenes.Scene(name)
    h = max(3, int(round(width * s.dim.height / s.dim.width)))
    dim, cam, cfg = s.view(width=width, height=h, anti_alias=True, seed=1)
    ds = render.DeviceScene(s, device=0)
    a = ds.render(dim, cam, cfg)
    b = ds.render(dim, cam, cfg)
    ds.close()
    want = oracle.render(s, dim, cam, cfg)
    diff = np.abs(a["rgb"].astype(int) - want["rgb"].astype(int)).max(axis=2)
    print(f"{name:26s} {width}x{h}: two runs equal {np.array_equal(a['rgb'], b['rgb']) and np.array_equal(a['hit_id'], b['hit_id'])}, "
          f"hit ids differing from the oracle {(a['hit_id'] != want['hit_id']).sum()}, pixels beyond 1 LSB {(diff > 1).sum()}", flush=True)
