#!/usr/bin/env python3
"""For the scenes that hold exact geometric ties: primary hit ids and pixels that differ between the HIP path and
the CPU oracle at the parity tests' size, with the queue pipeline's exact tie decisions on and off (RT_WF_TIES)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
scenes = importlib.import_module("harness.scenes")
render = importlib.import_module("rust-raytracer_amd.render")
from oracle import binding as oracle

names = sys.argv[1:] or ["child_transform_test_1", "subdivision_light", "refraction_03", "octahedron_basic", "octahedron_basic_02",
                         "triangle_basic", "scene_light_1_05", "plane_texture"]
for name in names:
    s = scenes.Scene(name)
    h = max(3, int(round(112 * s.dim.height / s.dim.width)))
    dim, cam, cfg = s.view(width=112, height=h, anti_alias=True, seed=1)
    want = oracle.render(s, dim, cam, cfg)
    for ties in ("1", "0"):
        os.environ["RT_WF_TIES"] = ties
        ds = render.DeviceScene(s, device=0)
        got = ds.render(dim, cam, cfg, count_work=True)
        ds.close()
        diff = np.abs(got["rgb"].astype(int) - want["rgb"].astype(int)).max(axis=2)
        mse = ((got["rgb"].astype(float) - want["rgb"].astype(float)) ** 2).mean()
        psnr = float("inf") if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
        print(f"{name:26s} ties={ties} schedule {got['counters']['schedule']}: hit ids differing {(got['hit_id'] != want['hit_id']).sum():4d}, "
              f"within 1 LSB {100 * (diff <= 1).mean():.3f} %, PSNR {psnr:.1f} dB, aa mask differs {(got['aa_mask'] != want['aa_mask']).sum()}", flush=True)
