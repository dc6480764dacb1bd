#!/usr/bin/env python3
"""Generates tests/golden/oracle_<scene>.npz: outputs of the CPU oracle (pinned against
the reference's gallery image, see tests/test_oracle_golden.py) for the scenes of the five
benchmark configurations and one scene per primitive / shader family, at 64 pixels wide,
anti-aliasing on, seed 1.  Inputs are the named scenes of harness/scenes.cpp;
expected outputs are RGB8, primary hit ids and the AA mask."""
import importlib, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
scn = importlib.import_module("harness.scenes")
from oracle import binding as orc

SCENES = ["basic_setup_scene", "plane_test_1", "cube_test_1", "shadow_test_1", "child_transform_test_1",
          "light_room", "glass_monkey", "mesh_basic", "monkey_3", "many_balls", "many_balls_forest",
          "subtraction_outside1", "subtraction_outside4", "xor_test", "reuleaux", "bump_map_basic_1",
          "octahedron_basic", "polyhedron_basic", "primitive_zoo", "scene_test_4", "background_1",
          "direction_light_room", "dice_scene_lo_res", "texture_cube", "transparent_shapes", "normal_map_room_1", "tetrahedron_basic"]
WIDTH = 64

def golden_view(scene):
    h = max(3, int(round(WIDTH * scene.dim.height / scene.dim.width)))
    return scene.view(width=WIDTH, height=h, anti_alias=True, seed=1)

if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    for name in SCENES:
        s = scn.Scene(name)
        dim, cam, cfg = golden_view(s)
        r = orc.render(s, dim, cam, cfg)
        assert r["assert_failures"] == 0, name
        np.savez_compressed(os.path.join(out, f"oracle_{name}.npz"), rgb=r["rgb"], hit_id=r["hit_id"], aa_mask=r["aa_mask"])
        print(name, r["rgb"].shape, r["counters"]["rays_total"])
