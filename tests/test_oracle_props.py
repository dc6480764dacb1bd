"""Properties of the CPU oracle itself (the checker must be trustworthy before it checks)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def small_view(scene, width=64, **kw):
    h = max(3, int(round(width * scene.dim.height / scene.dim.width)))
    return scene.view(width=width, height=h, **kw)


@pytest.mark.parametrize("name", ["basic_setup_scene", "light_room", "glass_monkey", "many_balls", "subtraction_outside1",
                                  "primitive_zoo", "monkey_3", "xor_test"])
def test_committed_goldens_are_reproduced(name, scene_cache, oracle):
    s = scene_cache(name)
    dim, cam, cfg = small_view(s, anti_alias=True, seed=1)
    r = oracle.render(s, dim, cam, cfg)
    g = np.load(os.path.join(GOLDEN, f"oracle_{name}.npz"))
    assert r["assert_failures"] == 0
    assert np.array_equal(r["rgb"], g["rgb"])
    assert np.array_equal(r["hit_id"], g["hit_id"])
    assert np.array_equal(r["aa_mask"], g["aa_mask"])


def test_kd_tree_equals_brute_force(scene_cache, oracle):
    """The reference's kd-tree walk (mesh.rs:167-279) returns the exact closest hit: any exact
    structure — such as the device BVH — must therefore agree with it."""
    s = scene_cache("glass_monkey")
    dim, cam, cfg = small_view(s, 96, anti_alias=False)
    a = oracle.render(s, dim, cam, cfg)
    oracle.lib().oracle_set_mesh_brute_force(1)
    try:
        b = oracle.render(s, dim, cam, cfg)
    finally:
        oracle.lib().oracle_set_mesh_brute_force(0)
    assert np.array_equal(a["color"], b["color"])
    assert a["counters"]["rays_total"] == b["counters"]["rays_total"]


def test_seed_controls_the_stochastic_stages(scene_cache, oracle):
    s = scene_cache("light_room")
    dim, cam, cfg = small_view(s, 48, anti_alias=True, seed=1)
    a = oracle.render(s, dim, cam, cfg)
    b = oracle.render(s, dim, cam, cfg)
    assert np.array_equal(a["color"], b["color"])
    dim, cam, cfg2 = small_view(s, 48, anti_alias=True, seed=2)
    c = oracle.render(s, dim, cam, cfg2)
    assert not np.array_equal(a["color"], c["color"])
    # thread count and chunking must not matter (multithread.rs only schedules)
    dim, cam, cfg3 = small_view(s, 48, anti_alias=True, seed=1, num_threads=3, workload_split=7)
    d = oracle.render(s, dim, cam, cfg3)
    assert np.array_equal(a["color"], d["color"])


def test_aa_quirks(scene_cache, oracle):
    s = scene_cache("glass_monkey")
    dim, cam, cfg = small_view(s, 64, anti_alias=True)
    r = oracle.render(s, dim, cam, cfg)
    m = r["aa_mask"]
    assert m.sum() > 0
    # border pixels are never re-sampled (render.rs:205-206); re-sampled pixels get distance 0 (render.rs:306)
    assert m[0].sum() == m[-1].sum() == m[:, 0].sum() == m[:, -1].sum() == 0
    assert np.all(r["distance"][m == 1] == 0.0)
    dim, cam, cfg0 = small_view(s, 64, anti_alias=False)
    r0 = oracle.render(s, dim, cam, cfg0)
    assert np.array_equal(r0["color"][m == 0], r["color"][m == 0])
    assert r["counters"]["rays_aa"] == 5 * int(m.sum())


def test_row_block_split_is_invisible(scene_cache, oracle):
    s = scene_cache("glass_monkey")
    dim, cam, cfg = small_view(s, 48, anti_alias=True)
    full = oracle.render(s, dim, cam, cfg)
    out = np.zeros_like(full["rgb"])
    for rank in range(3):
        dim, cam, c = small_view(s, 48, anti_alias=True, rank=rank, nranks=3, tile_rows=4)
        part = oracle.render(s, dim, cam, c)
        rows = [y for y in range(dim.height) if (y // 4) % 3 == rank]
        out[rows] = part["rgb"][rows]
    assert np.array_equal(out, full["rgb"])


def test_missing_root_primitive_renders_black(scene_cache, oracle):
    # tests/basic_test.rs:24-33 replaces the root by a node without primitive
    s = scene_cache("white_sphere_one_light")
    dim, cam, cfg = small_view(s, 32)
    r = oracle.render(s, dim, cam, cfg)
    assert r["rgb"].max() == 0 and np.all(np.isinf(r["distance"])) and np.all(r["hit_id"] == 0xFFFFFFFF)
