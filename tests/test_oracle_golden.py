"""Pins the CPU oracle against the reference's own output.

gallery/glass_monkey.png (1000x1000) is the one image the reference ships for the
north-star scene.  It was rendered by an earlier revision of tests/mesh_test.rs /
src/shader/translucent.rs: glass() weighted (0.5, 0.5) and TranslucentShader cast only the
transmitted ray — the line that survives as a comment at translucent.rs:59.  With those
two settings (the oracle's `legacy_translucent` mode; everything else — camera, room,
lights, mesh, kd-tree, refraction, reflection, Phong, shadows, recursion cut-offs,
AA detector, encode — as in the current sources) every pixel the AA pass did not
re-sample must be bit-identical.  AA pixels carry the reference's unseedable thread_rng
jitter and are only checked statistically.

The Fresnel split of the current TranslucentShader (translucent.rs:19-46) is not
covered by any reference output: parity unpinned for those 28 lines."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gallery_glass_monkey.png")


@pytest.fixture(scope="module")
def legacy_scene(scenes_mod, abi_mod, oracle):
    s = scenes_mod.Scene("glass_monkey")
    d = C.cast(s.desc, C.POINTER(abi_mod.RtSceneDesc)).contents
    glass = d.shaders[d.nodes[8].shader]
    d.shader_child_weight[glass.first_child] = 0.5
    d.shader_child_weight[glass.first_child + 1] = 0.5
    oracle.lib().oracle_set_legacy_translucent(1)
    yield s
    oracle.lib().oracle_set_legacy_translucent(0)


def test_oracle_reproduces_the_reference_gallery_image(legacy_scene, oracle):
    gold = np.asarray(Image.open(GOLD).convert("RGB"))
    assert gold.shape == (1000, 1000, 3)
    # every 4th block of 10 rows (plus the AA halo) keeps the CPU suite short: 250 rows
    dim, cam, cfg = legacy_scene.view(rank=1, nranks=4, tile_rows=10)
    assert (dim.width, dim.height, cfg.anti_alias, cfg.aa_rays, cfg.recursion_depth) == (1000, 1000, 1, 5, 20)
    r = oracle.render(legacy_scene, dim, cam, cfg)
    assert r["assert_failures"] == 0
    rows = np.array([y for y in range(1000) if (y // 10) % 4 == 1])
    rgb, g, mask, hid = r["rgb"][rows], gold[rows], r["aa_mask"][rows], r["hit_id"][rows]
    diff = np.abs(rgb.astype(int) - g.astype(int)).max(axis=2)
    det = mask == 0
    assert det.sum() > 200000
    assert (hid[det] == 8).sum() > 5000, "the band must contain deterministic glass pixels"
    assert diff[det].max() == 0, f"{(diff[det] > 0).sum()} deterministic pixels differ from the reference image"
    # AA pixels: different jitter, same distribution — block means agree
    aa = ~det
    assert aa.sum() > 20000
    assert abs(rgb[aa].astype(float).mean() - g[aa].astype(float).mean()) < 1.5


def test_hand_checked_wall_pixels(legacy_scene, oracle):
    """Five wall pixels of the gallery image (x, y) → RGB, all outside the AA mask."""
    gold = np.asarray(Image.open(GOLD).convert("RGB"))
    expect = {(500, 20): (130, 0, 0), (100, 300): (128, 0, 128), (900, 300): (0, 107, 0), (30, 950): (109, 109, 0), (250, 150): (234, 0, 234)}
    for (x, y), rgb in expect.items():
        assert tuple(gold[y, x]) == rgb
    # rows 16..31 and 944..959 hold two of them: render those two blocks
    dim, cam, cfg = legacy_scene.view(rank=1, nranks=58, tile_rows=16)
    r = oracle.render(legacy_scene, dim, cam, cfg)
    assert tuple(r["rgb"][20, 500]) == (130, 0, 0)
    assert tuple(r["rgb"][950, 30]) == (109, 109, 0)


# ---- the area-light images of the reference's gallery (statistical: thread_rng, and a light-sampling
# revision whose penumbrae differ from today's subdivide_illumination; see tests/test_gpu_gallery.py) ----
def _blocks(img, b=16):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].astype(float).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("name,within1,within3", [("light_room", 0.80, 0.91), ("monkey_1", 0.68, 0.90)])
def test_oracle_agrees_with_the_area_light_gallery_images_in_block_means(name, within1, within3, scenes_mod, oracle):
    gold = np.asarray(Image.open(os.path.join(os.path.dirname(GOLD), f"gallery_{name}.png")).convert("RGB"))
    s = scenes_mod.Scene(name)
    # every 5th block of 16 rows: 200 of the 1000 rows
    dim, cam, cfg = s.view(width=1000, height=1000, anti_alias=True, seed=1, rank=2, nranks=5, tile_rows=16)
    r = oracle.render(s, dim, cam, cfg)
    assert r["assert_failures"] == 0
    rows = np.array([y for y in range(1000) if (y // 16) % 5 == 2 and y // 16 < 62])
    d = np.abs(_blocks(r["rgb"][rows]) - _blocks(gold[rows])).max(axis=2)
    assert (d <= 1.0).mean() >= within1 and (d <= 3.0).mean() >= within3, ((d <= 1.0).mean(), (d <= 3.0).mean())
    assert np.abs(r["rgb"][rows].reshape(-1, 3).mean(axis=0) - gold[rows].reshape(-1, 3).astype(float).mean(axis=0)).max() < 1.0
