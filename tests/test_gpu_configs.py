"""The benchmark configurations at their own sizes (light_room 1920x1080, many_balls_forest 3840x2160,
subtraction_outside1 3840x2160 with the detector's threshold and supersampled) through properties that
need no oracle: the frame is deterministic, anti-aliasing only touches flagged pixels, the AA pass casts
aa_rays rays per flagged pixel, the image does not depend on how rows are split over ranks, no fixed
stack overflowed, and the schedule that ran is the one the scene is meant for, in one attempt.  The
supersampled variant is also compared with the oracle at 256 pixels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QUEUES, PERSISTENT = 1, 2

CONFIGS = [
    # name, scene, width, height, config overrides, schedule
    ("light_room", "light_room", 1920, 1080, {}, PERSISTENT),          # square light: visibility inside the shader
    ("many_balls_forest", "many_balls_forest", 3840, 2160, {}, QUEUES),
    ("subtraction_outside", "subtraction_outside1", 3840, 2160, {}, QUEUES),
    ("subtraction_outside_supersampled", "subtraction_outside1", 3840, 2160, {"aa_threshold": 0.0}, QUEUES),
]


@pytest.mark.parametrize("label,name,width,height,overrides,schedule", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_config_at_full_size(label, name, width, height, overrides, schedule, render_mod, scene_cache):
    s = scene_cache(name)
    ds = render_mod.DeviceScene(s, device=0)
    try:
        dim, cam, cfg = s.view(width=width, height=height, anti_alias=True, seed=1, **overrides)
        a = ds.render(dim, cam, cfg, count_work=True)
        assert a["counters"]["stack_overflows"] == 0
        assert a["counters"]["schedule"] == schedule, "the frame ran on the other schedule"
        assert a["counters"]["attempts"] == 1, "a ray generation outgrew the queues and the frame had to be redone"
        b = ds.render(dim, cam, cfg)
        assert b["counters"]["schedule"] == schedule and b["counters"]["attempts"] == 1
        assert np.array_equal(a["rgb"], b["rgb"]) and np.array_equal(a["color"], b["color"])
        # AA only touches flagged interior pixels; everything else equals the AA-off frame
        dim, cam, cfg0 = s.view(width=width, height=height, anti_alias=False, seed=1, **overrides)
        c = ds.render(dim, cam, cfg0)
        m = a["aa_mask"] == 1
        assert m.sum() > 0 and m[0].sum() == m[-1].sum() == m[:, 0].sum() == m[:, -1].sum() == 0
        assert np.array_equal(a["color"][~m], c["color"][~m])
        assert np.all(a["distance"][m] == 0.0)
        assert a["counters"]["rays_aa"] == cfg.aa_rays * int(m.sum()) and a["counters"]["aa_pixels"] == int(m.sum())
        if overrides.get("aa_threshold") == 0.0:
            # supersampled: every interior pixel that hit something and differs at all from a neighbour
            hit = np.isfinite(c["distance"])
            assert m.sum() > 0.5 * hit[1:-1, 1:-1].sum()
        # the image must not depend on how rows are split over ranks
        lib = render_mod.hip_lib()
        out = np.zeros_like(a["rgb"])
        for rank in range(3):
            dim, cam, c3 = s.view(width=width, height=height, anti_alias=True, seed=1, rank=rank, nranks=3, tile_rows=8, **overrides)
            part = ds.render(dim, cam, c3, want_aux=False)["rgb"]
            assert part.shape[0] == lib.rt_owned_rows_capacity(height, 3, 8)
            for y in range(height):
                if (y // 8) % 3 == rank:
                    out[y] = part[((y // 8) // 3) * 8 + y % 8]
        assert np.array_equal(out, a["rgb"])
    finally:
        ds.close()


def test_supersampled_variant_matches_the_oracle(render_mod, scene_cache, oracle):
    s = scene_cache("subtraction_outside1")
    dim, cam, cfg = s.view(width=256, height=144, anti_alias=True, seed=1, aa_threshold=0.0)
    ds = render_mod.DeviceScene(s, device=0)
    got = ds.render(dim, cam, cfg, count_work=True)
    ds.close()
    want = oracle.render(s, dim, cam, cfg)
    diff = np.abs(got["rgb"].astype(int) - want["rgb"].astype(int)).max(axis=2)
    assert (diff <= 1).mean() >= 0.999
    assert np.array_equal(got["hit_id"], want["hit_id"]) and np.array_equal(got["aa_mask"], want["aa_mask"])
    assert got["aa_mask"].sum() > 0.5 * np.isfinite(want["distance"])[1:-1, 1:-1].sum() or got["aa_mask"].sum() > 1000
    assert got["counters"]["rays_aa"] == want["counters"]["rays_aa"] == 5 * int(got["aa_mask"].sum())
    assert np.quantile(np.abs(got["color"] - want["color"]).max(axis=2), 0.999) < 1e-6
