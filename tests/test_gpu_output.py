"""The output stage: rt_write_png / rt_write_png_device (write_to_png, src/lib.rs:41-46) and the
progress reports of rt_render (ProgressTracker, src/progress_tracker.rs:48-77).  A PNG written by the
library must decode (with an independent decoder) to exactly the bytes of the frame."""
import os

import numpy as np
import pytest
import torch   # (before the render library is loaded: the two bring their own HIP runtimes, and this order is the one that works)
from PIL import Image

pytestmark = pytest.mark.gpu


def test_png_of_a_rendered_frame_decodes_to_the_frame(render_mod, scene_cache, tmp_path):
    s = scene_cache("glass_monkey")
    ds = render_mod.DeviceScene(s, device=0)
    try:
        dim, cam, cfg = s.view(width=333, height=217, anti_alias=True, seed=1)
        rgb = ds.render(dim, cam, cfg, want_aux=False)["rgb"]
    finally:
        ds.close()
    stem = str(tmp_path / "frame")
    render_mod.write_png(rgb, stem)
    assert os.path.exists(stem + ".png")                       # the reference appends the extension too
    back = np.asarray(Image.open(stem + ".png"))
    assert back.dtype == np.uint8 and back.shape == rgb.shape and np.array_equal(back, rgb)
    assert os.path.getsize(stem + ".png") < rgb.size           # filtered and deflated, not stored


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (5, 1), (64, 64), (3, 1025)])
def test_png_of_arbitrary_bytes_round_trips(render_mod, tmp_path, shape):
    rng = np.random.default_rng(shape[0] * 10007 + shape[1])
    rgb = rng.integers(0, 256, size=(shape[0], shape[1], 3), dtype=np.uint8)
    if shape == (64, 64):                                      # rows that favour each of the five filters
        rgb[0:8] = 17
        rgb[8:16] = np.arange(64, dtype=np.uint8)[None, :, None] * 3
        rgb[16:24] = rgb[15]
        rgb[24:32] = (np.arange(64)[None, :, None] * 2 + np.arange(8)[:, None, None] * 5).astype(np.uint8)
    stem = str(tmp_path / f"noise_{shape[0]}x{shape[1]}")
    render_mod.write_png(rgb, stem)
    assert np.array_equal(np.asarray(Image.open(stem + ".png").convert("RGB")), rgb)
    if shape == (64, 64):
        import struct
        import zlib
        raw = open(stem + ".png", "rb").read()
        at, idat = 8, b""
        while at < len(raw):
            n, kind = struct.unpack(">I4s", raw[at:at + 8])
            if kind == b"IDAT":
                idat += raw[at + 8:at + 8 + n]
            assert zlib.crc32(raw[at + 4:at + 8 + n]) == struct.unpack(">I", raw[at + 8 + n:at + 12 + n])[0]
            at += 12 + n
        lines = zlib.decompress(idat)
        filters = {lines[y * (1 + 3 * 64)] for y in range(64)}
        assert len(filters) >= 3 and filters <= {0, 1, 2, 3, 4}   # the choice is made per row


def test_png_from_device_memory(render_mod, scene_cache, tmp_path):
    s = scene_cache("cube_test_1")
    ds = render_mod.DeviceScene(s, device=0)
    try:
        dim, cam, cfg = s.view(width=96, height=64, anti_alias=False, seed=1)
        want = ds.render(dim, cam, cfg, want_aux=False)["rgb"]
        d_rgb = torch.zeros(64 * 96 * 3, dtype=torch.uint8, device="cuda:0")
        stream = torch.cuda.current_stream().cuda_stream
        ds.render_device(dim, cam, cfg, d_rgb.data_ptr(), stream)
        stem = str(tmp_path / "device_frame")
        render_mod.write_png_device(d_rgb.data_ptr(), 96, 64, stem, stream_ptr=stream)
        assert ds.frame_dropped() == 0
    finally:
        ds.close()
    assert np.array_equal(np.asarray(Image.open(stem + ".png")), want)


def test_png_errors_are_statuses(render_mod, tmp_path):
    rgb = np.zeros((4, 4, 3), dtype=np.uint8)
    with pytest.raises(render_mod.RtError) as e:
        render_mod.write_png(rgb, str(tmp_path / "no_such_directory" / "x"))   # the reference creates no directories either
    assert e.value.code == 1 and "cannot write" in str(e.value)
    with pytest.raises(render_mod.RtError):
        render_mod.write_png(rgb, str(tmp_path / "x"), device=99)


def test_progress_reports_follow_the_tracker(render_mod, scene_cache):
    s = scene_cache("glass_monkey")
    ds = render_mod.DeviceScene(s, device=0)
    try:
        seen = []
        ds.set_progress(lambda stage, done, total, t: seen.append((stage, done, total, t)))
        dim, cam, cfg = s.view(width=160, height=120, anti_alias=True, seed=1)
        out = ds.render(dim, cam, cfg)
        flagged = int(out["aa_mask"].sum())
        assert [e[0] for e in seen] == [0, 1, 2, 3, 4]
        assert seen[0][1:3] == (0, 160 * 120) and seen[1][1:3] == (160 * 120, 160 * 120)
        assert seen[2][1:3] == (0, flagged) and seen[3][1:3] == (flagged, flagged) and flagged > 0
        assert 0.0 < seen[1][3] <= seen[3][3] <= seen[4][3]
        assert abs(seen[4][3] - (out["ms"]["pass1"] + out["ms"]["aa"] + out["ms"]["encode"]) * 1e-3) < 1e-9
        seen.clear()
        dim, cam, cfg = s.view(width=160, height=120, anti_alias=False, seed=1)
        ds.render(dim, cam, cfg)
        assert [e[0] for e in seen] == [0, 1, 4]               # no "START ANTIALIASING" without the pass
        seen.clear()
        ds.set_progress(None)
        ds.render(dim, cam, cfg)
        assert seen == []
    finally:
        ds.close()
