"""Time of the output stage: rt_write_png of a rendered frame (filters on the device, deflate on the host)
at 1920x1080 and 3840x2160, and what PIL's encoder takes for the same frame on the host."""
import importlib
import io
import os
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
render = importlib.import_module("rust-raytracer_amd.render")
scenes = importlib.import_module("harness.scenes")

s = scenes.Scene("glass_monkey")
ds = render.DeviceScene(s)
with tempfile.TemporaryDirectory() as d:
    for w, h in ((1920, 1080), (3840, 2160)):
        dim, cam, cfg = s.view(width=w, height=h, anti_alias=True, seed=1)
        rgb = ds.render(dim, cam, cfg, want_aux=False)["rgb"]
        stem = os.path.join(d, f"f{w}")
        render.write_png(rgb, stem)
        t0 = time.perf_counter()
        for _ in range(3):
            render.write_png(rgb, stem)
        ours = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="PNG", compress_level=6)
        pil = time.perf_counter() - t0
        assert np.array_equal(np.asarray(Image.open(stem + ".png")), rgb)
        print(f"{w}x{h}: rt_write_png {ours * 1e3:.1f} ms ({os.path.getsize(stem + '.png') / 1e6:.2f} MB); PIL level 6 on the host {pil * 1e3:.1f} ms ({len(buf.getvalue()) / 1e6:.2f} MB)")
