#!/usr/bin/env python3
"""Do the area-light images of the reference's gallery come from the sampling that SquareLight still carries as
dead code (400 uniform, shadow-tested samples: src/light.rs:355-363 through the default get_illums_at, :17-44)
rather than from today's subdivide_illumination (:248-298)?  Renders a fifth of light_room / monkey_1 at the
images' size with the CPU oracle both ways and compares 16 x 16 block means with the gallery image.
Usage: tools/penumbra_experiment.py [light_room|monkey_1 ...]      (minutes of CPU time per scene)"""
import importlib, os, sys, time
import numpy as np
from PIL import Image
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
scenes = importlib.import_module("harness.scenes")
from oracle import binding as oracle


def blocks(img, b=16):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].astype(float).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))


for name in sys.argv[1:] or ["light_room", "monkey_1"]:
    gold = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", f"gallery_{name}.png")).convert("RGB"))
    s = scenes.Scene(name)
    dim, cam, cfg = s.view(width=1000, height=1000, anti_alias=True, seed=1, rank=2, nranks=5, tile_rows=16)
    cfg.num_threads = os.cpu_count() or 8
    rows = np.array([y for y in range(1000) if (y // 16) % 5 == 2 and y // 16 < 62])
    for legacy in (0, 1):
        oracle.lib().oracle_set_legacy_square_light(legacy)
        t = time.time()
        r = oracle.render(s, dim, cam, cfg)
        oracle.lib().oracle_set_legacy_square_light(0)
        d = np.abs(blocks(r["rgb"][rows]) - blocks(gold[rows])).max(axis=2)
        print(f"{name}: {'400 uniform shadow-tested samples (dead code)' if legacy else 'subdivide_illumination (current)':48s} "
              f"blocks within 1 / 3 / 8 LSB of the gallery image: {100 * (d <= 1).mean():.1f} % / {100 * (d <= 3).mean():.1f} % / {100 * (d <= 8).mean():.1f} %, "
              f"worst block {d.max():.1f}, {time.time() - t:.0f} s", flush=True)
