"""Exploration: how close are full-size GPU renders to the reference's gallery images?
Prints block-mean statistics (16x16 blocks) gallery vs render and render(seed 1) vs render(seed 2)."""
import importlib, os, sys
import numpy as np
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
scenes = importlib.import_module("harness.scenes")
render = importlib.import_module("rust-raytracer_amd.render")

def blocks(img, b=16):
    h, w, _ = img.shape
    return img[:h // b * b, :w // b * b].astype(float).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3))

for name in sys.argv[1:] or ["light_room", "monkey_1"]:
    gold = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", f"gallery_{name}.png")).convert("RGB"))
    s = scenes.Scene(name)
    ds = render.DeviceScene(s, device=0)
    out = []
    for seed in (1, 2):
        dim, cam, cfg = s.view(width=gold.shape[1], height=gold.shape[0], anti_alias=True, seed=seed)
        r = ds.render(dim, cam, cfg)
        out.append(r)
        print(name, "seed", seed, "ms", r["ms"], flush=True)
    a, b = out[0]["rgb"], out[1]["rgb"]
    Image.fromarray(a).save(os.path.join(ROOT, "gpurun_out", f"render_{name}.png"))
    for label, x, y in (("seed1 vs seed2", a, b), ("gallery vs seed1", gold, a)):
        d = np.abs(blocks(x) - blocks(y))
        px = np.abs(x.astype(int) - y.astype(int)).max(axis=2)
        print(f"{name}: {label}: block-mean |diff| max {d.max():.3f} mean {d.mean():.4f} p99 {np.quantile(d, 0.99):.3f};"
              f" pixels: equal {100 * (px == 0).mean():.2f}% <=1 {100 * (px <= 1).mean():.2f}% <=2 {100 * (px <= 2).mean():.2f}% max {px.max()}", flush=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"gallery_blocks_{name}.npz"), a=a, b=b, mask=out[0]["aa_mask"], hit=out[0]["hit_id"])
    ds.close()
