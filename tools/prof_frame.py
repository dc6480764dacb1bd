#!/usr/bin/env python3
"""Render a few frames of one scene through the C ABI (no torch): the program to put
behind `rocprofv3 ... -- python3 tools/prof_frame.py`.
Usage: prof_frame.py [scene] [width] [height] [frames] [aa] [nocount]
(nocount: skip the first frame, which runs the work-counting build of the kernels)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
scn = importlib.import_module("harness.scenes")
rnd = importlib.import_module("rust-raytracer_amd.render")
name = sys.argv[1] if len(sys.argv) > 1 else "glass_monkey"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
aa = int(sys.argv[5]) if len(sys.argv) > 5 else 1
s = scn.Scene(name)
over = {}
if os.environ.get("RT_PROF_AA_THRESHOLD"):      # e.g. 0: the supersampled variant, every interior hit pixel gets aa_rays more
    over["aa_threshold"] = float(os.environ["RT_PROF_AA_THRESHOLD"])
dim, cam, cfg = s.view(width=w, height=h, anti_alias=bool(aa), seed=1, **over)
ds = rnd.DeviceScene(s)
if not (len(sys.argv) > 6 and sys.argv[6] == "nocount"):
    r = ds.render(dim, cam, cfg, want_aux=False, count_work=True)
    c = r["counters"]
    print({k: c[k] for k in c}, flush=True)
    print("per ray: instance %.1f box %.1f tri %.1f" % (c["instance_tests"] / c["rays_total"], c["box_tests"] / c["rays_total"], c["tri_tests"] / c["rays_total"]))
for i in range(frames):
    t = time.time()
    r = ds.render(dim, cam, cfg, want_aux=False)
    print(f"frame {i}: pass1 {r['ms']['pass1']:.2f} ms aa {r['ms']['aa']:.2f} ms wall {1e3*(time.time()-t):.1f} ms", flush=True)
