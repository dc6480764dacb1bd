"""Builds the native pieces in-tree (the .so files travel to the GPU box with the tree).

  csrc/librt_amd.so            HIP render path for gfx950 (hipcc cross-compiles without a GPU); the kernel
                               variants under csrc/variants/ are separate translation units built in parallel
  ../harness/libraytracer_host.so   (test harness) C++ mirror of the reference API + the transcribed test scenes
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(HERE, "csrc", "librt_amd.so")
HOST_LIB = os.path.join(ROOT, "harness", "libraytracer_host.so")

# -ffp-contract=off keeps the device arithmetic un-fused like the reference's (Rust never
# contracts a*b+c); RT_FP_CONTRACT=fast trades that for FMA throughput.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=" + os.environ.get("RT_FP_CONTRACT", "off")]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _deps(depfile):
    """prerequisites listed in a compiler-written dependency file (make syntax), or None when it is missing"""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    deps = []
    for part in text.split(":", 1)[-1].split():
        deps.append(part)
    return deps


def build_hip(force=False, jobs=None):
    """every csrc/*.hip and csrc/variants/*.hip compile to objects side by side (csrc/obj/), then link.
    An object is rebuilt when its source or any header it included last time (csrc/obj/*.d) is newer."""
    from concurrent.futures import ThreadPoolExecutor
    src_dir = os.path.join(HERE, "csrc")
    obj_dir = os.path.join(src_dir, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".h", ".hpp"))]
    headers.append(os.path.join(INCLUDE, "rt_amd.h"))
    var_dir = os.path.join(src_dir, "variants")
    sources = sorted(os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith(".hip")) + \
        sorted(os.path.join(var_dir, f) for f in os.listdir(var_dir) if f.endswith(".hip"))
    flags = [f for f in HIP_FLAGS if f != "-shared"]
    todo, objects = [], []
    for src in sources:
        stem = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(obj_dir, stem + ".o")
        dep = os.path.join(obj_dir, stem + ".d")
        objects.append(obj)
        known = _deps(dep)
        mine = [h for h in known if os.path.exists(h) and not h.startswith("/opt/")] if known is not None else headers
        if force or _newer(obj, mine + [src]):
            todo.append(["hipcc", *flags, "-c", "-MD", "-MF", dep, "-I" + INCLUDE, "-o", obj, src])
    if todo:
        jobs = jobs or min(len(todo), max(1, min(16, os.cpu_count() or 1)))
        with ThreadPoolExecutor(max_workers=jobs) as pool:
            list(pool.map(_run, todo))
    if todo or _newer(HIP_LIB, objects):
        _run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB, *objects, "-lz", "-pthread"])   # zlib: the PNG writer's deflate (in bands, on host threads)
    return HIP_LIB


def build_host(force=False):
    src_dir = os.path.join(ROOT, "harness")
    sources = [os.path.join(src_dir, f) for f in ("raytracer.cpp", "scenes.cpp", "raytracer.hpp")]
    sources.append(os.path.join(INCLUDE, "rt_amd.h"))
    if force or _newer(HOST_LIB, sources):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + INCLUDE, "-o", HOST_LIB,
              os.path.join(src_dir, "raytracer.cpp"), os.path.join(src_dir, "scenes.cpp")])
    return HOST_LIB


def build_all(force=False):
    build_host(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
