"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from the product package."""
import ctypes as C
import importlib
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
_abi = importlib.import_module("rust-raytracer_amd._abi")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} is missing: run `make -C oracle`")
        l = C.CDLL(LIB)
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.c_void_p, C.POINTER(_abi.RtImageDim), C.POINTER(_abi.RtCamera),
                                    C.POINTER(_abi.RtRenderConfig), C.c_void_p, C.POINTER(_abi.RtAux)]
        l.oracle_set_faithful_cost.argtypes = [C.c_int]
        l.oracle_last_assert_failures.restype = C.c_ulonglong
        l.oracle_last_assert_failures.argtypes = []
        _lib = l
    return _lib


def render(scene, dim, cam, cfg, want_aux=True, faithful_cost=False):
    """Render with the CPU restatement.  Returns dict(rgb, color, distance, hit_id, aa_mask, counters)."""
    l = lib()
    l.oracle_set_faithful_cost(1 if faithful_cost else 0)
    w, h = dim.width, dim.height
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    aux = _abi.RtAux()
    out = {"rgb": rgb}
    if want_aux:
        out["color"] = np.zeros((h, w, 3), dtype=np.float64)
        out["distance"] = np.zeros((h, w), dtype=np.float64)
        out["hit_id"] = np.zeros((h, w), dtype=np.uint32)
        out["aa_mask"] = np.zeros((h, w), dtype=np.uint8)
        aux.color = out["color"].ctypes.data
        aux.distance = out["distance"].ctypes.data
        aux.hit_id = out["hit_id"].ctypes.data
        aux.aa_mask = out["aa_mask"].ctypes.data
    rc = l.oracle_render(scene.desc, C.byref(dim), C.byref(cam), C.byref(cfg), rgb.ctypes.data, C.byref(aux))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed with status {rc}")
    out["counters"] = aux.counters.as_dict()
    out["assert_failures"] = int(l.oracle_last_assert_failures())
    return out
