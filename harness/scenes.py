"""Named scenes: the reference's test scenes as transcribed in harness/scenes.cpp.

`Scene(name)` builds the scene in the C++ host layer and exposes the pointer to its
flat RtSceneDesc plus the image size / camera / config the reference test uses."""
import ctypes as C
import os

import importlib

_abi = importlib.import_module("rust-raytracer_amd._abi")
RtCamera, RtImageDim, RtRenderConfig = _abi.RtCamera, _abi.RtImageDim, _abi.RtRenderConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
HOST_LIB = os.path.join(_HERE, "libraytracer_host.so")

_lib = None


def host_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_LIB)
        lib.rth_last_error.restype = C.c_char_p
        lib.rth_scene_count.restype = C.c_int
        lib.rth_scene_name.restype = C.c_char_p
        lib.rth_scene_name.argtypes = [C.c_int]
        lib.rth_scene_build.restype = C.c_int
        lib.rth_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rth_scene_desc.restype = C.c_void_p
        lib.rth_scene_desc.argtypes = [C.c_void_p]
        lib.rth_scene_view.restype = C.c_int
        lib.rth_scene_view.argtypes = [C.c_void_p, C.POINTER(RtImageDim), C.POINTER(RtCamera), C.POINTER(RtRenderConfig)]
        lib.rth_scene_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def scene_names():
    lib = host_lib()
    return [lib.rth_scene_name(i).decode() for i in range(lib.rth_scene_count())]


class Scene:
    """A described scene.  `.desc` is a `const RtSceneDesc*` valid while this object lives."""

    def __init__(self, name, asset_root=REPO_ROOT):
        lib = host_lib()
        handle = C.c_void_p()
        rc = lib.rth_scene_build(name.encode(), asset_root.encode(), C.byref(handle))
        if rc != 0:
            raise RuntimeError(f"scene '{name}': {lib.rth_last_error().decode()}")
        self.name = name
        self._handle = handle
        self.desc = C.c_void_p(lib.rth_scene_desc(handle))
        self.dim = RtImageDim()
        self.cam = RtCamera()
        self.cfg = RtRenderConfig()
        lib.rth_scene_view(handle, C.byref(self.dim), C.byref(self.cam), C.byref(self.cfg))

    def view(self, width=None, height=None, anti_alias=None, seed=None, **cfg_overrides):
        """(dim, cam, cfg) of the reference test, optionally at another size / AA setting."""
        dim = RtImageDim(width or self.dim.width, height or self.dim.height)
        cfg = self.cfg.copy()
        if anti_alias is not None:
            cfg.anti_alias = 1 if anti_alias else 0
        if seed is not None:
            cfg.seed = seed
        for k, v in cfg_overrides.items():
            setattr(cfg, k, v)
        return dim, self.cam, cfg

    def __del__(self):
        try:
            if self._handle:
                host_lib().rth_scene_free(self._handle)
                self._handle = None
        except Exception:
            pass
