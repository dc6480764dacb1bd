"""ctypes binding of the HIP render path (csrc/librt_amd.so, C ABI in include/rt_amd.h).

There is no CPU fallback: if the library is missing, or no gfx950 device is visible,
the calls raise."""
import ctypes as C
import os

import numpy as np

from ._abi import RT_OK, RtAux, RtCamera, RtImageDim, RtObjMesh, RtRenderConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("RT_AMD_LIB") or os.path.join(_HERE, "csrc", "librt_amd.so")   # RT_AMD_LIB: another build of the same library (kernel tuning experiments)

EXPORTS = ["rt_device_count", "rt_last_error", "rt_scene_create", "rt_scene_destroy", "rt_render",
           "rt_render_device", "rt_owned_rows_capacity", "rt_last_timings", "rt_frame_status", "rt_render_device_noalloc",
           "rt_write_png", "rt_write_png_device", "rt_set_progress", "rt_scene_create_with", "rt_obj_parse", "rt_obj_release"]

INGEST_AUTO, INGEST_HOST, INGEST_DEVICE = 0, 1, 2

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_double)

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt_amd status {code}: {msg}")
        self.code = code


def hip_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is missing: the HIP render path was not built "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(HIP_LIB)
        lib.rt_device_count.restype = C.c_int
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_scene_create.restype = C.c_int
        lib.rt_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_render.restype = C.c_int
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(RtImageDim), C.POINTER(RtCamera), C.POINTER(RtRenderConfig),
                                  C.c_void_p, C.POINTER(RtAux)]
        lib.rt_render_device.restype = C.c_int
        lib.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RtImageDim), C.POINTER(RtCamera), C.POINTER(RtRenderConfig),
                                         C.c_void_p, C.POINTER(RtAux), C.c_void_p]
        lib.rt_owned_rows_capacity.restype = C.c_uint32
        lib.rt_owned_rows_capacity.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        lib.rt_last_timings.restype = C.c_int
        lib.rt_last_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.rt_render_device_noalloc.restype = C.c_int
        lib.rt_render_device_noalloc.argtypes = [C.c_void_p, C.POINTER(RtImageDim), C.POINTER(RtCamera), C.POINTER(RtRenderConfig),
                                                 C.c_void_p, C.c_void_p]
        lib.rt_frame_status.restype = C.c_int
        lib.rt_frame_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.rt_scene_create_with.restype = C.c_int
        lib.rt_scene_create_with.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.rt_obj_parse.restype = C.c_int
        lib.rt_obj_parse.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.POINTER(RtObjMesh)]
        lib.rt_obj_release.argtypes = [C.POINTER(RtObjMesh)]
        lib.rt_write_png.restype = C.c_int
        lib.rt_write_png.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p]
        lib.rt_write_png_device.restype = C.c_int
        lib.rt_write_png_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_void_p]
        lib.rt_set_progress.restype = C.c_int
        lib.rt_set_progress.argtypes = [C.c_void_p, PROGRESS_FN, C.c_void_p]
        _lib = lib
    return _lib


def obj_parse(text, device=0):
    """rt_obj_parse: OBJ text (bytes) -> dict of numpy arrays (positions (n,3), normals (n,3), texcoords (n,2), faces (n,3))."""
    lib = hip_lib()
    m = RtObjMesh()
    _check(lib.rt_obj_parse(device, text, len(text), C.byref(m)))
    try:
        def take(ptr, rows, cols, dtype):
            if rows == 0:
                return np.zeros((0, cols), dtype=dtype)
            return np.ctypeslib.as_array(ptr, shape=(rows, cols)).astype(dtype, copy=True)
        return {"positions": take(m.positions, m.num_positions, 3, np.float64), "normals": take(m.normals, m.num_normals, 3, np.float64),
                "texcoords": take(m.texcoords, m.num_texcoords, 2, np.float64), "faces": take(m.faces, m.num_faces, 3, np.uint32)}
    finally:
        lib.rt_obj_release(C.byref(m))


def write_png(rgb, file_name, device=0):
    """rt_write_png: an (H, W, 3) uint8 array as "<file_name>.png" (src/lib.rs:41-46)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    _check(hip_lib().rt_write_png(device, rgb.ctypes.data, w, h, os.fsencode(file_name)))


def write_png_device(d_rgb_ptr, width, height, file_name, device=0, stream_ptr=None):
    """rt_write_png_device: the same for a frame in device memory."""
    _check(hip_lib().rt_write_png_device(device, C.c_void_p(d_rgb_ptr), width, height, os.fsencode(file_name),
                                         C.c_void_p(stream_ptr) if stream_ptr else None))


def _check(rc):
    if rc != RT_OK:
        raise RtError(rc, hip_lib().rt_last_error().decode())


class DeviceScene:
    """A scene flattened and resident in HBM (rt_scene_create)."""

    def __init__(self, scene, device=0, ingest=None):
        lib = hip_lib()
        self.scene = scene   # keeps the description alive
        self.device = device
        handle = C.c_void_p()
        if ingest is None:
            _check(lib.rt_scene_create(scene.desc, device, C.byref(handle)))
        else:   # rt_scene_create_with: where the hierarchies are built and the images flipped (INGEST_*)
            _check(lib.rt_scene_create_with(scene.desc, device, ingest, C.byref(handle)))
        self._handle = handle

    @property
    def handle(self):
        return self._handle

    def render(self, dim, cam, cfg, want_aux=True, count_work=False):
        """rt_render: host-memory outputs as numpy arrays."""
        lib = hip_lib()
        w, h = dim.width, dim.height
        rows = lib.rt_owned_rows_capacity(h, cfg.nranks, cfg.tile_rows)
        rgb = np.zeros((rows, w, 3), dtype=np.uint8)
        aux = RtAux()
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
        aux.count_work = 1 if count_work else 0
        _check(lib.rt_render(self._handle, C.byref(dim), C.byref(cam), C.byref(cfg), rgb.ctypes.data, C.byref(aux)))
        out["counters"] = aux.counters.as_dict()
        out["ms"] = {"pass1": aux.ms_pass1, "aa": aux.ms_aa, "encode": aux.ms_encode}
        return out

    def render_device(self, dim, cam, cfg, d_rgb_ptr, stream_ptr=None, d_aux=None):
        """rt_render_device: asynchronous, outputs stay in device memory (raw pointers)."""
        lib = hip_lib()
        _check(lib.rt_render_device(self._handle, C.byref(dim), C.byref(cam), C.byref(cfg), C.c_void_p(d_rgb_ptr),
                                    C.byref(d_aux) if d_aux is not None else None,
                                    C.c_void_p(stream_ptr) if stream_ptr else None))

    def render_device_noalloc(self, dim, cam, cfg, d_rgb_ptr, stream_ptr=None):
        """rt_render_device_noalloc: like render_device, workspace in the library's static device array."""
        _check(hip_lib().rt_render_device_noalloc(self._handle, C.byref(dim), C.byref(cam), C.byref(cfg), C.c_void_p(d_rgb_ptr),
                                                  C.c_void_p(stream_ptr) if stream_ptr else None))

    def set_progress(self, fn):
        """rt_set_progress: fn(stage, done, total, elapsed_seconds) after every rt_render frame (None: off)."""
        self._progress = PROGRESS_FN(lambda user, stage, done, total, t: fn(stage, done, total, t)) if fn else PROGRESS_FN()
        _check(hip_lib().rt_set_progress(self._handle, self._progress, None))

    def last_timings(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _check(hip_lib().rt_last_timings(self._handle, C.byref(a), C.byref(b), C.byref(c)))
        return {"pass1": a.value, "aa": b.value, "encode": c.value}

    def frame_dropped(self):
        """rt_frame_status: waits for the last rt_render_device frame; rays it (or frames since the last call) lost."""
        n = C.c_uint64()
        rc = hip_lib().rt_frame_status(self._handle, C.byref(n))
        if rc not in (RT_OK, 5):
            _check(rc)
        return int(n.value)

    def close(self):
        if self._handle:
            hip_lib().rt_scene_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
