"""ctypes mirror of the POD structs in include/rt_amd.h that cross the C ABI by value
or by pointer from Python.  RtSceneDesc itself is never built in Python: the host
harness (harness/scenes.cpp) produces it and Python only carries the pointer."""
import ctypes as C

RT_NONE = 0xFFFFFFFF
RT_OK = 0


class RtImageDim(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32)]


class RtCamera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3), ("fov_y", C.c_double)]


class RtRenderConfig(C.Structure):
    _fields_ = [
        ("num_threads", C.c_uint64),
        ("workload_split", C.c_uint32),
        ("anti_alias", C.c_uint32),
        ("aa_threshold", C.c_double),
        ("aa_rays", C.c_uint32),
        ("recursion_depth", C.c_uint32),
        ("interactive", C.c_uint32),
        ("_pad", C.c_uint32),
        ("seed", C.c_uint64),
        ("rank", C.c_uint32),
        ("nranks", C.c_uint32),
        ("tile_rows", C.c_uint32),
        ("_pad2", C.c_uint32),
    ]

    def copy(self):
        out = RtRenderConfig()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(self))
        return out


class RtCounters(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_aa", C.c_uint64),
        ("aa_pixels", C.c_uint64),
        ("instance_tests", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("shade_evals", C.c_uint64),
        ("stack_overflows", C.c_uint64),
        ("flops", C.c_uint64),
        ("schedule", C.c_uint64),
        ("attempts", C.c_uint64),
        ("box_tests_f32", C.c_uint64),
        ("csg_events", C.c_uint64),
        ("flops_f32", C.c_uint64),
    ]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_}
        d["rays_total"] = d["rays_primary"] + d["rays_secondary"] + d["rays_shadow"] + d["rays_aa"]
        return d


class RtAux(C.Structure):
    _fields_ = [
        ("color", C.c_void_p),
        ("distance", C.c_void_p),
        ("hit_id", C.c_void_p),
        ("aa_mask", C.c_void_p),
        ("counters", RtCounters),
        ("count_work", C.c_uint32),
        ("_pad", C.c_uint32),
        ("ms_pass1", C.c_double),
        ("ms_aa", C.c_double),
        ("ms_encode", C.c_double),
    ]


class RtMat4(C.Structure):
    _fields_ = [("m", C.c_double * 16)]


class RtNode(C.Structure):
    _fields_ = [("parent", C.c_uint32), ("prim", C.c_uint32), ("shader", C.c_uint32), ("subtree_size", C.c_uint32), ("trans", RtMat4)]


class RtPrim(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("ref", C.c_uint32), ("count", C.c_uint32), ("_pad", C.c_uint32), ("p", C.c_double * 12)]


class RtCsgNode(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("prim", C.c_uint32), ("first_child", C.c_uint32), ("num_children", C.c_uint32), ("trans", RtMat4)]


class RtShader(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("ref", C.c_uint32), ("first_child", C.c_uint32), ("num_children", C.c_uint32), ("p", C.c_double * 10)]


class RtLight(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("position", C.c_double * 3), ("color", C.c_double * 3),
                ("power", C.c_double), ("falloff", C.c_double * 3), ("size", C.c_double)]


class RtImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channels", C.c_uint32), ("_pad", C.c_uint32), ("offset", C.c_uint64)]


class RtNormalMap(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("image", C.c_uint32), ("depth", C.c_double)]


class RtMesh(C.Structure):
    _fields_ = [("num_positions", C.c_uint32), ("num_normals", C.c_uint32), ("num_texcoords", C.c_uint32), ("num_faces", C.c_uint32),
                ("positions_off", C.c_uint64), ("normals_off", C.c_uint64), ("texcoords_off", C.c_uint64), ("faces_off", C.c_uint64)]


class RtObjMesh(C.Structure):
    _fields_ = [("num_positions", C.c_uint32), ("num_normals", C.c_uint32), ("num_texcoords", C.c_uint32), ("num_faces", C.c_uint32),
                ("positions", C.POINTER(C.c_double)), ("normals", C.POINTER(C.c_double)), ("texcoords", C.POINTER(C.c_double)),
                ("faces", C.POINTER(C.c_uint32))]


class RtSceneDesc(C.Structure):
    """Read-only view used by tests to inspect a description produced by the host layer."""
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("num_nodes", C.c_uint32), ("nodes", C.POINTER(RtNode)),
        ("num_prims", C.c_uint32), ("prims", C.POINTER(RtPrim)),
        ("num_points", C.c_uint32), ("points", C.POINTER(C.c_double)),
        ("num_csg", C.c_uint32), ("csg", C.POINTER(RtCsgNode)),
        ("num_csg_children", C.c_uint32), ("csg_children", C.POINTER(C.c_uint32)),
        ("num_shaders", C.c_uint32), ("shaders", C.POINTER(RtShader)),
        ("num_shader_children", C.c_uint32), ("shader_child_idx", C.POINTER(C.c_uint32)),
        ("shader_child_weight", C.POINTER(C.c_double)),
        ("num_lights", C.c_uint32), ("lights", C.POINTER(RtLight)),
        ("ambient_color", C.c_double * 3),
        ("ambient_power", C.c_double),
        ("num_images", C.c_uint32), ("images", C.POINTER(RtImage)),
        ("texel_bytes", C.c_uint64), ("texels", C.POINTER(C.c_uint8)),
        ("num_normal_maps", C.c_uint32), ("normal_maps", C.POINTER(RtNormalMap)),
        ("num_meshes", C.c_uint32), ("meshes", C.POINTER(RtMesh)),
        ("mesh_f64_count", C.c_uint64), ("mesh_f64", C.POINTER(C.c_double)),
        ("mesh_u32_count", C.c_uint64), ("mesh_u32", C.POINTER(C.c_uint32)),
        ("has_background", C.c_uint32),
        ("background_image", C.c_uint32),
        ("background_trans", RtMat4),
    ]
