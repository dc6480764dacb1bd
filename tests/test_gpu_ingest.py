"""Scene ingest on the device (the step before the render path): the meshes' bounding hierarchies, the
vertical flip of images, and the OBJ reader behind Mesh::from_path (src/mesh.rs:383-414, tobj 0.1.7's
load_obj).  The host's and the device's preparation of a scene must give the same device scene, which
shows as identical frames *and* identical box / triangle test counts (those depend on every node of a
hierarchy); the device's OBJ reader must return the arrays a sequential reader returns."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST, DEVICE = 1, 2


def frames(render_mod, scene, ingest, width, height, desc=None):
    class Carrier:          # DeviceScene only reads .desc
        pass
    carrier = scene
    if desc is not None:
        carrier = Carrier()
        carrier.desc = C.cast(C.pointer(desc), C.c_void_p)
    ds = render_mod.DeviceScene(carrier, device=0, ingest=ingest)
    try:
        dim, cam, cfg = scene.view(width=width, height=height, anti_alias=True, seed=1)
        return ds.render(dim, cam, cfg, count_work=True)
    finally:
        ds.close()


def assert_same_frame(a, b):
    for k in ("rgb", "color", "hit_id", "distance", "aa_mask"):
        assert np.array_equal(a[k], b[k]), k
    for k in ("box_tests", "tri_tests", "instance_tests", "rays_total", "shade_evals"):
        assert a["counters"][k] == b["counters"][k], k


@pytest.mark.parametrize("name", ["glass_monkey", "monkey_1", "monkey_3", "light_room", "many_balls"])
def test_host_and_device_preparation_give_the_same_scene(name, render_mod, scene_cache):
    s = scene_cache(name)
    assert_same_frame(frames(render_mod, s, HOST, 200, 150), frames(render_mod, s, DEVICE, 200, 150))


def blob_mesh(n_lat, n_lon):
    """A bumpy closed surface of 2 * n_lat * n_lon triangles with f32 coordinates (so the LDS copy stays exact if it fits)."""
    lat = np.linspace(0.02, np.pi - 0.02, n_lat + 1)
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon, endpoint=False)
    la, lo = np.meshgrid(lat, lon, indexing="ij")
    r = 1.0 + 0.15 * np.sin(7 * la) * np.cos(5 * lo) + 0.05 * np.cos(23 * lo + 3 * la)
    p = np.stack([r * np.sin(la) * np.cos(lo), r * np.cos(la), r * np.sin(la) * np.sin(lo)], axis=-1)
    p = p.astype(np.float32).astype(np.float64).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(n_lat), np.arange(n_lon), indexing="ij")
    a = i * n_lon + j
    b = i * n_lon + (j + 1) % n_lon
    c = (i + 1) * n_lon + j
    d = (i + 1) * n_lon + (j + 1) % n_lon
    faces = np.concatenate([np.stack([a, c, b], -1).reshape(-1, 3), np.stack([b, c, d], -1).reshape(-1, 3)]).astype(np.uint32)
    return p, faces


def test_a_large_mesh_gets_the_same_hierarchy_on_the_device(render_mod, scene_cache, abi_mod):
    """glass_monkey with its mesh replaced by one of 131 072 faces (above RT_INGEST_AUTO_FACES, so the default path
    builds it on the device too)."""
    s = scene_cache("glass_monkey")
    src = C.cast(s.desc, C.POINTER(abi_mod.RtSceneDesc)).contents
    assert src.num_meshes == 1
    p, f = blob_mesh(256, 256)
    assert len(f) == 131072
    desc = abi_mod.RtSceneDesc()
    C.memmove(C.byref(desc), C.byref(src), C.sizeof(desc))
    mesh = abi_mod.RtMesh(len(p), 0, 0, len(f), 0, 0, 0, 0)
    pf = np.ascontiguousarray(p.reshape(-1))
    ff = np.ascontiguousarray(f.reshape(-1))
    desc.meshes = C.pointer(mesh)
    desc.mesh_f64_count = pf.size
    desc.mesh_f64 = pf.ctypes.data_as(C.POINTER(C.c_double))
    desc.mesh_u32_count = ff.size
    desc.mesh_u32 = ff.ctypes.data_as(C.POINTER(C.c_uint32))
    host = frames(render_mod, s, HOST, 160, 120, desc)
    dev = frames(render_mod, s, DEVICE, 160, 120, desc)
    auto = frames(render_mod, s, 0, 160, 120, desc)
    assert host["counters"]["tri_tests"] > 0 and (host["hit_id"] != 0xFFFFFFFF).any()
    assert_same_frame(host, dev)
    assert_same_frame(host, auto)


def test_coinciding_centroids_and_tiny_meshes(render_mod, scene_cache, abi_mod):
    """A mesh of identical triangles (no split plane exists: halves by position), and meshes of 1 to 5 faces."""
    s = scene_cache("glass_monkey")
    src = C.cast(s.desc, C.POINTER(abi_mod.RtSceneDesc)).contents
    tri = np.array([[-0.5, -0.5, 0.0], [0.5, -0.5, 0.0], [0.0, 0.5, 0.0]])
    cases = [(np.tile(tri, (9, 1)), np.arange(27, dtype=np.uint32).reshape(9, 3))]
    for n in (1, 2, 3, 5):
        p = np.concatenate([tri + [0.0, 0.0, 0.1 * k] for k in range(n)])
        cases.append((p, np.arange(3 * n, dtype=np.uint32).reshape(n, 3)))
    for p, f in cases:
        desc = abi_mod.RtSceneDesc()
        C.memmove(C.byref(desc), C.byref(src), C.sizeof(desc))
        mesh = abi_mod.RtMesh(len(p), 0, 0, len(f), 0, 0, 0, 0)
        pf = np.ascontiguousarray(p.reshape(-1), dtype=np.float64)
        ff = np.ascontiguousarray(f.reshape(-1), dtype=np.uint32)
        desc.meshes = C.pointer(mesh)
        desc.mesh_f64_count = pf.size
        desc.mesh_f64 = pf.ctypes.data_as(C.POINTER(C.c_double))
        desc.mesh_u32_count = ff.size
        desc.mesh_u32 = ff.ctypes.data_as(C.POINTER(C.c_uint32))
        assert_same_frame(frames(render_mod, s, HOST, 96, 72, desc), frames(render_mod, s, DEVICE, 96, 72, desc))


# ---------------------------------------------------------------- OBJ text
_libc = C.CDLL(None)
_libc.strtof.restype = C.c_float
_libc.strtof.argtypes = [C.c_char_p, C.c_void_p]


def strtof(tok):
    return float(_libc.strtof(tok, None))


def sequential_obj(text):
    """What a line-by-line reader with tobj 0.1.7's rules returns (the rules of harness/raytracer.cpp, Mesh::from_path)."""
    v, vt, vn = [], [], []
    positions, normals, texcoords, faces = [], [], [], []
    index, offset, count, has_faces = {}, 0, 0, False

    def resolve(i, n):
        return i - 1 if i > 0 else n + i if i < 0 else -1
    for line in text.split(b"\n"):
        tok = line.split()
        if not tok:
            continue
        if tok[0] in (b"v", b"vn"):
            vals = [strtof(t) for t in tok[1:4]] + [0.0] * (3 - len(tok[1:4]))
            (v if tok[0] == b"v" else vn).append(vals)
        elif tok[0] == b"vt":
            vt.append([strtof(t) for t in tok[1:3]] + [0.0] * (2 - len(tok[1:3])))
        elif tok[0] == b"f":
            corner = []
            for t in tok[1:]:
                parts = t.split(b"/")
                iv = int(parts[0])
                it = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                inn = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                key = (resolve(iv, len(v)), resolve(it, len(vt)), resolve(inn, len(vn)))
                if key not in index:
                    positions.append(v[key[0]])
                    if vt and key[1] >= 0:
                        texcoords.append(vt[key[1]])
                    if vn and key[2] >= 0:
                        normals.append(vn[key[2]])
                    index[key] = count
                    count += 1
                corner.append(index[key])
            for i in range(1, len(corner) - 1):
                faces.append([offset + corner[0], offset + corner[i], offset + corner[i + 1]])
            has_faces = True
        elif tok[0] in (b"o", b"g"):
            if has_faces:
                offset, count, index, has_faces = offset + count, 0, {}, False
    return {"positions": np.array(positions, dtype=np.float64).reshape(-1, 3), "normals": np.array(normals, dtype=np.float64).reshape(-1, 3),
            "texcoords": np.array(texcoords, dtype=np.float64).reshape(-1, 2), "faces": np.array(faces, dtype=np.uint32).reshape(-1, 3)}


def assert_same_mesh(got, want):
    for k in ("positions", "normals", "texcoords", "faces"):
        assert got[k].shape == want[k].shape, (k, got[k].shape, want[k].shape)
        assert np.array_equal(got[k].view(np.uint64) if got[k].dtype == np.float64 else got[k],
                              want[k].view(np.uint64) if want[k].dtype == np.float64 else want[k]), k


def test_obj_reader_returns_the_mesh_the_harness_reads(render_mod, scene_cache, abi_mod):
    """assets/models/monkey2.obj through the device reader == the arrays Mesh::from_path put into glass_monkey's description."""
    s = scene_cache("glass_monkey")
    d = C.cast(s.desc, C.POINTER(abi_mod.RtSceneDesc)).contents
    m = d.meshes[0]
    f64 = np.ctypeslib.as_array(d.mesh_f64, shape=(d.mesh_f64_count,))
    u32 = np.ctypeslib.as_array(d.mesh_u32, shape=(d.mesh_u32_count,))
    want = {"positions": f64[m.positions_off:m.positions_off + 3 * m.num_positions].reshape(-1, 3),
            "normals": f64[m.normals_off:m.normals_off + 3 * m.num_normals].reshape(-1, 3),
            "texcoords": f64[m.texcoords_off:m.texcoords_off + 2 * m.num_texcoords].reshape(-1, 2),
            "faces": u32[m.faces_off:m.faces_off + 3 * m.num_faces].reshape(-1, 3)}
    text = open(os.path.join(ROOT, "assets", "models", "monkey2.obj"), "rb").read()
    got = render_mod.obj_parse(text)
    assert got["faces"].shape[0] == 967 or got["faces"].shape[0] == m.num_faces
    assert_same_mesh(got, {k: np.ascontiguousarray(v) for k, v in want.items()})
    assert_same_mesh(got, sequential_obj(text))


def test_obj_text_to_frame_on_the_device(render_mod, scene_cache, abi_mod):
    """The whole step before the path on the device: monkey2.obj as text -> rt_obj_parse -> the mesh of glass_monkey's
    description -> hierarchy built on the device -> the frame the harness-read, host-built scene renders."""
    s = scene_cache("glass_monkey")
    src = C.cast(s.desc, C.POINTER(abi_mod.RtSceneDesc)).contents
    m = render_mod.obj_parse(open(os.path.join(ROOT, "assets", "models", "monkey2.obj"), "rb").read())
    f64 = np.ascontiguousarray(np.concatenate([m["positions"].reshape(-1), m["normals"].reshape(-1), m["texcoords"].reshape(-1)]))
    u32 = np.ascontiguousarray(m["faces"].reshape(-1))
    npos, nnorm, ntex = len(m["positions"]), len(m["normals"]), len(m["texcoords"])
    mesh = abi_mod.RtMesh(npos, nnorm, ntex, len(m["faces"]), 0, 3 * npos, 3 * npos + 3 * nnorm, 0)
    desc = abi_mod.RtSceneDesc()
    C.memmove(C.byref(desc), C.byref(src), C.sizeof(desc))
    desc.meshes = C.pointer(mesh)
    desc.mesh_f64_count = f64.size
    desc.mesh_f64 = f64.ctypes.data_as(C.POINTER(C.c_double))
    desc.mesh_u32_count = u32.size
    desc.mesh_u32 = u32.ctypes.data_as(C.POINTER(C.c_uint32))
    assert_same_frame(frames(render_mod, s, HOST, 240, 180), frames(render_mod, s, DEVICE, 240, 180, desc))


TRICKY = b"""# comments, blank lines, CRLF, tabs, quads and a pentagon, negative indices, groups, corners without vt / vn
mtllib nothing.mtl
o first
v 0 0 0
v 1.5 0 0\r
v 1.5e0 2.25 -0
v\t0 2.25 0.1
v 0.1 0.2 0.30000001192092896
vt 0 0
vt 1 0
vt 1 1
vn 0 0 1
s off
f 1/1/1 2/2/1 3/3/1 4/1/1
f 1//1 2//1 5//1
f -5 -4 -1

g second
v 1e-3 16777217 3.4028234e38
v -7.5 .5 5.
v 1.17549435E-38 1E+2 -2.5e-1
v 0.1 0.7 123456789.123456789
f 6 7 8 9 1
f 6/1 7/2 8/3
g empty_group
o third
f 1/1/1 2/2/1 3/3/1
f 9/3/1 8/3/1 7/3/1
"""


def test_obj_reader_follows_the_sequential_rules(render_mod):
    assert_same_mesh(render_mod.obj_parse(TRICKY), sequential_obj(TRICKY))
    # no trailing newline, a single triangle
    small = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3"
    got = render_mod.obj_parse(small)
    assert got["faces"].tolist() == [[0, 1, 2]] and got["positions"].shape == (3, 3) and got["normals"].shape == (0, 3)


def test_obj_reader_rounds_every_literal_once(render_mod):
    """Literals chosen at random, and literals that sit exactly halfway between two f32 values as decimals
    (where rounding through f64 first would go wrong): all must come out as strtof rounds them."""
    rng = np.random.default_rng(5)
    toks = []
    for _ in range(3000):
        kind = rng.integers(0, 4)
        if kind == 0:
            toks.append(b"%.6f" % rng.uniform(-2, 2))
        elif kind == 1:
            toks.append(b"%.9g" % (rng.uniform(-1, 1) * 10.0 ** rng.integers(-8, 9)))
        elif kind == 2:
            toks.append(b"%d" % rng.integers(-10**9, 10**9))
        else:   # the decimal expansion of a midpoint between two neighbouring f32 values, cut or nudged
            f = np.float32(rng.uniform(0.5, 4.0))
            mid = (float(f) + float(np.nextafter(f, np.float32(10.0)))) / 2.0
            digits = b"%.30f" % mid
            toks.append(digits[:rng.integers(12, 31)] + (b"1" if rng.integers(0, 2) else b""))
    while len(toks) % 3:
        toks.append(b"0")
    lines = [b"v " + b" ".join(toks[i:i + 3]) for i in range(0, len(toks), 3)]
    lines.append(b"f " + b" ".join(b"%d" % (i + 1) for i in range(len(lines))))
    text = b"\n".join(lines) + b"\n"
    got = render_mod.obj_parse(text)
    want = np.array([strtof(t) for t in toks], dtype=np.float64).reshape(-1, 3)
    assert np.array_equal(got["positions"].view(np.uint64), want.view(np.uint64))


def random_obj(seed):
    """A random OBJ text: pools that grow between faces, polygons of 3 to 6 corners in all four corner forms,
    positive and negative indices, repeated corners, groups with and without faces, comments, blank lines, CRLF."""
    rng = np.random.default_rng(seed)
    lines, nv, nvt, nvn = [], 0, 0, 0
    fmt = [b"%.6f", b"%.3f", b"%g", b"%.9e"]

    def num():
        return fmt[rng.integers(0, len(fmt))] % float(rng.uniform(-5, 5))

    def add_pool(kind, n):
        nonlocal nv, nvt, nvn
        for _ in range(n):
            k = 2 if kind == b"vt" else 3
            lines.append(kind + b" " + b" ".join(num() for _ in range(k)))
        if kind == b"v":
            nv += n
        elif kind == b"vt":
            nvt += n
        else:
            nvn += n
    add_pool(b"v", 4)
    for _ in range(int(rng.integers(20, 120))):
        what = rng.integers(0, 10)
        if what == 0:
            add_pool(b"v", int(rng.integers(1, 6)))
        elif what == 1:
            add_pool(b"vt", int(rng.integers(1, 4)))
        elif what == 2:
            add_pool(b"vn", int(rng.integers(1, 4)))
        elif what == 3:
            lines.append([b"g part", b"o thing", b"# note", b"", b"usemtl m", b"s 1"][rng.integers(0, 6)])
        else:
            form = int(rng.integers(0, 4))          # v, v/vt, v//vn, v/vt/vn
            if form in (1, 3) and nvt == 0:
                form = 0
            if form in (2, 3) and nvn == 0:
                form = 0 if form == 2 else 1
            if form in (1, 3) and nvt == 0:
                form = 0
            corners = []
            pool = [int(rng.integers(0, nv)) for _ in range(4)]      # few distinct vertices: corners repeat across faces
            for _ in range(int(rng.integers(3, 7))):
                def ref(i, n):
                    return b"%d" % (i + 1 if rng.integers(0, 2) else i - n)
                iv = pool[rng.integers(0, 4)]
                t = ref(iv, nv)
                if form in (1, 3):
                    t += b"/" + ref(int(rng.integers(0, nvt)), nvt)
                if form == 2:
                    t += b"/"
                if form in (2, 3):
                    t += b"/" + ref(int(rng.integers(0, nvn)), nvn)
                corners.append(t)
            lines.append(b"f " + (b" " if rng.integers(0, 2) else b"\t").join(corners))
    eol = b"\r\n" if seed % 3 == 0 else b"\n"
    return eol.join(lines) + (eol if seed % 2 else b"")


@pytest.mark.parametrize("seed", range(24))
def test_obj_reader_on_random_texts(render_mod, seed):
    text = random_obj(seed)
    if b"\nf " not in text:
        pytest.skip("no face drawn")
    assert_same_mesh(render_mod.obj_parse(text), sequential_obj(text))


def test_obj_reader_errors(render_mod):
    with pytest.raises(render_mod.RtError) as e:
        render_mod.obj_parse(b"v 0 0 0\nv 1 0 0\nv 0 1 0\n")
    assert e.value.code == 1 and "no faces" in str(e.value)
    with pytest.raises(render_mod.RtError) as e:
        render_mod.obj_parse(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n")
    assert e.value.code == 1
    # what the reference's parser (tobj 0.1.7) fails on or reads differently is refused, not guessed at:
    # line / point elements, a v / vt / vn line with too few numbers, a token that is no decimal number
    tri = b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
    for bad in (tri + b"l 1 2\n", tri + b"p 1\n", b"v 0 0\n" + tri, b"vt 0.5\n" + tri, b"vn 0 1\n" + tri,
                b"v 1.0f 0 0\n" + tri, b"v 0x10 0 0\n" + tri, b"v - 0 0\n" + tri, b"v 1 2 # three\n" + tri):
        with pytest.raises(render_mod.RtError) as e:
            render_mod.obj_parse(bad)
        assert e.value.code == 2, bad
    # a material line after faces opens no new model when no material library was read (none ever is: the
    # text is all the reader gets), so the corners before and after it share their vertices
    got = render_mod.obj_parse(b"v 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\nusemtl b\nf 3 2 1\n")
    assert got["positions"].shape == (3, 3) and got["faces"].tolist() == [[0, 1, 2], [2, 1, 0]]


def test_unknown_ingest_mode_is_refused(render_mod, scene_cache):
    with pytest.raises(render_mod.RtError) as e:
        render_mod.DeviceScene(scene_cache("glass_monkey"), device=0, ingest=7)
    assert e.value.code == 1 and "ingest" in str(e.value)
