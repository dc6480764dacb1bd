"""The C-ABI library loads on a machine without a GPU, exports every entry point the
header declares, and its POD structs have the layout the ctypes mirror assumes."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", text)))


def test_header_functions_are_exported(render_mod):
    lib = render_mod.hip_lib()
    names = declared_functions()
    assert "rt_render" in names and "rt_scene_create" in names and len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"librt_amd.so does not export {n}"
    assert sorted(render_mod.EXPORTS) == names


def test_no_device_here_is_reported_not_faked(render_mod):
    # No compute call without a GPU: only the query.  (On a GPU box this returns >= 1.)
    assert render_mod.hip_lib().rt_device_count() >= 0


def test_struct_layouts_match_the_header(abi_mod):
    names = ["RtImageDim", "RtCamera", "RtRenderConfig", "RtCounters", "RtAux", "RtNode", "RtPrim", "RtCsgNode",
             "RtShader", "RtLight", "RtImage", "RtNormalMap", "RtMesh", "RtSceneDesc", "RtObjMesh"]
    src = '#include <stdio.h>\n#include "rt_amd.h"\nint main(){' + "".join(
        f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n in names:
        assert C.sizeof(getattr(abi_mod, n)) == sizes[n], n


def test_host_library_scene_registry(scenes_mod):
    names = scenes_mod.scene_names()
    for required in ["basic_setup_scene", "plane_test_1", "shadow_test_1", "light_room", "glass_monkey",
                     "many_balls", "many_balls_forest", "subtraction_outside1"]:
        assert required in names


def test_entry_point_modules_resolve():
    """Everything __graft_entry__.smoke() imports must be importable here too (it only runs on the GPU box)."""
    import importlib
    import re
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    names = re.findall(r'import_module\("([^"]+)"\)', src)
    assert "harness.scenes" in names and "rust-raytracer_amd.render" in names
    for n in names:
        importlib.import_module(n)
    from oracle import binding  # noqa: F401
