"""The Rust FFI crate (ffi/) cannot be compiled here (no cargo / rustc), so its text is checked
against the C header: every #[repr(C)] struct field for field, every constant, the extern block."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
LIB_RS = open(os.path.join(ROOT, "ffi", "src", "lib.rs")).read()
DESCRIBE_RS = open(os.path.join(ROOT, "ffi", "src", "describe.rs")).read()

C_TO_RUST = {"uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "int": "c_int", "void": "c_void", "char": "c_char"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def c_structs():
    """name -> [(field, rust-spelled type)] from `typedef struct X { ... } X;`."""
    text = strip_c_comments(HEADER)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            const = decl.startswith("const ")
            if const:
                decl = decl[len("const "):]
            base, rest = decl.split(" ", 1)
            for d in rest.split(","):
                d = d.strip()
                ptr = d.startswith("*") or base.endswith("*")
                d = d.lstrip("* ")
                b = C_TO_RUST.get(base.rstrip("*"), base.rstrip("*"))
                arr = re.match(r"(\w+)\[(\d+)\]$", d)
                if arr:
                    fields.append((arr.group(1), f"[{b}; {arr.group(2)}]"))
                elif ptr:
                    fields.append((d, f"*{'const' if const else 'mut'} {b}"))
                else:
                    fields.append((d, b))
        out[m.group(1)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\n\}", LIB_RS, flags=re.S):
        fields = []
        for line in m.group(2).split("\n"):
            line = line.split("//")[0].strip().rstrip(",")
            if not line:
                continue
            fm = re.match(r"pub (r#)?(\w+): (.+)$", line)
            assert fm, line
            fields.append((fm.group(2), fm.group(3).strip()))
        out[m.group(1)] = fields
    return out


def test_every_struct_matches_the_header_field_for_field():
    c, r = c_structs(), rust_structs()
    assert len(c) >= 15 and "RtSceneDesc" in c and "RtAux" in c
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name] == r[name], (name, [x for x in zip(c[name], r[name]) if x[0] != x[1]][:3], len(c[name]), len(r[name]))


def c_constants():
    text = strip_c_comments(HEADER)
    vals = {}
    for m in re.finditer(r"enum\s*\{(.*?)\}", text, flags=re.S):
        nxt = 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                nxt = int(v, 0)
            else:
                k = item
            vals[k] = nxt
            nxt += 1
    for m in re.finditer(r"#define\s+(RT_[A-Z_]+)\s+(.+)", text):
        expr = m.group(2).strip().replace("u", "").replace("U", "")
        if re.fullmatch(r"[0-9xXa-fA-F()<> ]+", expr):
            vals[m.group(1)] = eval(expr)
    return vals


def test_constants_have_the_header_values():
    c = c_constants()
    assert c["RT_ERR_CAPACITY"] == 5 and c["RT_PRIM_CSG"] == 11 and c["RT_NONE"] == 0xFFFFFFFF
    r = {}
    for m in re.finditer(r"pub const (RT_\w+): \w+ = ([^;]+);", LIB_RS):
        r[m.group(1)] = eval(m.group(2).replace("_", ""))
    assert sorted(k for k in c if k != "RT_AMD_H") == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for k, v in r.items():
        assert c[k] == v, k


def test_extern_block_lists_the_header_functions():
    text = strip_c_comments(HEADER)
    c = {m.group(1): len([a for a in m.group(2).split(",") if a.strip() not in ("", "void")])
         for m in re.finditer(r"\b(rt_[a-z_]+)\s*\(([^)]*)\)\s*;", text)}
    block = re.search(r'extern "C" \{(.*?)\n\}', LIB_RS, flags=re.S).group(1)
    r = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
         for m in re.finditer(r"pub fn (rt_\w+)\(([^)]*)\)", block, flags=re.S)}
    assert c == r


def test_describe_covers_every_scene_type_of_the_harness():
    hpp = open(os.path.join(ROOT, "harness", "raytracer.hpp")).read()
    described = set(re.findall(r"impl Describe\w+ for (\w+)", DESCRIBE_RS))
    described |= set(re.search(r"composite_as_prim!\(([^)]*)\)", DESCRIBE_RS).group(1).replace(" ", "").split(","))
    for t in ["Sphere", "Rectangle", "Plane", "Triangle", "Polygon", "Cube", "Tetrahedron", "Polyhedron", "OneWay", "Inverted", "Mesh",
              "BaseShape", "SubtractShape", "AndShape", "OrShape", "XorShape", "MultiAndShape",
              "PhongShader", "TextureShader", "ReflectionShader", "TranslucentShader", "MixShader", "CompositeShader", "ChainShader",
              "NormalMapShader", "PointLight", "SquareLight", "DirectionLight", "SceneNode", "ImageTexture", "NormalMap", "BumpMap"]:
        assert t in described, t
        assert re.search(r"\b" + t + r"\b", hpp), f"{t} is not a type of the C++ harness"


REFERENCE = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is not present on this machine")
def test_describe_names_only_fields_the_reference_types_have():
    """`self.<field>` in `impl Describe.. for T` must be a field of the reference's `struct T`."""
    ref = ""
    for d, _, files in os.walk(REFERENCE):
        for f in files:
            if f.endswith(".rs"):
                ref += open(os.path.join(d, f)).read() + "\n"
    for m in re.finditer(r"impl Describe\w+ for (\w+) \{(.*?)\n\}", DESCRIBE_RS, flags=re.S):
        t, body = m.group(1), m.group(2)
        sm = re.search(r"pub struct " + t + r"\s*\{(.*?)\n\}", ref, flags=re.S)
        assert sm, t
        fields = set(re.findall(r"(?:pub )?(\w+)\s*:", sm.group(1)))
        for f in set(re.findall(r"self\.(\w+)\b(?!\()", body)):
            assert f in fields, (t, f, sorted(fields))


def test_build_rs_builds_what_build_py_builds():
    """ffi/build.rs must compile the library's translation units with the flags, and link the libraries, that
    rust-raytracer_amd/build.py does: same source directories, same hipcc flags, same -l options."""
    build_rs = open(os.path.join(ROOT, "ffi", "build.rs")).read()
    build_py = open(os.path.join(ROOT, "rust-raytracer_amd", "build.py")).read()
    # source directories: every *.hip under csrc/ and csrc/variants/, in both scripts
    assert 'hip_sources(&csrc)' in build_rs and 'hip_sources(&csrc.join("variants"))' in build_rs
    assert re.search(r'os\.listdir\(src_dir\) if f\.endswith\("\.hip"\)', build_py) and re.search(r'os\.listdir\(var_dir\) if f\.endswith\("\.hip"\)', build_py)
    assert 'e == "hip"' in build_rs
    # compile flags
    rs_flags = re.search(r"const HIP_FLAGS: &\[&str\] = &\[(.*?)\];", build_rs, flags=re.S).group(1)
    rs_flags = set(re.findall(r'"([^"]+)"', rs_flags))
    py_flags = re.search(r"HIP_FLAGS = \[(.*?)\]\n", build_py, flags=re.S).group(1)
    py_flags = set(re.findall(r'"(-[^"]+)"', py_flags)) - {"-shared"}
    py_flags = {f if f != "-ffp-contract=" else "-ffp-contract=off" for f in py_flags}      # (the script's default)
    assert rs_flags == py_flags, (rs_flags, py_flags)
    # link libraries
    rs_libs = set(re.findall(r'"(-l\w+|-pthread)"', re.search(r"const LINK_LIBS: &\[&str\] = &\[(.*?)\];", build_rs).group(1)))
    py_link = re.search(r'_run\(\["hipcc", "--offload-arch=gfx950", "-shared".*?\]\)', build_py, flags=re.S).group(0)
    py_libs = set(re.findall(r'"(-l\w+|-pthread)"', py_link))
    assert rs_libs == py_libs and rs_libs, (rs_libs, py_libs)


def test_every_declared_entry_point_is_defined_in_a_compiled_unit():
    """Every rt_* function of the header (= of the crate's extern block, see above) has its definition in one of
    the csrc/*.hip files both build scripts compile."""
    text = strip_c_comments(HEADER)
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\([^)]*\)\s*;", text))
    csrc = os.path.join(ROOT, "rust-raytracer_amd", "csrc")
    units = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
    assert len(units) >= 3
    defined = set()
    for u in units:
        src = open(u).read()
        defined |= set(re.findall(r"^(?:int|void|uint32_t|const char\*)\s+(rt_[a-z_]+)\s*\([^;{]*\)\s*\{", src, flags=re.M))
    missing = declared - defined
    assert not missing, missing
