// build.rs — compiles the HIP render path for gfx950 and links it.
// Every translation unit of the library — csrc/*.hip (frame driver, scene ingest, output stage) and
// csrc/variants/*.hip (the kernel variants) — becomes an object file; the objects are linked into
// librt_amd.so next to the other build outputs, with zlib (the PNG writer's deflate) and the thread
// library (its bands run on host threads).  The same sources, flags and libraries as
// rust-raytracer_amd/build.py, which builds the library for the Python-driven tests.
use std::{env, fs, path::{Path, PathBuf}, process::Command};

const HIP_FLAGS: &[&str] = &["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"];
const LINK_LIBS: &[&str] = &["-lz", "-pthread"];

fn hip_sources(dir: &Path) -> Vec<PathBuf> {
    let mut v: Vec<PathBuf> = fs::read_dir(dir)
        .unwrap_or_else(|_| panic!("{}", dir.display()))
        .map(|e| e.unwrap().path())
        .filter(|p| p.extension().map_or(false, |e| e == "hip"))
        .collect();
    v.sort();
    v
}

fn main() {
    let repo = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let csrc = repo.join("rust-raytracer_amd").join("csrc");
    let include = repo.join("include");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());

    let mut sources = hip_sources(&csrc);
    sources.extend(hip_sources(&csrc.join("variants")));

    let mut objects = Vec::new();
    for src in &sources {
        let obj = out.join(src.file_stem().unwrap()).with_extension("o");
        let status = Command::new("hipcc")
            .args(HIP_FLAGS)
            .arg("-c")
            .arg(format!("-I{}", include.display()))
            .arg("-o").arg(&obj).arg(src)
            .status().expect("hipcc not found");
        assert!(status.success(), "hipcc failed on {}", src.display());
        objects.push(obj);
    }
    let so = out.join("librt_amd.so");
    let status = Command::new("hipcc")
        .args(&["--offload-arch=gfx950", "-shared", "-fPIC", "-o"]).arg(&so).args(&objects).args(LINK_LIBS)
        .status().expect("hipcc not found");
    assert!(status.success(), "linking librt_amd.so failed");

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", include.join("rt_amd.h").display());
}
