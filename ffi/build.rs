// build.rs — compiles the HIP render path for gfx950 and links it.
// Every translation unit under csrc/ (rt_api.hip and csrc/variants/*.hip) becomes an object file;
// the objects are linked into librt_amd.so next to the other build outputs.
use std::{env, fs, path::PathBuf, process::Command};

fn main() {
    let repo = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let csrc = repo.join("rust-raytracer_amd").join("csrc");
    let include = repo.join("include");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());

    let mut sources = vec![csrc.join("rt_api.hip")];
    for entry in fs::read_dir(csrc.join("variants")).expect("csrc/variants") {
        let p = entry.unwrap().path();
        if p.extension().map_or(false, |e| e == "hip") {
            sources.push(p);
        }
    }
    sources.sort();

    let mut objects = Vec::new();
    for src in &sources {
        let obj = out.join(src.file_stem().unwrap()).with_extension("o");
        let status = Command::new("hipcc")
            .args(&["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c"])
            .arg(format!("-I{}", include.display()))
            .arg("-o").arg(&obj).arg(src)
            .status().expect("hipcc not found");
        assert!(status.success(), "hipcc failed on {}", src.display());
        objects.push(obj);
    }
    let so = out.join("librt_amd.so");
    let status = Command::new("hipcc")
        .args(&["--offload-arch=gfx950", "-shared", "-fPIC", "-o"]).arg(&so).args(&objects)
        .status().expect("hipcc not found");
    assert!(status.success(), "linking librt_amd.so failed");

    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-changed={}", csrc.display());
    println!("cargo:rerun-if-changed={}", include.join("rt_amd.h").display());
}
