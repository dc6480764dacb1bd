//! Rust side of `include/rt_amd.h`: the POD structs of the scene description, the render
//! parameters and the `extern "C"` entry points of `librt_amd.so`, field for field.
//!
//! The reference renderer keeps its public API (`render`, `render_with_config`, src/render.rs:89-102);
//! the body of `render_with_config` becomes "describe the scene (describe.rs) -> rt_scene_create ->
//! rt_render".  This crate is the thin layer that call goes through.
//!
//! UNCOMPILED where this repository is built (no Rust toolchain); `tests/test_rust_ffi.py` compares
//! every `#[repr(C)]` struct, constant and function below with the C header.
#![allow(non_camel_case_types, non_upper_case_globals, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const RT_NONE: u32 = 0xFFFF_FFFF;
pub const RT_ABI_VERSION: u32 = 1;
pub const RT_NOALLOC_SCRATCH_BYTES: u32 = 256 << 20;
pub const RT_NOALLOC_MAX_PIXELS: u32 = 32768;

// status codes
pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID_ARG: c_int = 1;
pub const RT_ERR_UNSUPPORTED: c_int = 2;
pub const RT_ERR_DEVICE: c_int = 3;
pub const RT_ERR_NO_DEVICE: c_int = 4;
pub const RT_ERR_CAPACITY: c_int = 5;

// primitives
pub const RT_PRIM_SPHERE: u32 = 0;
pub const RT_PRIM_RECTANGLE: u32 = 1;
pub const RT_PRIM_PLANE: u32 = 2;
pub const RT_PRIM_TRIANGLE: u32 = 3;
pub const RT_PRIM_POLYGON: u32 = 4;
pub const RT_PRIM_CUBE: u32 = 5;
pub const RT_PRIM_TETRAHEDRON: u32 = 6;
pub const RT_PRIM_POLYHEDRON: u32 = 7;
pub const RT_PRIM_ONEWAY: u32 = 8;
pub const RT_PRIM_INVERTED: u32 = 9;
pub const RT_PRIM_MESH: u32 = 10;
pub const RT_PRIM_CSG: u32 = 11;

// composite shapes
pub const RT_CSG_BASE: u32 = 0;
pub const RT_CSG_SUBTRACT: u32 = 1;
pub const RT_CSG_AND: u32 = 2;
pub const RT_CSG_OR: u32 = 3;
pub const RT_CSG_XOR: u32 = 4;
pub const RT_CSG_MULTI_AND: u32 = 5;

// shaders
pub const RT_SHADER_PHONG: u32 = 0;
pub const RT_SHADER_TEXTURE: u32 = 1;
pub const RT_SHADER_REFLECTION: u32 = 2;
pub const RT_SHADER_TRANSLUCENT: u32 = 3;
pub const RT_SHADER_MIX: u32 = 4;
pub const RT_SHADER_COMPOSITE: u32 = 5;
pub const RT_SHADER_CHAIN: u32 = 6;
pub const RT_SHADER_NORMAL_MAP: u32 = 7;

// lights
pub const RT_LIGHT_POINT: u32 = 0;
pub const RT_LIGHT_SQUARE: u32 = 1;
pub const RT_LIGHT_DIRECTION: u32 = 2;

// normal maps
pub const RT_NMAP_NORMAL: u32 = 0;
pub const RT_NMAP_BUMP: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtMat4 {
    pub m: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtPrim {
    pub kind: u32,
    pub r#ref: u32,
    pub count: u32,
    pub _pad: u32,
    pub p: [f64; 12],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCsgNode {
    pub kind: u32,
    pub prim: u32,
    pub first_child: u32,
    pub num_children: u32,
    pub trans: RtMat4,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtShader {
    pub kind: u32,
    pub r#ref: u32,
    pub first_child: u32,
    pub num_children: u32,
    pub p: [f64; 10],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtLight {
    pub kind: u32,
    pub _pad: u32,
    pub position: [f64; 3],
    pub color: [f64; 3],
    pub power: f64,
    pub falloff: [f64; 3],
    pub size: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtImage {
    pub width: u32,
    pub height: u32,
    pub channels: u32,
    pub _pad: u32,
    pub offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtNormalMap {
    pub kind: u32,
    pub image: u32,
    pub depth: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtMesh {
    pub num_positions: u32,
    pub num_normals: u32,
    pub num_texcoords: u32,
    pub num_faces: u32,
    pub positions_off: u64,
    pub normals_off: u64,
    pub texcoords_off: u64,
    pub faces_off: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtNode {
    pub parent: u32,
    pub prim: u32,
    pub shader: u32,
    pub subtree_size: u32,
    pub trans: RtMat4,
}

#[repr(C)]
pub struct RtSceneDesc {
    pub abi_version: u32,
    pub num_nodes: u32,
    pub nodes: *const RtNode,
    pub num_prims: u32,
    pub prims: *const RtPrim,
    pub num_points: u32,
    pub points: *const f64,
    pub num_csg: u32,
    pub csg: *const RtCsgNode,
    pub num_csg_children: u32,
    pub csg_children: *const u32,
    pub num_shaders: u32,
    pub shaders: *const RtShader,
    pub num_shader_children: u32,
    pub shader_child_idx: *const u32,
    pub shader_child_weight: *const f64,
    pub num_lights: u32,
    pub lights: *const RtLight,
    pub ambient_color: [f64; 3],
    pub ambient_power: f64,
    pub num_images: u32,
    pub images: *const RtImage,
    pub texel_bytes: u64,
    pub texels: *const u8,
    pub num_normal_maps: u32,
    pub normal_maps: *const RtNormalMap,
    pub num_meshes: u32,
    pub meshes: *const RtMesh,
    pub mesh_f64_count: u64,
    pub mesh_f64: *const f64,
    pub mesh_u32_count: u64,
    pub mesh_u32: *const u32,
    pub has_background: u32,
    pub background_image: u32,
    pub background_trans: RtMat4,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtImageDim {
    pub width: u32,
    pub height: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCamera {
    pub origin: [f64; 3],
    pub target: [f64; 3],
    pub up: [f64; 3],
    pub fov_y: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtRenderConfig {
    pub num_threads: u64,
    pub workload_split: u32,
    pub anti_alias: u32,
    pub aa_threshold: f64,
    pub aa_rays: u32,
    pub recursion_depth: u32,
    pub interactive: u32,
    pub _pad: u32,
    pub seed: u64,
    pub rank: u32,
    pub nranks: u32,
    pub tile_rows: u32,
    pub _pad2: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCounters {
    pub rays_primary: u64,
    pub rays_secondary: u64,
    pub rays_shadow: u64,
    pub rays_aa: u64,
    pub aa_pixels: u64,
    pub instance_tests: u64,
    pub box_tests: u64,
    pub tri_tests: u64,
    pub shade_evals: u64,
    pub stack_overflows: u64,
    pub flops: u64,
    pub schedule: u64,
    pub attempts: u64,
    pub box_tests_f32: u64,
    pub csg_events: u64,
    pub flops_f32: u64,
}

#[repr(C)]
pub struct RtAux {
    pub color: *mut f64,
    pub distance: *mut f64,
    pub hit_id: *mut u32,
    pub aa_mask: *mut u8,
    pub counters: RtCounters,
    pub count_work: u32,
    pub _pad: u32,
    pub ms_pass1: f64,
    pub ms_aa: f64,
    pub ms_encode: f64,
}

/// Where `rt_scene_create_with` prepares a scene (hierarchies of the meshes, image flip).
pub const RT_INGEST_AUTO: u32 = 0;
pub const RT_INGEST_HOST: u32 = 1;
pub const RT_INGEST_DEVICE: u32 = 2;
pub const RT_INGEST_AUTO_FACES: u32 = 65536;

/// `Mesh::from_path` (src/mesh.rs:383-414) on the device: the arrays are the library's until `rt_obj_release`.
#[repr(C)]
pub struct RtObjMesh {
    pub num_positions: u32,
    pub num_normals: u32,
    pub num_texcoords: u32,
    pub num_faces: u32,
    pub positions: *mut f64,
    pub normals: *mut f64,
    pub texcoords: *mut f64,
    pub faces: *mut u32,
}

/// Stages reported to an `RtProgressFn` (src/progress_tracker.rs:37-42).
pub const RT_PROGRESS_START: u32 = 0;
pub const RT_PROGRESS_PASS1: u32 = 1;
pub const RT_PROGRESS_START_AA: u32 = 2;
pub const RT_PROGRESS_AA: u32 = 3;
pub const RT_PROGRESS_END: u32 = 4;
pub type RtProgressFn = Option<unsafe extern "C" fn(user: *mut c_void, stage: u32, done: u64, total: u64, elapsed_seconds: f64)>;

/// Opaque: the flattened scene, resident in device memory.
pub enum RtScene {}

extern "C" {
    pub fn rt_device_count() -> c_int;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_scene_create(desc: *const RtSceneDesc, device: c_int, out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_render(scene: *const RtScene, dim: *const RtImageDim, cam: *const RtCamera, cfg: *const RtRenderConfig,
                     rgb_out: *mut u8, aux: *mut RtAux) -> c_int;
    pub fn rt_render_device(scene: *const RtScene, dim: *const RtImageDim, cam: *const RtCamera, cfg: *const RtRenderConfig,
                            d_rgb_out: *mut u8, d_aux: *mut RtAux, hip_stream: *mut c_void) -> c_int;
    pub fn rt_frame_status(scene: *const RtScene, dropped: *mut u64) -> c_int;
    pub fn rt_render_device_noalloc(scene: *const RtScene, dim: *const RtImageDim, cam: *const RtCamera,
                                    cfg: *const RtRenderConfig, d_rgb_out: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rt_owned_rows_capacity(height: u32, nranks: u32, tile_rows: u32) -> u32;
    pub fn rt_last_timings(scene: *const RtScene, ms_pass1: *mut f64, ms_aa: *mut f64, ms_encode: *mut f64) -> c_int;
    pub fn rt_scene_create_with(desc: *const RtSceneDesc, device: c_int, ingest: u32, out: *mut *mut RtScene) -> c_int;
    pub fn rt_obj_parse(device: c_int, text: *const c_char, bytes: u64, out: *mut RtObjMesh) -> c_int;
    pub fn rt_obj_release(mesh: *mut RtObjMesh);
    pub fn rt_write_png(device: c_int, rgb: *const u8, width: u32, height: u32, file_name: *const c_char) -> c_int;
    pub fn rt_write_png_device(device: c_int, d_rgb: *const u8, width: u32, height: u32, file_name: *const c_char,
                               hip_stream: *mut c_void) -> c_int;
    pub fn rt_set_progress(scene: *mut RtScene, callback: RtProgressFn, user: *mut c_void) -> c_int;
}

/// The message of the last failing call on this thread.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned()
}

/// A scene resident on one GPU; dropped with `rt_scene_destroy`.
pub struct DeviceScene(*mut RtScene);

impl DeviceScene {
    pub fn new(desc: &RtSceneDesc, device: i32) -> Result<DeviceScene, String> {
        let mut s: *mut RtScene = std::ptr::null_mut();
        let rc = unsafe { rt_scene_create(desc, device as c_int, &mut s) };
        if rc != RT_OK { return Err(last_error()); }
        Ok(DeviceScene(s))
    }

    /// `render_with_config` (src/render.rs:99-312): W*H*3 bytes, rows top first, RGB interleaved.
    pub fn render(&self, dim: RtImageDim, cam: &RtCamera, cfg: &RtRenderConfig) -> Result<Vec<u8>, String> {
        let mut rgb = vec![0u8; dim.width as usize * dim.height as usize * 3];
        let rc = unsafe { rt_render(self.0, &dim, cam, cfg, rgb.as_mut_ptr(), std::ptr::null_mut()) };
        if rc != RT_OK { return Err(last_error()); }
        Ok(rgb)
    }
}

impl Drop for DeviceScene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.0) }
    }
}
