//! `describe()` for every scene type of the reference renderer: the one host-side addition the GPU
//! render path needs.
//!
//! `Scene` is a tree of boxed trait objects with no introspection (src/scene.rs:47-53, 109-120;
//! src/geometry.rs:15-30; src/composite_shape.rs:125-127; src/shader.rs:20-30; src/light.rs:12-44),
//! so a flat description can only be produced from inside: each trait gains one supertrait from this
//! file (no existing signature changes), e.g.
//!
//! ```text
//! pub trait Intersectable: IntersectableClone + DescribePrim { ... }     // src/geometry.rs:15
//! pub trait Compositable: Intersectable + Transformable + ... + DescribeCsg { ... }   // src/composite_shape.rs:125
//! pub trait Shadable: ShadableClone + DescribeShader { ... }              // src/shader.rs:20
//! pub trait Lightable: LightableClone + DescribeLight { ... }             // src/light.rs:12
//! pub trait Traceable: TraceableClone + DescribeNode { ... }              // src/scene.rs:109
//! pub trait TextureMappable: TextureMappableClone + DescribeTexture { ... }   // src/texture.rs:6
//! pub trait NormalMappable: NormalMappableClone + DescribeNormalMap { ... }   // src/normal_map.rs:8
//! ```
//!
//! This file is meant to be dropped into the reference crate as `src/describe.rs` (`mod describe;` in
//! src/lib.rs, `extern crate rt_amd_sys;`): it reads private fields of the reference's types.
//! UNCOMPILED where this repository is built; the C++ twin that the tests exercise is
//! `harness/raytracer.hpp` (same builder, same field-for-field description).
//!
//! Conventions of the description (include/rt_amd.h): matrices column-major like `dmat4!`; light
//! colours after the constructors' `Color::normalize`; images as decoded, top row first (the
//! reference keeps them flipped, so rows are written back in reverse); nodes in DFS pre-order.

use std::collections::HashMap;

use euler::{DMat4, DVec3};
use image::{GrayImage, RgbImage};
use rt_amd_sys::*;

use color::Color;
use composite_shape::{AndShape, BaseShape, MultiAndShape, OrShape, SubtractShape, XorShape};
use geometry::Transformable;
use light::{DirectionLight, PointLight, SquareLight};
use mesh::Mesh;
use normal_map::{BumpMap, NormalMap};
use primitive::{Cube, Inverted, OneWay, Plane, Polygon, Polyhedron, Rectangle, Sphere, Tetrahedron, Triangle};
use render::{CameraConfig, ImageDimension, RenderConfig};
use scene::{Scene, SceneNode};
use shader::{ChainShader, CompositeShader, MixShader, NormalMapShader, PhongShader, ReflectionShader, TextureShader, TranslucentShader};
use texture::ImageTexture;

// ------------------------------------------------------------------------------------------------
/// Append-only arrays of one `RtSceneDesc`.  Shared sub-objects (the same `Arc` image behind two
/// textures, as `AssetManager` hands them out, src/asset_manager.rs:29-60) are written once.
#[derive(Default)]
pub struct SceneDescBuilder {
    pub nodes: Vec<RtNode>,
    pub prims: Vec<RtPrim>,
    pub points: Vec<f64>,
    pub csg: Vec<RtCsgNode>,
    pub csg_children: Vec<u32>,
    pub shaders: Vec<RtShader>,
    pub shader_child_idx: Vec<u32>,
    pub shader_child_weight: Vec<f64>,
    pub lights: Vec<RtLight>,
    pub images: Vec<RtImage>,
    pub texels: Vec<u8>,
    pub normal_maps: Vec<RtNormalMap>,
    pub meshes: Vec<RtMesh>,
    pub mesh_f64: Vec<f64>,
    pub mesh_u32: Vec<u32>,
    pub ambient_color: [f64; 3],
    pub ambient_power: f64,
    pub background: Option<(u32, RtMat4)>,
    image_by_address: HashMap<usize, u32>,
}

fn mat(m: &DMat4) -> RtMat4 {
    // euler's matrices name their elements m<column><row>; RtMat4 is column-major
    RtMat4 { m: [m.m00, m.m01, m.m02, m.m03, m.m10, m.m11, m.m12, m.m13,
                 m.m20, m.m21, m.m22, m.m23, m.m30, m.m31, m.m32, m.m33] }
}
fn v3(v: DVec3) -> [f64; 3] { [v.x, v.y, v.z] }
fn rgb(c: Color) -> [f64; 3] { [c.red, c.green, c.blue] }

impl SceneDescBuilder {
    pub fn new() -> SceneDescBuilder { SceneDescBuilder::default() }

    pub fn prim(&mut self, kind: u32, r: u32, count: u32, p: &[f64]) -> u32 {
        let mut rec = RtPrim { kind, r#ref: r, count, _pad: 0, p: [0.0; 12] };
        rec.p[..p.len()].copy_from_slice(p);
        self.prims.push(rec);
        (self.prims.len() - 1) as u32
    }

    pub fn csg_node(&mut self, kind: u32, prim: u32, trans: &DMat4, children: &[u32]) -> u32 {
        let first_child = self.csg_children.len() as u32;
        self.csg_children.extend_from_slice(children);
        self.csg.push(RtCsgNode { kind, prim, first_child, num_children: children.len() as u32, trans: mat(trans) });
        (self.csg.len() - 1) as u32
    }

    pub fn shader(&mut self, kind: u32, r: u32, p: &[f64], children: &[(f64, u32)]) -> u32 {
        let first_child = self.shader_child_idx.len() as u32;
        for &(w, c) in children {
            self.shader_child_weight.push(w);
            self.shader_child_idx.push(c);
        }
        let mut rec = RtShader { kind, r#ref: r, first_child, num_children: children.len() as u32, p: [0.0; 10] };
        rec.p[..p.len()].copy_from_slice(p);
        self.shaders.push(rec);
        (self.shaders.len() - 1) as u32
    }

    /// `rows` = the image as the reference holds it (flipped at load, src/texture.rs:46): written
    /// bottom row first so that the description carries it as decoded.
    pub fn image(&mut self, address: usize, width: u32, height: u32, channels: u32, rows: &[u8]) -> u32 {
        if let Some(&i) = self.image_by_address.get(&address) { return i; }
        let offset = self.texels.len() as u64;
        let stride = (width * channels) as usize;
        for y in (0..height as usize).rev() {
            self.texels.extend_from_slice(&rows[y * stride..(y + 1) * stride]);
        }
        self.images.push(RtImage { width, height, channels, _pad: 0, offset });
        let i = (self.images.len() - 1) as u32;
        self.image_by_address.insert(address, i);
        i
    }

    pub fn begin_node(&mut self, parent: u32, trans: &DMat4) -> u32 {
        self.nodes.push(RtNode { parent, prim: RT_NONE, shader: 0, subtree_size: 1, trans: mat(trans) });
        (self.nodes.len() - 1) as u32
    }
    pub fn end_node(&mut self, me: u32) -> u32 {
        self.nodes[me as usize].subtree_size = self.nodes.len() as u32 - me;
        me
    }

    /// Pointers into this builder: valid while it lives unchanged.
    pub fn finish(&self) -> RtSceneDesc {
        let (has_background, background_image, background_trans) = match self.background {
            Some((i, t)) => (1, i, t),
            None => (0, 0, RtMat4 { m: [0.0; 16] }),
        };
        RtSceneDesc {
            abi_version: RT_ABI_VERSION,
            num_nodes: self.nodes.len() as u32, nodes: self.nodes.as_ptr(),
            num_prims: self.prims.len() as u32, prims: self.prims.as_ptr(),
            num_points: (self.points.len() / 3) as u32, points: self.points.as_ptr(),
            num_csg: self.csg.len() as u32, csg: self.csg.as_ptr(),
            num_csg_children: self.csg_children.len() as u32, csg_children: self.csg_children.as_ptr(),
            num_shaders: self.shaders.len() as u32, shaders: self.shaders.as_ptr(),
            num_shader_children: self.shader_child_idx.len() as u32,
            shader_child_idx: self.shader_child_idx.as_ptr(), shader_child_weight: self.shader_child_weight.as_ptr(),
            num_lights: self.lights.len() as u32, lights: self.lights.as_ptr(),
            ambient_color: self.ambient_color, ambient_power: self.ambient_power,
            num_images: self.images.len() as u32, images: self.images.as_ptr(),
            texel_bytes: self.texels.len() as u64, texels: self.texels.as_ptr(),
            num_normal_maps: self.normal_maps.len() as u32, normal_maps: self.normal_maps.as_ptr(),
            num_meshes: self.meshes.len() as u32, meshes: self.meshes.as_ptr(),
            mesh_f64_count: self.mesh_f64.len() as u64, mesh_f64: self.mesh_f64.as_ptr(),
            mesh_u32_count: self.mesh_u32.len() as u64, mesh_u32: self.mesh_u32.as_ptr(),
            has_background, background_image, background_trans,
        }
    }
}

// ------------------------------------------------------------------------------------------------
/// Supertrait of `Intersectable`: returns the index of the primitive's `RtPrim`.
pub trait DescribePrim { fn describe(&self, b: &mut SceneDescBuilder) -> u32; }
/// Supertrait of `Compositable`: returns the index of the shape's `RtCsgNode`.
pub trait DescribeCsg { fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32; }
/// Supertrait of `Shadable`: returns the index of the shader's `RtShader`.
pub trait DescribeShader { fn describe(&self, b: &mut SceneDescBuilder) -> u32; }
/// Supertrait of `Lightable`: appends the light's `RtLight`.
pub trait DescribeLight { fn describe(&self, b: &mut SceneDescBuilder); }
/// Supertrait of `Traceable`: appends the node and its subtree in DFS pre-order.
pub trait DescribeNode { fn describe(&self, parent: u32, b: &mut SceneDescBuilder) -> u32; }
/// Supertrait of `TextureMappable`: returns the index of the texture's `RtImage` (3 channels).
pub trait DescribeTexture { fn describe(&self, b: &mut SceneDescBuilder) -> u32; }
/// Supertrait of `NormalMappable`: returns the index of the map's `RtNormalMap`.
pub trait DescribeNormalMap { fn describe(&self, b: &mut SceneDescBuilder) -> u32; }

// ---- primitives (src/primitive.rs, src/primitive/*.rs, src/mesh.rs)
impl DescribePrim for Sphere {                                        // sphere.rs:4-6
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.prim(RT_PRIM_SPHERE, RT_NONE, 0, &[self.radius]) }
}
impl DescribePrim for Rectangle {                                     // plane.rs:64-67
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.prim(RT_PRIM_RECTANGLE, RT_NONE, 0, &[self.width, self.height]) }
}
fn plane_params(p: &Plane) -> [f64; 11] {                             // plane.rs:5-10, the values after with_surface_scale
    [p.origin.x, p.origin.y, p.origin.z, p.normal.x, p.normal.y, p.normal.z,
     p.tangent.x, p.tangent.y, p.tangent.z, p.surface_scale.x, p.surface_scale.y]
}
impl DescribePrim for Plane {
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.prim(RT_PRIM_PLANE, RT_NONE, 0, &plane_params(self)) }
}
impl DescribePrim for Triangle {                                      // plane.rs:101-108
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let v = &self.vertices;
        b.prim(RT_PRIM_TRIANGLE, RT_NONE, 0, &[v[0].x, v[0].y, v[0].z, v[1].x, v[1].y, v[1].z, v[2].x, v[2].y, v[2].z])
    }
}
impl DescribePrim for Polygon {                                       // plane.rs:168-172: the vertices as given to from_vertices
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let first = (b.points.len() / 3) as u32;
        for v in self.vertices.iter() { b.points.extend_from_slice(&v3(*v)); }
        b.prim(RT_PRIM_POLYGON, first, self.vertices.len() as u32, &[])
    }
}
impl DescribePrim for Cube {                                          // cube.rs:4-9
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.prim(RT_PRIM_CUBE, RT_NONE, 0, &[self.length]) }
}
impl DescribePrim for Tetrahedron {                                   // cube.rs:91-97
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.prim(RT_PRIM_TETRAHEDRON, RT_NONE, 0, &[self.size]) }
}
impl DescribePrim for Polyhedron {                                    // polyhedron.rs:4-6: its planes, consecutive
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let first = b.prims.len() as u32;
        for p in self.planes.iter() { b.prim(RT_PRIM_PLANE, RT_NONE, 0, &plane_params(p)); }
        b.prim(RT_PRIM_POLYHEDRON, first, self.planes.len() as u32, &[])
    }
}
impl DescribePrim for OneWay {                                        // primitive.rs:19-46
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { let inner = self.primitive.describe(b); b.prim(RT_PRIM_ONEWAY, inner, 0, &[]) }
}
impl DescribePrim for Inverted {                                      // primitive.rs:48-81
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { let inner = self.primitive.describe(b); b.prim(RT_PRIM_INVERTED, inner, 0, &[]) }
}
impl DescribePrim for Mesh {                                          // mesh.rs:342-350: unified vertices, triangle indices
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let m = RtMesh {
            num_positions: self.positions.len() as u32, num_normals: self.vertex_normals.len() as u32,
            num_texcoords: self.tex_coords.len() as u32, num_faces: self.faces.len() as u32,
            positions_off: b.mesh_f64.len() as u64,
            normals_off: (b.mesh_f64.len() + 3 * self.positions.len()) as u64,
            texcoords_off: (b.mesh_f64.len() + 3 * self.positions.len() + 3 * self.vertex_normals.len()) as u64,
            faces_off: b.mesh_u32.len() as u64,
        };
        for p in self.positions.iter() { b.mesh_f64.extend_from_slice(&v3(*p)); }
        for n in self.vertex_normals.iter() { b.mesh_f64.extend_from_slice(&v3(*n)); }
        for t in self.tex_coords.iter() { b.mesh_f64.extend_from_slice(&[t.x, t.y]); }
        for &(i1, i2, i3) in self.faces.iter() { b.mesh_u32.extend_from_slice(&[i1 as u32, i2 as u32, i3 as u32]); }
        b.meshes.push(m);
        b.prim(RT_PRIM_MESH, (b.meshes.len() - 1) as u32, 0, &[])
    }
}

// ---- composite shapes (src/composite_shape/*.rs); as a node's primitive a composite is an RT_PRIM_CSG
macro_rules! composite_as_prim {
    ($($t:ty),*) => {$(
        impl DescribePrim for $t {
            fn describe(&self, b: &mut SceneDescBuilder) -> u32 { let c = self.describe_csg(b); b.prim(RT_PRIM_CSG, c, 0, &[]) }
        }
    )*};
}
composite_as_prim!(BaseShape, SubtractShape, AndShape, OrShape, XorShape, MultiAndShape);

impl DescribeCsg for BaseShape {                                      // base_shape.rs:4-16
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let p = self.primitive.describe(b);
        b.csg_node(RT_CSG_BASE, p, &self.get_transform(), &[])
    }
}
impl DescribeCsg for SubtractShape {                                  // subtract_shape.rs:4-8: (positive, negative)
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids = [self.positive.describe_csg(b), self.negative.describe_csg(b)];
        b.csg_node(RT_CSG_SUBTRACT, RT_NONE, &self.get_transform(), &kids)
    }
}
impl DescribeCsg for AndShape {                                       // and_shape.rs:4-8
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids = [self.primary.describe_csg(b), self.secondary.describe_csg(b)];
        b.csg_node(RT_CSG_AND, RT_NONE, &self.get_transform(), &kids)
    }
}
impl DescribeCsg for OrShape {                                        // or_shape.rs:4-8
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids = [self.primary.describe_csg(b), self.secondary.describe_csg(b)];
        b.csg_node(RT_CSG_OR, RT_NONE, &self.get_transform(), &kids)
    }
}
impl DescribeCsg for XorShape {                                       // xor_shape.rs:4-8
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids = [self.primary.describe_csg(b), self.secondary.describe_csg(b)];
        b.csg_node(RT_CSG_XOR, RT_NONE, &self.get_transform(), &kids)
    }
}
impl DescribeCsg for MultiAndShape {                                  // and_shape.rs:68-71
    fn describe_csg(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids: Vec<u32> = self.shapes.iter().map(|s| s.describe_csg(b)).collect();
        b.csg_node(RT_CSG_MULTI_AND, RT_NONE, &self.get_transform(), &kids)
    }
}

// ---- textures and normal maps (src/texture.rs, src/normal_map.rs)
impl DescribeTexture for ImageTexture {                               // texture.rs:30-47
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let img: &RgbImage = &self.image;
        b.image(img as *const RgbImage as usize, img.width(), img.height(), 3, &**img)
    }
}
impl DescribeNormalMap for NormalMap {                                // normal_map.rs:100-119
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let img: &RgbImage = &self.normal_map;
        let image = b.image(img as *const RgbImage as usize, img.width(), img.height(), 3, &**img);
        b.normal_maps.push(RtNormalMap { kind: RT_NMAP_NORMAL, image, depth: 0.0 });
        (b.normal_maps.len() - 1) as u32
    }
}
impl DescribeNormalMap for BumpMap {                                  // normal_map.rs:35-61
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let img: &GrayImage = &self.bump_map;
        let image = b.image(img as *const GrayImage as usize, img.width(), img.height(), 1, &**img);
        b.normal_maps.push(RtNormalMap { kind: RT_NMAP_BUMP, image, depth: self.depth });
        (b.normal_maps.len() - 1) as u32
    }
}

// ---- shaders (src/shader.rs, src/shader/*.rs)
impl DescribeShader for PhongShader {                                 // phong.rs:4-9
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let (d, s, a) = (rgb(self.diffuse), rgb(self.specular), rgb(self.ambient));
        b.shader(RT_SHADER_PHONG, RT_NONE, &[d[0], d[1], d[2], s[0], s[1], s[2], a[0], a[1], a[2], self.shininess], &[])
    }
}
impl DescribeShader for TextureShader {                               // shader/texture.rs:4-6
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { let i = self.texture.describe(b); b.shader(RT_SHADER_TEXTURE, i, &[], &[]) }
}
impl DescribeShader for ReflectionShader {                            // reflection.rs:4-6
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { b.shader(RT_SHADER_REFLECTION, RT_NONE, &rgb(self.reflectivity), &[]) }
}
impl DescribeShader for TranslucentShader {                           // translucent.rs:7-11 (p[4] = 0: the current Fresnel revision)
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let t = rgb(self.translucency);
        b.shader(RT_SHADER_TRANSLUCENT, RT_NONE, &[t[0], t[1], t[2], self.refractive_index, 0.0], &[])
    }
}
impl DescribeShader for MixShader {                                   // shader.rs:52-54
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids: Vec<(f64, u32)> = self.shaders.iter().map(|s| (1.0, s.describe(b))).collect();
        b.shader(RT_SHADER_MIX, RT_NONE, &[], &kids)
    }
}
impl DescribeShader for CompositeShader {                             // shader.rs:77-79: (weight, shader) pairs
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids: Vec<(f64, u32)> = self.shaders.iter().map(|&(w, ref s)| (w, s.describe(b))).collect();
        b.shader(RT_SHADER_COMPOSITE, RT_NONE, &[], &kids)
    }
}
impl DescribeShader for ChainShader {                                 // shader.rs:114-116
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 {
        let kids: Vec<(f64, u32)> = self.shaders.iter().map(|s| (1.0, s.describe(b))).collect();
        b.shader(RT_SHADER_CHAIN, RT_NONE, &[], &kids)
    }
}
impl DescribeShader for NormalMapShader {                             // shader.rs:156-158
    fn describe(&self, b: &mut SceneDescBuilder) -> u32 { let m = self.normal_map.describe(b); b.shader(RT_SHADER_NORMAL_MAP, m, &[], &[]) }
}

// ---- lights (src/light.rs)
impl DescribeLight for PointLight {                                   // light.rs:122-133 (colour already normalised by new())
    fn describe(&self, b: &mut SceneDescBuilder) {
        b.lights.push(RtLight { kind: RT_LIGHT_POINT, _pad: 0, position: v3(self.position), color: rgb(self.color), power: self.power,
                                falloff: [self.falloff.0, self.falloff.1, self.falloff.2], size: 0.0 });
    }
}
impl DescribeLight for SquareLight {                                  // light.rs:152-180
    fn describe(&self, b: &mut SceneDescBuilder) {
        b.lights.push(RtLight { kind: RT_LIGHT_SQUARE, _pad: 0, position: v3(self.position), color: rgb(self.color), power: self.power,
                                falloff: [self.falloff.0, self.falloff.1, self.falloff.2], size: self.size });
    }
}
impl DescribeLight for DirectionLight {                               // light.rs:82-92: `position` carries the (normalised) direction
    fn describe(&self, b: &mut SceneDescBuilder) {
        b.lights.push(RtLight { kind: RT_LIGHT_DIRECTION, _pad: 0, position: v3(self.direction), color: rgb(self.color), power: self.power,
                                falloff: [0.0; 3], size: 0.0 });
    }
}

// ---- the node tree (src/scene.rs:142-148)
impl DescribeNode for SceneNode {
    fn describe(&self, parent: u32, b: &mut SceneDescBuilder) -> u32 {
        let me = b.begin_node(parent, &self.get_transform());
        if let Some(ref p) = self.primitive {
            let prim = p.describe(b);
            b.nodes[me as usize].prim = prim;
        }
        let shader = self.material.describe(b);
        b.nodes[me as usize].shader = shader;
        for c in self.children.iter() { c.describe(me, b); }           // insertion order = DFS pre-order
        b.end_node(me)
    }
}

/// Scene (src/scene.rs:47-53): root, lights, ambient light, sky box.
pub fn describe_scene(scene: &Scene) -> SceneDescBuilder {
    let mut b = SceneDescBuilder::new();
    scene.root.describe(RT_NONE, &mut b);
    for l in scene.lights.iter() { l.describe(&mut b); }
    b.ambient_color = rgb(scene.ambient_light.color);
    b.ambient_power = scene.ambient_light.power;
    if let Some(ref sky) = scene.background {                          // SkyBox, scene.rs:16-21
        let image = sky.image.describe(&mut b);
        b.background = Some((image, mat(&sky.transform.get_transform())));
    }
    b
}

/// The new body of `render_with_config` (src/render.rs:99-312): describe, upload, render.
/// Panics on failure like the reference (`unwrap()`, src/render.rs:173,179,368).
pub fn render_with_config_gpu(scene: Scene, dim: ImageDimension, cam: CameraConfig, cfg: RenderConfig) -> RgbImage {
    assert!(!cfg.interactive, "the stdin probe (src/render.rs:135-144) is not part of the render path");
    let b = describe_scene(&scene);
    let desc = b.finish();
    let device_scene = DeviceScene::new(&desc, 0).unwrap_or_else(|e| panic!("{}", e));
    let d = RtImageDim { width: dim.width, height: dim.height };
    let c = RtCamera { origin: v3(cam.origin), target: v3(cam.target), up: v3(cam.up), fov_y: cam.fov_y };
    let r = RtRenderConfig {
        num_threads: cfg.num_threads as u64, workload_split: cfg.workload_split as u32,
        anti_alias: cfg.anti_alias as u32, aa_threshold: cfg.aa_threshold, aa_rays: cfg.aa_rays as u32,
        recursion_depth: cfg.recursion_depth as u32, interactive: 0, _pad: 0,
        seed: 1, rank: 0, nranks: 1, tile_rows: 0, _pad2: 0,
    };
    let bytes = device_scene.render(d, &c, &r).unwrap_or_else(|e| panic!("{}", e));
    RgbImage::from_raw(dim.width, dim.height, bytes).unwrap()          // the layout make_image produces, src/render.rs:361-370
}
