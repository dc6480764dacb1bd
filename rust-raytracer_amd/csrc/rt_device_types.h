// rt_device_types.h — flat, device-resident scene layout shared by the host-side
// flattening (rt_flatten.hpp) and the kernels (rt_kernels.hip).
//
// The reference walks a tree of boxed trait objects per ray (src/scene.rs:196-225).
// Here the tree is flattened once per scene into arrays of PODs that live in HBM for
// the lifetime of the RtScene and are read-only during rendering:
//   instances[]  one per SceneNode that owns a primitive, with the composed
//                world<->local transforms of all its ancestors
//   prims[]      analytic primitives (+ wrapper chain), tris[], faced[], planes
//   meshes[]     per mesh: BVH nodes + triangle records + vertex attributes
//   csg_*        composite shapes as a postfix program over transformed leaves
//   shaders[]    the shader tree, walked by a small interpreter per hit
//   lights[], images[] + texel blob, normal maps, gamma table
#pragma once
#include <stdint.h>

#define RT_MAX_STACK 128       // secondary-ray stack entries per camera sample
#define RT_BATCH 8             // most scene queries (shadow tests + rays off the stack) a lane runs per trip
#define RT_BVH_STACK 48        // BVH traversal stack entries (deeper trees are rejected at scene creation)
#define RT_LDS_STACK 16        // ... of which this many live in LDS, the rest in the lane's global memory
#define RT_HOT_LIMIT (24u * 1024u)   // largest set of small scene tables that is copied into LDS
#define RT_MAX_SHADOW_JOBS 32  // deferred shadow tests a lane may hold between two trips
#define RT_CSG_MAX_HITS 16     // hits per composite-shape list
#define RT_CSG_ARENA 96        // hits across all live lists of one composite query
#define RT_CSG_MAX_LISTS 12
#define RT_SHADE_MAX_FRAMES 8  // nesting depth of group shaders
#define RT_SHADE_MAX_SPAWN 16  // secondary + deferred shadow rays one hit may spawn

// Scene features: the trace kernels are compiled for a handful of feature sets so that a
// scene only pays (registers, scratch, code) for what it contains.
#define RTF_MESH 1u            // triangle meshes
#define RTF_CSG 2u             // composite shapes
#define RTF_FACED 4u           // Cube / Tetrahedron
#define RTF_MISC 8u            // Plane, Triangle, Polygon, Polyhedron
#define RTF_WRAP 16u           // OneWay / Inverted
#define RTF_INLINE_LIGHTS 32u  // lights whose shadow tests cannot be deferred (square, directional, Phong under a ray-spawning Mix)
#define RTF_IMAGES 64u         // textures, normal / bump maps, skybox
#define RTF_ALL 127u           // (feature bits a scene can have apart from the two below)
#define RTF_TLAS 256u           // a BVH over the instances (scenes with many of them)
#define RT_TLAS_STACK 32
#define RT_TLAS_MIN_INSTANCES 12
#define RTF_LDS_BVH 128u        // kernel variant: mesh BVHs are walked out of LDS (512-thread workgroups, one per CU)

struct DAffine {               // y = R x + t, rows of R in r[0..3), r[4..7), r[8..11) with t at r[3], r[7], r[11]
    double r[12];
};

struct DXform {
    DAffine w2l;               // parent/world → local
    DAffine l2w;               // local → parent/world
    double nmat[9];            // normals: n_world ∝ nmat * n_local  (= transpose of the linear part of w2l)
    double uniform_inv_s;      // 1/|w2l_lin d| for unit d when the scale is uniform, else 0
    double uniform_s;          // |w2l_lin d| for unit d when the scale is uniform, else 0
};

enum DPrimKind {
    DP_SPHERE = 0, DP_RECTANGLE, DP_PLANE, DP_TRIANGLE, DP_POLYGON, DP_FACED, DP_POLYHEDRON, DP_MESH, DP_CSG
};
enum { DWRAP_NONE = 0, DWRAP_ONEWAY = 1, DWRAP_INVERTED = 2 };

struct DPrim {
    uint32_t kind;
    uint32_t ref;              // TRIANGLE: tris[]; POLYGON: polys[]; FACED: faced[]; POLYHEDRON: first plane prim; MESH: meshes[]; CSG: csg_roots[]
    uint32_t count;            // POLYHEDRON: number of planes
    uint32_t wrap;             // wrapper chain, 2 bits per level, innermost first
    double p[12];              // SPHERE r; RECTANGLE w,h; PLANE origin,normal,tangent,scale
};

struct DTri {                  // primitive::Triangle with its constructor-derived fields (plane.rs:111-126)
    double v[9], e[9], n[3], tangent[3];
    double sc_width, sc_height, side_c;
};

struct DPoly {                 // primitive::Polygon: projected vertices and edges live in points[] (plane.rs:181-202)
    uint32_t first_vertex, first_edge, count, _pad;
    double n[3];
};

struct DFaced {                // Cube (6 rectangles) / Tetrahedron (4 triangles): cube.rs
    uint32_t nfaces, base_is_triangle, base_tri, is_cube;
    double base_w, base_h;     // rectangle base
    DXform face[6];
};

struct DBvhNode {              // binary BVH; both children's boxes are stored in the parent
    double lo[2][3], hi[2][3];
    uint32_t child[2];         // bit 31: leaf, bits 28..30: triangle count, low 28 bits: first triangle; else node index
    uint32_t count[2];         // leaf: triangle count (host-side bookkeeping)
};

struct DMeshTri {              // one mesh face in BVH order (mesh.rs:416-466 inputs)
    double v1[3], v2[3], v3[3], n[3];
    double inv_area;
    uint32_t i1, i2, i3, face;
};

// What shading needs of a mesh face beyond its DMeshTri, in BVH order like it: the normals and texture
// coordinates of its three corners, gathered once on the host so that a hit fetches them with one read
// alongside the triangle instead of following three vertex indices (mesh.rs:449-461).
struct DMeshAttr {
    double n1[3], n2[3], n3[3];
    double uv[6];              // (u, v) of the corners
    uint32_t flags;            // bit 0: the mesh has a normal for the face's first vertex index (mesh.rs:449), bit 1: a texture coordinate (:455)
    uint32_t _pad;
};

// Copies of the BVH for the LDS-resident walk: boxes rounded outward to f32 (they only cull;
// every arithmetic step still runs in f64), triangles without their attribute indices.
struct LBvhNode {              // 64 bytes
    float lo[2][3], hi[2][3];
    uint32_t child[2];         // bit 31: leaf, bits 28..30: triangle count, low bits: first triangle; else node index
    uint32_t _pad[2];
};
struct LMeshTri {              // 64 bytes: vertices as f32 (exact: only used when every coordinate is an f32 value,
    float v1[3], v2[3], v3[3]; //           as it always is for OBJ input, mesh.rs:352-359), the unit normal as f64
    float _pad;
    double n[3];
};

struct DMesh {
    uint32_t node_off, tri_off, num_tris, num_nodes;
    uint32_t lds_node_off, lds_tri_off;   // byte offsets into the scene's LDS blob
    uint32_t normal_off, uv_off;   // (unused: the corners' attributes are gathered per face, DMeshAttr)
    uint32_t num_normals, num_uvs;
    double lo[3], hi[3];
    double max_abs;            // largest |coordinate| of any node box (bounds the rounding error of the f32 box test)
};

enum DCsgKind { DC_BASE = 0, DC_SUBTRACT, DC_AND, DC_OR, DC_XOR, DC_MULTI_AND };
struct DCsgOp {                // postfix program: children precede their parent
    uint32_t kind, num_children, leaf, _pad;
};
struct DCsgLeaf {
    DXform x;                  // composite-root parent space ↔ leaf primitive space
    uint32_t prim, _pad;
};
struct DCsgRoot {
    uint32_t op_begin, op_count;
    uint32_t leaf_begin, num_leaves;   // its leaves are consecutive in csg_leaves[], in the order of their BASE ops
    // box around all leaves in the composite's own space (bounded == 0: some leaf is infinite): a ray that
    // misses it hits no leaf, so every hit list is empty and the composite has no hit either
    uint32_t bounded, _pad;
    double lo[3], hi[3];
};

enum DShaderKind { DS_PHONG = 0, DS_TEXTURE, DS_REFLECTION, DS_TRANSLUCENT, DS_MIX, DS_COMPOSITE, DS_CHAIN, DS_NORMAL_MAP };
struct DShader {
    uint32_t kind, ref, first_child, num_children;
    uint32_t nm_begin, nm_count;   // CHAIN: normal maps applied (in order) by modify_intersect
    double p[10];
};

struct DNormalMap { uint32_t kind, image; double depth; };
struct DImage { uint32_t width, height, channels, _pad; uint64_t offset; };

struct DLight {
    uint32_t kind, index;
    double position[3], color[3], power, falloff[3], size, inv_area;
};

struct DInstance {
    DXform x;
    uint32_t prim, shader, node_id;
    uint32_t is_root;          // bit 0: the scene's root node (its shadow test is inclusive, scene.rs:236); bit 1: shadow tests pass through (RT_INST_CLEAR_TO_LIGHT)
};
#define RT_INST_ROOT 1u
#define RT_INST_CLEAR_TO_LIGHT 2u

// The reference's own chain of transforms, node by node, for the rare query that has to be decided the way
// SceneNode::trace decides it (two hits at distances that differ by rounding only): per scene node its matrix
// and the inverse as DMat4::inverse computes it, per instance the nodes from the root down to its own.
struct DNodeXf {
    DAffine trans, inv;        // node -> parent, parent -> node
};
struct DChain {
    uint32_t first, len;       // chain_nodes[first .. first + len): node indices, root first
};

// Per-lane working memory of the trace kernel (global memory, one contiguous block per lane so
// that a lane's 16-byte accesses stay within its own cache lines whatever its stack depth).
struct alignas(16) LaneRay {   // one pending camera / secondary ray (Ray of src/geometry/ray.rs:4-9 plus its pixel weight)
    double o[3], d[3];
    double w[3];               // product of shader weights from the camera to this ray
    double c[3];               // Ray::contribution
    uint32_t depth, path;
    uint32_t slot;             // the sample it belongs to: pixel index (pass 1) or AA sample index; also its accumulator
    uint32_t flags;            // bit 0: pass-1 camera ray (its hit is the pixel's distance / hit id); bit 1: camera ray of either pass
};
struct alignas(16) ShadowJob { // a deferred point-light visibility test
    double o[3], d[3];
    double dist;               // distance to the light
    double w[3];               // colour added to the sample if nothing blocks the light
    uint32_t slot;
    uint32_t _pad[3];
};
struct alignas(16) TraceJob {  // one scene query of the current trip, between its phases
    double t;                  // closest-hit: world distance of the best hit so far (+inf: none); any-hit: the distance limit
    uint32_t inst;
    uint32_t found;            // closest-hit: a hit exists; any-hit: an occluder was found
    double ct;                 // candidate in the instance's primitive space (Cand)
    uint32_t ca, cb;
    uint32_t src;              // byte offset of the ray's o[3], d[3] in the lane's memory
    uint32_t flags;            // bit 0: any-hit query; bit 1: nothing left to do (resolved, or never started)
    uint32_t _pad[2];
};
#define RT_LANE_OFF_STACK 0u
#define RT_LANE_OFF_JOBS (RT_LANE_OFF_STACK + RT_MAX_STACK * 112u)
#define RT_LANE_OFF_SHADOW (RT_LANE_OFF_JOBS + (RT_BATCH + RT_MAX_SHADOW_JOBS) * 48u)
#define RT_LANE_OFF_BVH (RT_LANE_OFF_SHADOW + RT_MAX_SHADOW_JOBS * 96u)
#define RT_LANE_BYTES ((RT_LANE_OFF_BVH + (RT_BVH_STACK - RT_LDS_STACK) * 4u + 255u) & ~255u)

struct DScene {
    const DInstance* instances;  uint32_t num_instances;
    const DNodeXf* node_xf;          // per scene node (global memory: read by tie decisions only)
    const DChain* inst_chain;        // per instance
    const uint32_t* chain_nodes;
    const uint32_t* mesh_instances; uint32_t num_mesh_instances;   // instances whose primitive is an un-wrapped mesh, ascending
    const DBvhNode* tlas_nodes;      // BVH over the world-space boxes of the bounded instances (null: test every instance in turn)
    const uint32_t* tlas_index;      // leaf contents: instance indices
    const uint32_t* tlas_unbounded;  // instances without a finite box (planes, polyhedra, composites): always tested
    uint32_t num_unbounded;
    const double* inst_box;          // scenes without a TLAS: per instance its world-space box, lo[3] hi[3]; null when no instance is worth a box test
    uint32_t inst_box_mask;          // ... bit i: test instance i's box before the instance itself
    const DPrim* prims;
    const DTri* tris;
    const DPoly* polys;
    const double* points;
    const DFaced* faced;
    const DMesh* meshes;
    const DBvhNode* bvh_nodes;
    const DMeshTri* mesh_tris;
    const DMeshAttr* mesh_attrs;     // one per mesh_tris entry
    const DCsgRoot* csg_roots;
    const DCsgOp* csg_ops;
    const DCsgLeaf* csg_leaves;
    const DShader* shaders;
    const uint32_t* shader_bounds;   // per shader: most rays (low 16 bits) and most deferred shadow tests (high bits) a hit on it can ask for
    const uint32_t* shader_child_idx;
    const double* shader_child_weight;
    const uint32_t* chain_nmaps;
    const DNormalMap* normal_maps;
    const DImage* images;
    const uint8_t* texels;       // rows already flipped vertically (texture.rs:46)
    const double* gamma_lut;     // 256 entries: (i/255)^2.2
    const DLight* lights;        uint32_t num_lights;
    double ambient[3];           // ambient colour * power
    uint32_t has_background, background_image;
    DAffine background_w2l;
    const uint8_t* lds_blob;     // LBvhNode / LMeshTri arrays of every mesh, staged into LDS by each workgroup
    uint32_t lds_blob_bytes;     // 0: the meshes do not fit; walk the global-memory BVH instead
    // The small tables (instances, primitives, shaders, lights, ... everything but BVHs, mesh attributes
    // and texels) lie back to back in one allocation; when that is small enough the kernels copy it
    // into LDS and follow their pointers there (see localise_scene).
    const uint8_t* hot_blob;
    uint32_t hot_bytes;          // 0: too large for LDS, the tables are read from global memory
    uint32_t features;           // RTF_* bits this scene needs
    uint32_t max_ray_spawn;      // most secondary rays one hit can push
    uint32_t max_shadow_spawn;   // most deferred shadow tests one hit can push
    uint32_t inline_point_lights; // 1: point-light shadow tests run inside Phong (needs RTF_INLINE_LIGHTS)
    uint32_t has_clear_to_light;  // 1: some instance carries RT_INST_CLEAR_TO_LIGHT
};

struct DCamera {
    double origin[3], side[3], up[3], view[3];
    double x_factor, fov_factor;
    uint32_t width, height;
};

struct DRenderParams {
    DCamera cam;
    uint32_t recursion_depth, aa_rays;
    double aa_threshold;
    uint64_t seed;
    uint32_t rank, nranks, tile_rows, _pad;
};

struct DCounters {             // mirrors RtCounters' first ten fields
    unsigned long long rays_primary, rays_secondary, rays_shadow, rays_aa, aa_pixels,
        instance_tests, box_tests, tri_tests, shade_evals, stack_overflows;
    unsigned long long box_tests_f32;   // of box_tests: those that ran as packed-f32 slab tests (the LDS-resident BVH walk)
    unsigned long long csg_events;      // leaf hits a composite-shape evaluation stepped through (composite_shape.rs:111-121)
    // lane occupancy of the trace kernel's phases (instrumented launches only): per phase the
    // number of loop trips a wave made and the number of lanes that had work in them
    unsigned long long prof[16];
};
// (The queue pipeline has no trips or phases A/B: its shade kernel files its own sections under those
// slots - PF_TRIPS/PF_TRIP_LANES trips and hits, PF_A_ITERS.. PF_CYC_T cycles per section, see k_wf_shade -
// and its walk kernel uses the PF_T_* slots as the persistent kernel does.)
enum { PF_TRIPS = 0, PF_TRIP_LANES, PF_A_ITERS, PF_A_LANES, PF_T_ITERS, PF_T_LANES, PF_T_REFILLS, PF_T_REFILL_LANES,
       PF_T_LEAF_ITERS, PF_T_LEAF_LANES, PF_B_ITERS, PF_B_LANES, PF_CYC_A, PF_CYC_T, PF_CYC_B, PF_CYC_FETCH };
