/*
 * rt_amd.h — C ABI of the MI355X render path.
 *
 * The reference renderer has no FFI of its own: its seam is the public Rust API
 *   pub fn render(scene, image_dimension, camera_config) -> RgbImage              (src/render.rs:89-95)
 *   pub fn render_with_config(scene, image_dimension, camera_config, render_config) (src/render.rs:99-102)
 * whose `Scene` is a tree of trait objects (src/scene.rs:47-53, 142-148).  A Rust
 * host keeps that API and replaces the body of `render_with_config` by
 * "describe the scene into an RtSceneDesc, then call rt_render".  Everything in
 * this header is plain C: pointers, sizes and POD structs; no C++ or torch types.
 *
 * Conventions
 *   - all real numbers are IEEE f64, like the reference (DVec3 / DMat4 / Color);
 *   - matrices are column-major, element (row r, column c) at m[c*4 + r], the
 *     order of the reference's `dmat4!` constructor (src/geometry/matrix.rs:85-90);
 *   - indices are u32; RT_NONE marks "absent";
 *   - the scene description is caller-owned and copied by rt_scene_create.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_NONE 0xFFFFFFFFu
#define RT_ABI_VERSION 1

/* ---- status codes (the reference panics instead; src/render.rs:173,179,368) ---- */
enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = 1,   /* null pointer, zero-sized image, index out of range  */
    RT_ERR_UNSUPPORTED = 2,   /* scene uses a construct the device path cannot run   */
    RT_ERR_DEVICE = 3,        /* HIP runtime error (message in rt_last_error)        */
    RT_ERR_NO_DEVICE = 4,     /* no gfx950 device visible                            */
    RT_ERR_CAPACITY = 5       /* a fixed-capacity buffer would overflow              */
};

typedef struct RtMat4 { double m[16]; } RtMat4;

/* ---- primitives: Intersectable impls (src/primitive.rs, src/primitive/{sphere,plane,cube,polyhedron}.rs, src/mesh.rs) ---- */
enum {
    RT_PRIM_SPHERE = 0,      /* p[0] = radius                         sphere.rs:4-6      */
    RT_PRIM_RECTANGLE = 1,   /* p[0] = width, p[1] = height           plane.rs:64-67     */
    RT_PRIM_PLANE = 2,       /* p[0..3) origin, p[3..6) normal, p[6..9) tangent,
                                p[9..11) surface_scale — the values *after*
                                Plane::with_surface_scale             plane.rs:28-34     */
    RT_PRIM_TRIANGLE = 3,    /* p[0..9) = v1, v2, v3                  plane.rs:111       */
    RT_PRIM_POLYGON = 4,     /* ref = first vertex in points[], count = #vertices (as
                                passed to Polygon::from_vertices)     plane.rs:181       */
    RT_PRIM_CUBE = 5,        /* p[0] = length                         cube.rs:14         */
    RT_PRIM_TETRAHEDRON = 6, /* p[0] = size                           cube.rs:103        */
    RT_PRIM_POLYHEDRON = 7,  /* ref = first of `count` consecutive RT_PRIM_PLANE
                                entries in prims[]                    polyhedron.rs:9    */
    RT_PRIM_ONEWAY = 8,      /* ref = wrapped primitive               primitive.rs:19-46 */
    RT_PRIM_INVERTED = 9,    /* ref = wrapped primitive               primitive.rs:48-81 */
    RT_PRIM_MESH = 10,       /* ref = index into meshes[]             mesh.rs:342-350    */
    RT_PRIM_CSG = 11         /* ref = index into csg[]                composite_shape.rs */
};

typedef struct RtPrim {
    uint32_t kind;
    uint32_t ref;
    uint32_t count;
    uint32_t _pad;
    double p[12];
} RtPrim;

/* ---- CSG: Compositable impls (src/composite_shape/ shape files) ---- */
enum {
    RT_CSG_BASE = 0,       /* prim + transform                      base_shape.rs:4-16    */
    RT_CSG_SUBTRACT = 1,   /* children: positive, negative          subtract_shape.rs:4-8 */
    RT_CSG_AND = 2,        /* children: primary, secondary          and_shape.rs:4-8      */
    RT_CSG_OR = 3,         /*                                       or_shape.rs:4-8       */
    RT_CSG_XOR = 4,        /*                                       xor_shape.rs:4-8      */
    RT_CSG_MULTI_AND = 5   /* n children                            and_shape.rs:68-71    */
};

typedef struct RtCsgNode {
    uint32_t kind;
    uint32_t prim;          /* BASE only */
    uint32_t first_child;   /* offset into csg_children[] */
    uint32_t num_children;
    RtMat4 trans;
} RtCsgNode;

/* ---- shaders: Shadable impls (src/shader.rs, src/shader/{phong,texture,reflection,translucent}.rs) ---- */
enum {
    RT_SHADER_PHONG = 0,       /* p[0..3) diffuse, p[3..6) specular, p[6..9) ambient,
                                  p[9] shininess                     phong.rs:4-9        */
    RT_SHADER_TEXTURE = 1,     /* ref = image                        shader/texture.rs   */
    RT_SHADER_REFLECTION = 2,  /* p[0..3) reflectivity               reflection.rs:4-6   */
    RT_SHADER_TRANSLUCENT = 3, /* p[0..3) translucency, p[3] ior     translucent.rs:7-11
                                  p[4] != 0: the shader's earlier revision, still present as a
                                  comment at translucent.rs:59 — one transmitted ray, no Fresnel
                                  split; the revision the reference's gallery/glass_monkey.png
                                  was rendered with
                                  p[5] != 0 (on the shader of a node): shadow tests pass through the
                                  node's surface, as in the still older revision behind
                                  gallery/subtraction_room.png (opacity BLACK, translucent.rs:64-66;
                                  nothing in the current sources reads get_opacity)        */
    RT_SHADER_MIX = 4,         /* children                           shader.rs:52-54     */
    RT_SHADER_COMPOSITE = 5,   /* weighted children                  shader.rs:77-79     */
    RT_SHADER_CHAIN = 6,       /* children                           shader.rs:114-116   */
    RT_SHADER_NORMAL_MAP = 7   /* ref = normal map                   shader.rs:156-158   */
};

typedef struct RtShader {
    uint32_t kind;
    uint32_t ref;
    uint32_t first_child;   /* offset into shader_child_idx[] / shader_child_weight[] */
    uint32_t num_children;
    double p[10];
} RtShader;

/* ---- lights: Lightable impls (src/light.rs) ---- */
enum {
    RT_LIGHT_POINT = 0,      /* light.rs:122-148 */
    RT_LIGHT_SQUARE = 1,     /* light.rs:151-375 */
    RT_LIGHT_DIRECTION = 2   /* light.rs:82-120; position holds the (normalised) direction */
};

typedef struct RtLight {
    uint32_t kind;
    uint32_t _pad;
    double position[3];
    double color[3];      /* after the constructor's Color::normalize (light.rs:132,176) */
    double power;
    double falloff[3];
    double size;          /* SQUARE: edge length */
} RtLight;

/* ---- images, normal maps, meshes ---- */
typedef struct RtImage {
    uint32_t width, height;
    uint32_t channels;     /* 3 = RGB8 (`to_rgb()`), 1 = luma8 (`to_luma()`) */
    uint32_t _pad;
    uint64_t offset;       /* byte offset into texels[]; rows top-first as decoded —
                              the library applies flip_vertical itself (texture.rs:46) */
} RtImage;

enum { RT_NMAP_NORMAL = 0 /* normal_map.rs:100-129 */, RT_NMAP_BUMP = 1 /* normal_map.rs:35-97 */ };

typedef struct RtNormalMap {
    uint32_t kind;
    uint32_t image;
    double depth;          /* BUMP only */
} RtNormalMap;

typedef struct RtMesh {
    uint32_t num_positions, num_normals, num_texcoords, num_faces;
    uint64_t positions_off;   /* into mesh_f64[]: 3 per vertex */
    uint64_t normals_off;     /* into mesh_f64[]: 3 per vertex */
    uint64_t texcoords_off;   /* into mesh_f64[]: 2 per vertex */
    uint64_t faces_off;       /* into mesh_u32[]: 3 per face   */
} RtMesh;

/* ---- scene tree: SceneNode (src/scene.rs:142-148), DFS pre-order, root = 0 ---- */
typedef struct RtNode {
    uint32_t parent;        /* RT_NONE for the root                          */
    uint32_t prim;          /* RT_NONE when the node has no primitive        */
    uint32_t shader;
    uint32_t subtree_size;  /* this node + all descendants; the children of node i
                               are i+1, i+1+subtree_size(i+1), ...           */
    RtMat4 trans;           /* node → parent                                 */
} RtNode;

typedef struct RtSceneDesc {
    uint32_t abi_version;   /* RT_ABI_VERSION */
    uint32_t num_nodes;            const RtNode* nodes;
    uint32_t num_prims;            const RtPrim* prims;
    uint32_t num_points;           const double* points;            /* xyz triples */
    uint32_t num_csg;              const RtCsgNode* csg;
    uint32_t num_csg_children;     const uint32_t* csg_children;
    uint32_t num_shaders;          const RtShader* shaders;
    uint32_t num_shader_children;  const uint32_t* shader_child_idx;
                                   const double* shader_child_weight; /* COMPOSITE weights */
    uint32_t num_lights;           const RtLight* lights;
    double ambient_color[3];
    double ambient_power;
    uint32_t num_images;           const RtImage* images;
    uint64_t texel_bytes;          const uint8_t* texels;
    uint32_t num_normal_maps;      const RtNormalMap* normal_maps;
    uint32_t num_meshes;           const RtMesh* meshes;
    uint64_t mesh_f64_count;       const double* mesh_f64;
    uint64_t mesh_u32_count;       const uint32_t* mesh_u32;
    uint32_t has_background;       /* Scene::background (scene.rs:51)                */
    uint32_t background_image;
    RtMat4 background_trans;       /* SkyBox::transform (scene.rs:20)                */
} RtSceneDesc;

/* ---- render parameters ---- */
typedef struct RtImageDim { uint32_t width, height; } RtImageDim;            /* render.rs:54-57 */

typedef struct RtCamera {                                                     /* render.rs:70-75 */
    double origin[3], target[3], up[3];
    double fov_y;             /* degrees */
} RtCamera;

typedef struct RtRenderConfig {
    /* the reference's seven fields (src/render.rs:29-37) */
    uint64_t num_threads;     /* CPU pools only; ignored on device */
    uint32_t workload_split;  /* CPU pools only; ignored on device */
    uint32_t anti_alias;
    double   aa_threshold;
    uint32_t aa_rays;
    uint32_t recursion_depth;
    uint32_t interactive;     /* must be 0: the stdin probe is not part of the path */
    /* build-only fields */
    uint32_t _pad;
    uint64_t seed;            /* stream key of the counter-based RNG that replaces thread_rng */
    uint32_t rank;            /* multi-GPU split: this process renders row blocks b   */
    uint32_t nranks;          /*   with b % nranks == rank (nranks = 0 or 1: everything) */
    uint32_t tile_rows;       /* rows per block (0 → 16)                              */
    uint32_t _pad2;
} RtRenderConfig;

/* Ray and work counters.  One "ray" is one closest-hit query that passed the
 * depth/contribution gate (src/scene.rs:67,77) or one any-hit shadow query
 * (src/light.rs:34,266). */
typedef struct RtCounters {
    uint64_t rays_primary;
    uint64_t rays_secondary;
    uint64_t rays_shadow;
    uint64_t rays_aa;            /* primary rays of the anti-alias pass            */
    uint64_t aa_pixels;          /* pixels flagged by the detector                 */
    uint64_t instance_tests;     /* ray → node-local transforms + primitive tests  */
    uint64_t box_tests;          /* BVH child-box slab tests                       */
    uint64_t tri_tests;          /* mesh triangle tests                            */
    uint64_t shade_evals;        /* Phong light evaluations (Lit illums)           */
    uint64_t stack_overflows;    /* rays dropped because a fixed stack was full (must be 0) */
    uint64_t flops;              /* algorithmic flop estimate of the work above and below (f64 + f32) */
    uint64_t schedule;           /* which schedule rendered the frame: 1 ray queues in HBM (one launch
                                    sequence per ray generation), 2 persistent kernel             */
    uint64_t attempts;           /* frames enqueued for this call: 1, or more when a ray generation
                                    outgrew the queues and the frame was redone with larger ones  */
    uint64_t box_tests_f32;      /* of box_tests: those run as packed-f32 slab tests (LDS-resident BVH walk) */
    uint64_t csg_events;         /* leaf hits stepped through by composite-shape evaluations
                                    (src/composite_shape.rs:111-121), 20 flop each in `flops`               */
    uint64_t flops_f32;          /* the part of `flops` that ran in f32: 24 per f32 box test               */
} RtCounters;

/* Optional per-pixel outputs, all W*H (row-major, top row first) or NULL.
 * With a multi-GPU split only the rows this rank owns are written. */
typedef struct RtAux {
    double*   color;      /* W*H*3 linear f64 colour after the AA pass (before clamp/encode) */
    double*   distance;   /* W*H   primary-hit distance, +inf on miss, 0 on AA-resolved pixels
                             (src/render.rs:306)                                         */
    uint32_t* hit_id;     /* W*H   DFS pre-order index of the node whose primitive the
                             primary ray hit; RT_NONE on miss                             */
    uint8_t*  aa_mask;    /* W*H   1 where the AA detector flagged the pixel             */
    RtCounters counters;  /* filled when count_work != 0                                 */
    uint32_t  count_work; /* in: run the instrumented kernels and fill `counters`        */
    uint32_t  _pad;
    double    ms_pass1, ms_aa, ms_encode; /* out: device time of the three stages (HIP events) */
} RtAux;

typedef struct RtScene RtScene;   /* opaque: device-resident flattened scene */

/* Number of gfx950 devices visible to the process (0 if none / no HIP runtime). */
int rt_device_count(void);

/* Thread-local message of the last failing call on this thread. */
const char* rt_last_error(void);

/* Flatten + upload a scene to `device`.  Replaces the 2×500 `scene.clone()`s of
 * src/render.rs:158,237: the device copy is immutable and shared by all frames. */
int rt_scene_create(const RtSceneDesc* desc, int device, RtScene** out);
void rt_scene_destroy(RtScene* scene);

/* Drop-in for render_with_config (src/render.rs:99-312): camera basis, pass 1 over
 * all pixels, AA detect + resolve, clamp/gamma/encode.  `rgb_out` is host memory,
 * W*H*3 bytes, row-major, top row first, RGB interleaved — the RgbImage layout
 * (src/render.rs:361-370).  `aux` may be NULL; its arrays are host pointers. */
int rt_render(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam,
              const RtRenderConfig* cfg, uint8_t* rgb_out, RtAux* aux);

/* Same work, but asynchronous on `hip_stream` (a hipStream_t, NULL = default
 * stream) with every output left in device memory: `d_rgb_out` and the arrays in
 * `d_aux` are device pointers.  With nranks > 1, d_rgb_out receives only this
 * rank's row blocks, compacted: owned block k (k = 0,1,…) starts at byte
 * k*tile_rows*W*3; its capacity must be rt_owned_rows_capacity()*W*3 bytes.
 * This is the entry point the multi-GPU driver and the benchmark use; pair it with
 * rt_frame_status (below) before trusting the frame. */
int rt_render_device(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam,
                     const RtRenderConfig* cfg, uint8_t* d_rgb_out, RtAux* d_aux,
                     void* hip_stream);

/* Completeness of asynchronous frames.  rt_render_device returns before its frame has run, so it
 * cannot say whether a generation of rays outgrew the ray queues (rt_render and instrumented
 * rt_render_device calls wait, notice, and redo the frame themselves).  This call blocks until the
 * most recent rt_render_device call on `scene` has finished and returns RT_OK when no frame since the
 * previous rt_frame_status call lost anything, else RT_ERR_CAPACITY with the number of lost rays /
 * shadow tests in *dropped (may be NULL): such a frame is too dark where they are missing and must be
 * rendered again — the library has enlarged the queues (or moved the scene to its other schedule,
 * which cannot overflow) by then.  The reference has no counterpart: its recursion cannot lose rays. */
int rt_frame_status(const RtScene* scene, uint64_t* dropped);

/* rt_render_device for callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  (No counterpart in the reference.)  It performs no
 * hipMalloc / hipFree: everything a frame needs beyond `d_rgb_out` lives in one static device array
 * of RT_NOALLOC_SCRATCH_BYTES that each GPU holds from the first rt_scene_create on it until the
 * process ends.  Limits: width * height <= RT_NOALLOC_MAX_PIXELS (else RT_ERR_CAPACITY; the AA
 * buffers for aa_rays > 5 can also exceed the array), scenes and recursion depths that run on the
 * queue pipeline (no square / directional lights: RT_ERR_UNSUPPORTED otherwise), no aux outputs.
 * The ray queues get what the array leaves after the image-sized buffers (at the largest frame
 * about 13 entries per pixel; rt_render_device starts with 8 and at least 2^20) and cannot grow:
 * pair the call with rt_frame_status, which returns RT_ERR_CAPACITY for a frame that outgrew them.
 * Larger frames: rt_render_device.
 * The call consists of stream operations only (launches, memsets, one copy into pinned memory; no
 * event, no synchronisation): once an ordinary call has run on the device (kernels loaded, LDS
 * limits granted) it may be captured into a HIP graph and replayed.
 * All calls on one GPU — on whichever scene or stream — share that GPU's array.  Outside capture
 * they take turns on it (each waits for the previous one's last kernel through a per-device event);
 * graphs that contain the call, and a graph against a direct call, are NOT kept apart: their owner
 * must order them, or they race on the array.
 * Cost: the array is part of the library's code object, so every process that loads the library
 * holds RT_NOALLOC_SCRATCH_BYTES of device memory on each GPU it touches, used or not. */
#define RT_NOALLOC_SCRATCH_BYTES (256u << 20)
#define RT_NOALLOC_MAX_PIXELS 32768u
int rt_render_device_noalloc(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam,
                             const RtRenderConfig* cfg, uint8_t* d_rgb_out, void* hip_stream);

/* ceil(ceil(H/tile_rows)/nranks)*tile_rows: rows of the compact per-rank buffer. */
uint32_t rt_owned_rows_capacity(uint32_t height, uint32_t nranks, uint32_t tile_rows);

/* Device time (ms, HIP events on the render stream) of the most recent
 * rt_render_device call on `scene`, per stage; blocks until that call finished. */
int rt_last_timings(const RtScene* scene, double* ms_pass1, double* ms_aa, double* ms_encode);

/* ---- scene ingest: the step before the path ---- */

/* rt_scene_create with a say in where the scene is prepared.  The reference prepares a mesh on the
 * host when it is loaded (the bounding tree of src/mesh.rs:124-144,281-332) and flips every image
 * after decoding it (src/texture.rs:46, src/normal_map.rs:52,116).  RT_INGEST_HOST does both on the
 * host, RT_INGEST_DEVICE on `device` (one launch per level of a hierarchy, one workgroup per node;
 * the rows of an image change places where they landed), RT_INGEST_AUTO (what rt_scene_create
 * uses) on the device from RT_INGEST_AUTO_FACES mesh faces / 1 MiB of texels on.  Both give the
 * same device scene, bit for bit. */
enum { RT_INGEST_AUTO = 0, RT_INGEST_HOST = 1, RT_INGEST_DEVICE = 2 };
#define RT_INGEST_AUTO_FACES 65536u
int rt_scene_create_with(const RtSceneDesc* desc, int device, uint32_t ingest, RtScene** out);

/* Mesh::from_path (src/mesh.rs:383-414) without the file: Wavefront OBJ text -> the unified-vertex
 * triangle mesh that tobj 0.1.7's load_obj hands the reference (every distinct v/vt/vn corner of a
 * model — an `o` / `g` block with faces — one vertex, in order of first appearance; polygons as
 * triangle fans; positions, normals and texture coordinates parsed as f32 and widened), parsed on
 * `device`.  The arrays are host memory owned by the library until rt_obj_release; they are what
 * RtMesh / mesh_f64 / mesh_u32 of an RtSceneDesc take.  Texture coordinates and normals exist only
 * for vertices whose corner names one, like the reference's arrays.
 * tobj's source is not part of the reference tree (a Cargo dependency), so the match is pinned on the
 * reference's own .obj assets and on a sequential reader of the same rules, not on tobj itself.
 * Where tobj is known to read a text differently, the text is refused rather than guessed at:
 *   - `l` and `p` elements (tobj sends them through its face path, where they add vertices):
 *     RT_ERR_UNSUPPORTED;
 *   - a v / vn line with fewer than three numbers, a vt line with fewer than two, a token that is
 *     no decimal number ("1.0f", hex floats; tobj fails the whole load): RT_ERR_UNSUPPORTED;
 *   - a face that names a vertex, texture coordinate or normal the text does not hold:
 *     RT_ERR_INVALID_ARG.
 * Materials are ignored: only the text is read, never a .mtl file, and without a resolved material
 * library tobj opens no new model at a `usemtl` line either, so models are bounded by o / g lines
 * alone.  Whether a model has texture coordinates / normals is decided from the counts of the whole
 * text, not from the counts at the point where tobj would export the model; the two differ only for
 * texts that define their first vt / vn after a model that uses none. */
typedef struct RtObjMesh {
    uint32_t num_positions, num_normals, num_texcoords, num_faces;
    double* positions;     /* 3 per vertex */
    double* normals;       /* 3 per normal */
    double* texcoords;     /* 2 per texture coordinate */
    uint32_t* faces;       /* 3 per face */
} RtObjMesh;
int rt_obj_parse(int device, const char* text, uint64_t bytes, RtObjMesh* out);
void rt_obj_release(RtObjMesh* mesh);

/* ---- output stage: the step after the path ---- */

/* write_to_png (src/lib.rs:41-46): the frame as "<file_name>.png" (the reference appends the
 * extension the same way and creates no directories).  `rgb` is host memory in the layout rt_render
 * leaves (W*H*3, top row first).  The scanline filters are chosen and applied on `device` (one
 * workgroup per row), the host deflates and writes the file.  Any PNG decoder returns exactly the
 * bytes of `rgb`.  The reference prints "save successful" / "FAILED TO SAVE" and returns nothing;
 * this returns a status instead and prints nothing. */
int rt_write_png(int device, const uint8_t* rgb, uint32_t width, uint32_t height, const char* file_name);

/* The same for a frame still in device memory (what rt_render_device leaves in d_rgb_out with
 * nranks <= 1); blocks until the file is written. */
int rt_write_png_device(int device, const uint8_t* d_rgb, uint32_t width, uint32_t height,
                        const char* file_name, void* hip_stream);

/* ProgressTracker (src/progress_tracker.rs:37-42,48-77).  A frame runs on the device without
 * coming back to the host, so rt_render reports its stages once the frame is complete, in the
 * tracker's order and with the device's clock: RT_PROGRESS_START (done 0, total = pixels:
 * "START RENDERING"), RT_PROGRESS_PASS1 (done = total = pixels traced by pass 1, elapsed = its
 * device time), RT_PROGRESS_START_AA (done 0, total = pixels the detector flagged: "START
 * ANTIALIASING: n / total"), RT_PROGRESS_AA (done = total = flagged pixels), RT_PROGRESS_END
 * (elapsed = device time of the whole frame: "Elapsed Time").  `fn` = NULL switches reporting off.
 * Called on the thread that called rt_render, before it returns. */
enum { RT_PROGRESS_START = 0, RT_PROGRESS_PASS1 = 1, RT_PROGRESS_START_AA = 2, RT_PROGRESS_AA = 3, RT_PROGRESS_END = 4 };
typedef void (*RtProgressFn)(void* user, uint32_t stage, uint64_t done, uint64_t total, double elapsed_seconds);
int rt_set_progress(RtScene* scene, RtProgressFn fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
