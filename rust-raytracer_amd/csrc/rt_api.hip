// rt_api.hip — C ABI of the MI355X render path (include/rt_amd.h): scene upload,
// frame driver, error reporting.  The frame driver replaces render_with_config
// (src/render.rs:99-312); the scene upload replaces the per-job `scene.clone()`
// (src/render.rs:158,237) by one immutable device copy.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_flatten.hpp"
#define RT_FRAME_KERNELS   // this file holds the small per-frame kernels (generation 0, resolve, AA, encode)
#include "rt_variants.h"

// Workspace of rt_render_device_noalloc: every buffer a frame needs beyond its inputs and outputs
// (image-sized intermediates, accumulators, ray queues, walk stacks) is carved out of this array.
// Each GPU the library runs on holds its own copy at its own address until the process ends.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Caller-provided device memory that buffers are carved from instead of hipMalloc (the entry point that
// allocates nothing): a bump pointer, reset for every frame.
struct Arena {
    uint8_t* base = nullptr;
    size_t capacity = 0, used = 0;
};

struct DeviceBuf {
    void* p = nullptr;
    size_t bytes = 0;
    Arena* arena = nullptr;            // set: memory comes out of the arena and is never freed here
    int ensure(size_t n) {
        if (n <= bytes) return RT_OK;
        if (arena) {
            const size_t at = (arena->used + 255u) & ~(size_t)255u;
            if (at + n > arena->capacity) return fail(RT_ERR_CAPACITY, "frame workspace exceeds the static scratch array");
            p = arena->base + at; bytes = n; arena->used = at + n;
            return RT_OK;
        }
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return RT_OK;
    }
    void release() { if (p && !arena) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Device address of `scratch` on each GPU (in host code the array's name is only a handle): looked up
// once per device, with that device current, when the first scene is created on it.
std::mutex g_scratch_mu;
void* g_scratch_ptr[64] = {};
// ... and one event per device, recorded behind every frame that used the array: the next such frame waits for it,
// so that frames of different scenes or streams take turns on the array instead of racing.
hipEvent_t g_scratch_free[64] = {};
int scratch_lookup(int device) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (device < 0 || device >= 64) return fail(RT_ERR_INVALID_ARG, "device index out of range");
    if (g_scratch_ptr[device]) return RT_OK;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RT_ERR_DEVICE, "hipGetSymbolAddress(scratch) returned no address");
    HIP_TRY(hipEventCreateWithFlags(&g_scratch_free[device], hipEventDisableTiming));
    g_scratch_ptr[device] = ptr;
    return RT_OK;
}

template <class T>
int upload(const std::vector<T>& v, std::vector<void*>& owned, const T** out) {
    *out = nullptr;
    if (v.empty()) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
    owned.push_back(p);
    HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

}  // namespace

// The other translation units of the library report their errors through the same thread-local message.
int rt_set_error(int code, const std::string& msg) { return fail(code, msg); }

// rt_ingest.hip: the meshes' hierarchies built on `device`, and images flipped where they land
rtflat::MeshBuilder rt_device_mesh_builder(int device);
hipError_t rt_device_flip_rows(uint8_t* texels, uint32_t row_bytes, uint32_t height, hipStream_t stream);

// Device memory one frame works in.  A scene owns one (grown on demand with hipMalloc, reused from
// frame to frame); the entry point that allocates nothing builds a temporary one over the static scratch array.
struct Workspace {
    DeviceBuf color, distance, hit_id, mask, list, samples, rows, scalars, rgb_full;
    DeviceBuf queue[2], acc, acc_flags, lane_mem;
    // queue-pipeline path: rays / shadow tests of two generations, their query records, walk list
    DeviceBuf wf_rays[2], wf_shadows[2], wf_rjobs, wf_sjobs, wf_walk_list, wf_csg_list, wf_overflow, wf_scalars, wf_ties;
    uint32_t wf_cap = 0;
    bool acc_dirty = false;                        // a frame was abandoned between its rounds and its resolve
    std::vector<DeviceBuf*> all() {
        return {&color, &distance, &hit_id, &mask, &list, &samples, &rows, &scalars, &rgb_full, &queue[0], &queue[1], &acc, &acc_flags, &lane_mem,
                &wf_rays[0], &wf_rays[1], &wf_shadows[0], &wf_shadows[1], &wf_rjobs, &wf_sjobs, &wf_walk_list, &wf_csg_list, &wf_overflow, &wf_scalars, &wf_ties};
    }
    void use(Arena* a) { for (DeviceBuf* b : all()) b->arena = a; }
    void release() { for (DeviceBuf* b : all()) b->release(); }
};

struct RtScene {
    int device = 0;
    int num_cus = 256;
    DScene ds{};
    std::vector<void*> owned;
    uint32_t lds_limit = 65536;
    // per-frame workspace, grown on demand and reused across frames
    std::mutex mu;
    Workspace ws;
    // The device keeps one running count of lost entries (never reset, wf_sticky); every queue-pipeline frame
    // ends with a copy of it to pinned host memory.  `dropped_seen` is the value the host has acted on (by
    // enlarging the queues or moving the scene to the persistent kernel), `dropped_reported` the value the
    // caller has been told about through rt_frame_status.
    DeviceBuf wf_sticky;
    unsigned int* h_dropped = nullptr;
    unsigned int dropped_seen = 0, dropped_reported = 0;
    // Shade kernel of the queue pipeline per hit list (materials without / with rays): 0 the general kernel, 1 / 2 the
    // kernel built for a class of simple materials (k_wf_shade in rt_wavefront.h), when every material on the list is of it
    int shade_cls[2] = {0, 0};
    bool shade_none[2] = {false, false};   // no material of the scene belongs on that hit list: its launch is left out
    bool ties_possible = false;            // two or more instances of the kinds whose ties k_wf_ties decides
    uint32_t wf_scale = 8;                 // queue entries per pixel of the frame; doubled when a generation did not fit
    bool last_used_queues = false, queue_overflowed = false;
    int last_attempts = 1;                 // frames the most recent synchronous call enqueued
    bool last_noalloc = false;             // the most recent frame ran in the static scratch array: its queues cannot grow
    hipEvent_t ev_done = nullptr;          // after the last command of the most recent frame
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool timed = false;
    // queue pipeline as several chains side by side: chain 0 runs on the caller's stream, the others on these
    hipStream_t chain_stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t chain_fork = nullptr, chain_join[3] = {nullptr, nullptr, nullptr};
    RtProgressFn progress = nullptr;       // rt_set_progress
    void* progress_user = nullptr;
};

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

const char* rt_last_error(void) { return g_error.c_str(); }

void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->owned) (void)hipFree(p);
    s->ws.release();
    s->wf_sticky.release();
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    if (s->ev_done) (void)hipEventDestroy(s->ev_done);
    for (auto& e : s->chain_join) if (e) (void)hipEventDestroy(e);
    if (s->chain_fork) (void)hipEventDestroy(s->chain_fork);
    for (auto& st : s->chain_stream) if (st) (void)hipStreamDestroy(st);
    if (s->h_dropped) (void)hipHostFree(s->h_dropped);
    delete s;
}

int rt_scene_create(const RtSceneDesc* desc, int device, RtScene** out) { return rt_scene_create_with(desc, device, RT_INGEST_AUTO, out); }

int rt_scene_create_with(const RtSceneDesc* desc, int device, uint32_t ingest, RtScene** out) {
    if (!desc || !out) return fail(RT_ERR_INVALID_ARG, "rt_scene_create: null argument");
    *out = nullptr;
    if (ingest > RT_INGEST_DEVICE) return fail(RT_ERR_INVALID_ARG, "rt_scene_create_with: unknown ingest mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(RT_ERR_INVALID_ARG, "device index out of range");
    rtflat::FlatScene fs;
    {   // where the meshes' hierarchies are built and the images flipped: both give the same scene, bit for bit
        uint64_t faces = 0;
        for (uint32_t i = 0; desc->meshes && i < desc->num_meshes; ++i) faces += desc->meshes[i].num_faces;
        const bool on_device = ingest == RT_INGEST_DEVICE || (ingest == RT_INGEST_AUTO && faces >= RT_INGEST_AUTO_FACES);
        if (on_device) fs.mesh_builder = rt_device_mesh_builder(device);
        fs.flip_on_device = ingest == RT_INGEST_DEVICE || (ingest == RT_INGEST_AUTO && desc->texel_bytes >= (1u << 20));
    }
    const auto t_begin = std::chrono::steady_clock::now();
    try {
        rtflat::flatten(*desc, fs);
    } catch (const rtflat::Unsupported& e) {
        return fail(RT_ERR_UNSUPPORTED, e.what());
    } catch (const rtflat::DeviceFailure& e) {
        return fail(RT_ERR_DEVICE, e.what());
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_DEVICE, "rt_scene_create: out of host memory");
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID_ARG, e.what());
    }
    const auto t_flat = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(device));
    { int sr = scratch_lookup(device); if (sr != RT_OK) return sr; }
    RtScene* s = new (std::nothrow) RtScene();
    if (!s) return fail(RT_ERR_DEVICE, "out of host memory");
    s->device = device;
    if (const char* e = std::getenv("RT_WF_SCALE")) { long v = std::atol(e); if (v >= 1 && v <= 32) s->wf_scale = (uint32_t)v; }   // tests: start with small queues
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount;
    DScene& d = s->ds;
    // the small tables back to back in one allocation (16-byte aligned sections), the big ones on their own
    std::vector<uint8_t> hot;
    std::vector<std::pair<const void**, size_t>> hot_fix;   // (pointer field, offset into `hot`)
    auto pack = [&](const void* data, size_t bytes, const void** field) {
        *field = nullptr;
        if (bytes == 0) return;
        while (hot.size() % 16) hot.push_back(0);
        hot_fix.emplace_back(field, hot.size());
        const uint8_t* p = static_cast<const uint8_t*>(data);
        hot.insert(hot.end(), p, p + bytes);
    };
#define HOT(field, vec) pack(vec.data(), vec.size() * sizeof(vec[0]), reinterpret_cast<const void**>(&d.field))
    HOT(instances, fs.instances); HOT(inst_box, fs.inst_box); HOT(mesh_instances, fs.mesh_instances); HOT(prims, fs.prims); HOT(tris, fs.tris); HOT(polys, fs.polys);
    HOT(points, fs.points); HOT(faced, fs.faced); HOT(meshes, fs.meshes); HOT(csg_roots, fs.csg_roots); HOT(csg_ops, fs.csg_ops);
    HOT(csg_leaves, fs.csg_leaves); HOT(shaders, fs.shaders); HOT(shader_bounds, fs.shader_bounds); HOT(shader_child_idx, fs.shader_child_idx);
    HOT(shader_child_weight, fs.shader_child_weight); HOT(chain_nmaps, fs.chain_nmaps); HOT(normal_maps, fs.normal_maps);
    HOT(images, fs.images); HOT(gamma_lut, fs.gamma_lut); HOT(lights, fs.lights);
#undef HOT
    while (hot.size() % 16) hot.push_back(0);
    int rc = upload(hot, s->owned, &d.hot_blob);
    for (auto& f : hot_fix) *f.first = d.hot_blob + f.second;
    d.hot_bytes = hot.size() <= RT_HOT_LIMIT ? (uint32_t)hot.size() : 0u;
#define UP(field, vec) if (rc == RT_OK) rc = upload(vec, s->owned, &d.field)
    UP(tlas_nodes, fs.tlas_nodes); UP(tlas_index, fs.tlas_index); UP(tlas_unbounded, fs.tlas_unbounded);
    UP(bvh_nodes, fs.bvh_nodes); UP(mesh_tris, fs.mesh_tris); UP(mesh_attrs, fs.mesh_attrs); UP(texels, fs.texels);
    UP(node_xf, fs.node_xf); UP(inst_chain, fs.inst_chain); UP(chain_nodes, fs.chain_nodes);
#undef UP
    if (rc != RT_OK) { rt_scene_destroy(s); return rc; }
    if (fs.flip_on_device)
        for (const DImage& im : fs.images) {
            hipError_t fe = rt_device_flip_rows(const_cast<uint8_t*>(d.texels) + im.offset, im.width * im.channels, im.height, nullptr);
            if (fe != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, std::string("image flip: ") + hipGetErrorString(fe)); }
        }
    d.num_instances = (uint32_t)fs.instances.size();
    d.num_mesh_instances = (uint32_t)fs.mesh_instances.size();
    d.num_unbounded = (uint32_t)fs.tlas_unbounded.size();
    d.inst_box_mask = fs.inst_box_mask;
    d.num_lights = (uint32_t)fs.lights.size();
    std::memcpy(d.ambient, fs.ambient, sizeof d.ambient);
    d.has_background = fs.has_background; d.background_image = fs.background_image; d.background_w2l = fs.background_w2l;
    d.features = fs.features; d.max_ray_spawn = fs.max_ray_spawn; d.max_shadow_spawn = fs.max_shadow_spawn;
    d.inline_point_lights = fs.inline_point_lights;
    d.has_clear_to_light = 0;
    for (const DInstance& in : fs.instances) if (in.is_root & RT_INST_CLEAR_TO_LIGHT) d.has_clear_to_light = 1;
    {
        uint32_t chainable = 0, edged = 0;
        for (const DInstance& in : fs.instances) {
            const DPrim& p = fs.prims[in.prim];
            if (p.kind != DP_MESH && p.kind != DP_CSG && !p.wrap) ++chainable;
            if ((p.kind == DP_TRIANGLE || p.kind == DP_POLYGON || p.kind == DP_POLYHEDRON) && !p.wrap) ++edged;   // (one is enough: a ray along its edge, resolve_doubt)
        }
        s->ties_possible = chainable >= 2 || edged >= 1;
    }
    {   // (RT_WF_SHADE_CLS=0: every list on the general kernel, for tests that compare the two)
        const char* e = std::getenv("RT_WF_SHADE_CLS");
        const bool allow = !(e && std::atoi(e) == 0);
        const uint32_t glass = (1u << DS_REFLECTION) | (1u << DS_TRANSLUCENT) | (1u << DS_COMPOSITE);
        if (allow && fs.shade_kinds[0] == (1u << DS_PHONG) && fs.shade_depth[0] == 0) s->shade_cls[0] = 1;
        if (allow && fs.shade_kinds[1] != 0 && (fs.shade_kinds[1] & ~glass) == 0 && fs.shade_depth[1] <= 1) s->shade_cls[1] = 2;
        s->shade_none[0] = fs.shade_kinds[0] == 0; s->shade_none[1] = fs.shade_kinds[1] == 0;
    }
    {   // LDS-resident BVH when every mesh fits beside nothing else in one CU's LDS
        while (fs.lds_blob.size() % 16) fs.lds_blob.push_back(0);
        int lds_max = 0;
        (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device);
        s->lds_limit = lds_max > 0 ? (uint32_t)lds_max : 65536u;
        d.lds_blob = nullptr; d.lds_blob_bytes = 0;
        if ((fs.features & RTF_MESH) && fs.lds_exact && !fs.lds_blob.empty() && fs.lds_blob.size() + std::max<size_t>(RT_LDS_FIXED_BYTES(512u) + ((fs.features & RTF_INLINE_LIGHTS) ? RT_AREA_LDS_BYTES(8u) : 0u), RT_WF_WALK_FIXED_BYTES(1024u)) <= s->lds_limit) {
            rc = upload(fs.lds_blob, s->owned, &d.lds_blob);
            if (rc != RT_OK) { rt_scene_destroy(s); return rc; }
            d.lds_blob_bytes = (uint32_t)fs.lds_blob.size();
        }
    }
    for (auto& e : s->ev)
        if (hipEventCreate(&e) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipEventCreate failed"); }
    for (auto& st : s->chain_stream)
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipStreamCreate failed"); }
    for (auto& e : s->chain_join)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipEventCreate failed"); }
    if (hipEventCreateWithFlags(&s->chain_fork, hipEventDisableTiming) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipEventCreate failed"); }
    if (hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipEventCreate failed"); }
    if (s->wf_sticky.ensure(16) != RT_OK || hipMemset(s->wf_sticky.p, 0, 16) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipMalloc failed"); }
    if (hipHostMalloc((void**)&s->h_dropped, sizeof(unsigned int), hipHostMallocDefault) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_DEVICE, "hipHostMalloc failed"); }
    *s->h_dropped = 0;
    if (std::getenv("RT_DEBUG_INGEST")) {
        (void)hipDeviceSynchronize();
        std::fprintf(stderr, "[rt] scene create: flatten %.2f ms, total %.2f ms\n", std::chrono::duration<double, std::milli>(t_flat - t_begin).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    }
    *out = s;
    return RT_OK;
}

uint32_t rt_owned_rows_capacity(uint32_t height, uint32_t nranks, uint32_t tile_rows) {
    if (tile_rows == 0) tile_rows = 16;
    if (nranks <= 1) return height;
    uint32_t blocks = (height + tile_rows - 1) / tile_rows;
    return ((blocks + nranks - 1) / nranks) * tile_rows;
}

}  // extern "C"

namespace {

// Which variant (rt_variants.h) runs a scene: the smallest feature set that covers what it contains.
int launched(const LaunchStatus& st) {
    if (st.code != hipSuccess) return fail(RT_ERR_DEVICE, std::string(st.what) + ": " + hipGetErrorString(st.code));
    return RT_OK;
}
int launch_trace(uint32_t features, bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const TraceArgs& ta) {
    auto covers = [&](uint32_t v) { return (features & ~v) == 0; };
    if (covers(0u)) return launched(rt_launch_trace_plain(count, grid, stream, ds, rp, ta));
    if (covers(V_MESH) && ds.lds_blob_bytes) return launched(rt_launch_trace_mesh_lds(count, grid, stream, ds, rp, ta));
    if (covers(V_MESH)) return launched(rt_launch_trace_mesh(count, grid, stream, ds, rp, ta));
    if (covers(V_BALLS)) return launched(rt_launch_trace_balls(count, grid, stream, ds, rp, ta));
    if (covers(V_AREA)) return launched(rt_launch_trace_area(count, grid, stream, ds, rp, ta));
    if (covers(V_AREA_MESH) && ds.lds_blob_bytes) return launched(rt_launch_trace_area_mesh_lds(count, grid, stream, ds, rp, ta));
    if (covers(V_AREA_MESH)) return launched(rt_launch_trace_area_mesh(count, grid, stream, ds, rp, ta));
    if (covers(V_CSG)) return launched(rt_launch_trace_csg(count, grid, stream, ds, rp, ta));
    return launched(rt_launch_trace_everything(count, grid, stream, ds, rp, ta));
}

constexpr uint32_t RT_MAX_ROUNDS = 40;
// ---- queue-pipeline path (rt_wavefront.h): per-generation launches of the feature variant that covers the scene
int wf_launch(int which, uint32_t features, bool count, uint32_t grid, hipStream_t stream, const DScene& ds, const DRenderParams& rp, const WfArgs& wa) {
    auto covers = [&](uint32_t v) { return (features & ~v) == 0; };
    if (covers(0u)) return launched(rt_launch_wf_plain(which, count, grid, stream, ds, rp, wa));
    if (covers(V_MESH)) return launched(rt_launch_wf_mesh(which, count, grid, stream, ds, rp, wa));
    if (covers(V_BALLS)) return launched(rt_launch_wf_balls(which, count, grid, stream, ds, rp, wa));
    if (covers(V_CSG)) return launched(rt_launch_wf_csg(which, count, grid, stream, ds, rp, wa));
    return launched(rt_launch_wf_everything(which, count, grid, stream, ds, rp, wa));
}
int wf_launch_walk(bool count, uint32_t num_cus, hipStream_t stream, const DScene& ds, const WfArgs& wa) {
    return launched(rt_launch_wf_walk(count, num_cus, stream, ds, wa));
}
// u32 words of the pipeline's scalar block, per pass: queue fill [generation][rays, shadow tests], walk list fill and cursor per generation
constexpr uint32_t WF_MAX_GEN = RT_MAX_ROUNDS + 2;
constexpr size_t WF_CAP_MAX = (size_t)1 << 26;      // entries per queue array
constexpr uint32_t WF_N = 0, WF_WALK_COUNT = WF_N + 2 * RT_WF_Q * (WF_MAX_GEN + 1), WF_WALK_NEXT = WF_WALK_COUNT + RT_WF_Q * (WF_MAX_GEN + 1),
                   WF_PASS_WORDS = WF_WALK_NEXT + RT_WF_Q * (WF_MAX_GEN + 1);
constexpr uint32_t WF_HIT_COUNT = 2 * WF_PASS_WORDS;            // per pass 2 * RT_WF_Q * (WF_MAX_GEN + 1) words
constexpr uint32_t WF_CSG_COUNT = WF_HIT_COUNT + 2 * 2 * RT_WF_Q * (WF_MAX_GEN + 1);   // per pass 2 * RT_WF_Q * (WF_MAX_GEN + 1) words
constexpr size_t WF_SCALAR_BYTES = (WF_CSG_COUNT + 2 * 2 * RT_WF_Q * (WF_MAX_GEN + 1) + 4) * 4;
// RT_PATH=persistent forces the persistent kernel for every scene (read at every frame, so that a
// test can render the same scene down both schedules in one process).
bool use_wavefront(uint32_t features) {
    if (features & RTF_INLINE_LIGHTS) return false;      // visibility inside the shader: persistent kernel only
    const char* e = std::getenv("RT_PATH");
    return !(e && std::strcmp(e, "persistent") == 0);
}

// Layout of the per-frame scalar block (u32 words), zeroed at the start of every frame.
constexpr uint32_t SC_AA_COUNT = 0;
constexpr uint32_t SC_NEXT0 = 16, SC_NEXT1 = SC_NEXT0 + RT_MAX_ROUNDS + 1;          // work counters, one per round
constexpr uint32_t SC_QUEUE0 = SC_NEXT1 + RT_MAX_ROUNDS + 1, SC_QUEUE1 = SC_QUEUE0 + RT_MAX_ROUNDS + 1;   // queue fill, one per round
constexpr uint32_t SC_IDLE0 = SC_QUEUE1 + RT_MAX_ROUNDS + 1, SC_IDLE1 = SC_IDLE0 + RT_MAX_ROUNDS + 1;      // idle lanes, one per round
constexpr size_t SCALAR_BYTES = 2048;
static_assert((SC_IDLE1 + RT_MAX_ROUNDS + 1) * 4 <= SCALAR_BYTES, "scalar block too small");
uint32_t idle_divisor() {
    static const uint32_t d = [] { const char* e = std::getenv("RT_IDLE_DIVISOR"); long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)v : 8u; }();
    return d;
}

uint32_t trace_waves_per_cu() {   // waves of the persistent kernel per CU (kernel tuning experiments)
    static const uint32_t w = [] { const char* e = std::getenv("RT_TRACE_WAVES_PER_CU"); long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)std::min<long>(v, 32) : 8u; }();
    return w;
}
// Scene queries per lane per trip of the persistent kernel.  Several let the lanes walk the meshes for a
// whole batch at once; scenes whose lights are sampled inside the shader take one, so that all lanes of a
// wave reach the light together and share its quadtree walk (measured: light_room 47 -> 37 ms,
// monkey_1 129 -> 111 ms against four).
uint32_t batch_limit(uint32_t features) {
    static const uint32_t forced = [] { const char* e = std::getenv("RT_BATCH_LIMIT"); long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)std::min<long>(v, RT_BATCH) : 0u; }();
    if (forced) return forced;
    return (features & RTF_INLINE_LIGHTS) ? 1u : 4u;
}
uint32_t refill_min() {
    static const uint32_t b = [] { const char* e = std::getenv("RT_REFILL_MIN"); long v = e ? std::atol(e) : 0; return v > 0 ? (uint32_t)std::min<long>(v, 64) : 16u; }();
    return b;
}
uint32_t descend_frac() {
    static const uint32_t b = [] { const char* e = std::getenv("RT_DESCEND_FRAC"); long v = e ? std::atol(e) : 2; return (uint32_t)std::max<long>(0, std::min<long>(v, 8)); }();
    return b;
}
// Chains of the queue pipeline (1, 2 or 4): the sub-queues are split among them and each runs its
// generations on a stream of its own, so that one chain's launch tails (the last long BVH walks,
// the latency of small launches) are filled by the others.
uint32_t wf_chains(size_t pixels) {
    const char* e = std::getenv("RT_WF_CHAINS");                       // (read per frame: the tests switch it)
    const long forced = e ? std::atol(e) : 0L;
    if (forced > 0) return forced >= 4 ? 4u : (forced >= 2 ? 2u : 1u);
    return pixels >= (1u << 20) ? 2u : 4u;                            // measured: 1080p frame 58.7 / 54.9 / 55.8 ms with 1 / 2 / 4 chains, an eighth of it 15.0 / 13.6 / 13.2 ms
}
uint32_t trace_rounds(uint32_t recursion_depth, uint32_t features) {
    static const long forced = [] { const char* e = std::getenv("RT_TRACE_ROUNDS"); return e ? std::atol(e) : 0L; }();
    if (forced > 0) return (uint32_t)std::min<long>(forced, RT_MAX_ROUNDS);
    // Lights sampled inside the shader: a hit costs far more than the rays it spawns, and what a wave still holds
    // when the work runs out is little; one launch that runs to completion beats dealing the rest out again
    // (measured: light_room 37.7 -> 34.7 ms, monkey_1 106 -> 99 ms).
    if (features & RTF_INLINE_LIGHTS) return 1u;
    return std::min(recursion_depth + 1, RT_MAX_ROUNDS);
}

int check_args(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam, const RtRenderConfig* cfg) {
    if (!scene || !dim || !cam || !cfg) return fail(RT_ERR_INVALID_ARG, "rt_render: null argument");
    if (dim->width == 0 || dim->height == 0) return fail(RT_ERR_INVALID_ARG, "rt_render: empty image");
    if ((uint64_t)dim->width * dim->height > 0x7FFFFFFFull / 8) return fail(RT_ERR_CAPACITY, "rt_render: image larger than 2^28 pixels");
    if (cfg->interactive) return fail(RT_ERR_UNSUPPORTED, "rt_render: interactive mode is not part of the render path");
    if (cfg->nranks > 1 && cfg->rank >= cfg->nranks) return fail(RT_ERR_INVALID_ARG, "rt_render: rank >= nranks");
    if (cfg->anti_alias && cfg->aa_rays == 0) return fail(RT_ERR_INVALID_ARG, "rt_render: anti_alias with aa_rays == 0");
    if (cfg->anti_alias && cfg->aa_rays > 64) return fail(RT_ERR_CAPACITY, "rt_render: aa_rays > 64");
    return RT_OK;
}

// The whole frame, enqueued on `stream`.  Device-resident results: s->color / distance /
// hit_id / mask (full W*H arrays) and `d_rgb` (owned rows).
int enqueue_frame(RtScene* s, Workspace& w, const RtImageDim& dim, const RtCamera& cc, const RtRenderConfig& rc, uint8_t* d_rgb,
                  bool count_work, hipStream_t stream) {
    const uint32_t W = dim.width, H = dim.height;
    const size_t npix = (size_t)W * H;
    DRenderParams rp{};
    // camera basis, src/render.rs:108-124
    {
        double aspect_ratio = (double)W / (double)H;
        double fov_factor = std::tan((cc.fov_y * (M_PI / 180.0)) / 2.0);
        rp.cam.x_factor = aspect_ratio * fov_factor;
        rp.cam.fov_factor = fov_factor;
        auto norm = [](double* v) { double inv = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] *= inv; v[1] *= inv; v[2] *= inv; };
        auto cross = [](const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
        double view[3] = {cc.target[0] - cc.origin[0], cc.target[1] - cc.origin[1], cc.target[2] - cc.origin[2]};
        norm(view);
        double side[3]; cross(view, cc.up, side); norm(side);
        double up[3]; cross(side, view, up); norm(up);
        std::memcpy(rp.cam.origin, cc.origin, sizeof rp.cam.origin);
        std::memcpy(rp.cam.side, side, sizeof side); std::memcpy(rp.cam.up, up, sizeof up); std::memcpy(rp.cam.view, view, sizeof view);
        rp.cam.width = W; rp.cam.height = H;
    }
    rp.recursion_depth = rc.recursion_depth; rp.aa_rays = rc.aa_rays; rp.aa_threshold = rc.aa_threshold; rp.seed = rc.seed;
    rp.rank = rc.rank; rp.nranks = rc.nranks ? rc.nranks : 1; rp.tile_rows = rc.tile_rows ? rc.tile_rows : 16;

    // rows this rank renders: owned row blocks plus the AA halo.  The list is written by the device (k_frame_rows);
    // the host only counts them, so a frame needs neither an upload nor a synchronisation.
    const bool halo = rc.anti_alias != 0;
    uint32_t num_rows = 0;
    for (uint32_t y = 0; y < H; ++y) if (rt_rank_renders_row(y, H, rp.rank, rp.nranks, rp.tile_rows, halo)) ++num_rows;
    {
        int re = w.rows.ensure(std::max<size_t>(1, num_rows) * sizeof(uint32_t));
        if (re != RT_OK) return re;
        hipLaunchKernelGGL(k_frame_rows, dim3(1), dim3(256), 0, stream, H, rp.rank, rp.nranks, rp.tile_rows, halo ? 1u : 0u, (uint32_t*)w.rows.p);
        HIP_TRY(hipGetLastError());
    }

    int e;
    if ((e = w.color.ensure(npix * 3 * sizeof(double))) != RT_OK) return e;
    if ((e = w.distance.ensure(npix * sizeof(double))) != RT_OK) return e;
    if ((e = w.hit_id.ensure(npix * sizeof(uint32_t))) != RT_OK) return e;
    if ((e = w.mask.ensure(npix)) != RT_OK) return e;
    if ((e = w.scalars.ensure(SCALAR_BYTES + sizeof(DCounters))) != RT_OK) return e;
    unsigned int* scalars = static_cast<unsigned int*>(w.scalars.p);
    unsigned int* aa_count = scalars + SC_AA_COUNT;
    DCounters* counters = reinterpret_cast<DCounters*>(static_cast<char*>(w.scalars.p) + SCALAR_BYTES);
    HIP_TRY(hipMemsetAsync(w.scalars.p, 0, SCALAR_BYTES + sizeof(DCounters), stream));
    // hand-over queues between rounds, one accumulator per sample slot
    const uint32_t queue_cap = (uint32_t)std::min<size_t>(std::max<size_t>(npix * 4, 1u << 22), 1u << 25);
    const size_t acc_slots = npix * std::max(1u, rc.anti_alias ? rc.aa_rays : 1u);
    const bool aa_pass = rc.anti_alias && W >= 3 && H >= 3 && num_rows > 0;
    if (aa_pass) {   // worst case every owned interior pixel is flagged
        const size_t owned_px = (size_t)W * num_rows;
        if ((e = w.list.ensure(owned_px * sizeof(uint32_t))) != RT_OK) return e;
        if ((e = w.samples.ensure(owned_px * rc.aa_rays * 3 * sizeof(double))) != RT_OK) return e;
    }
    // A workspace over caller-provided memory (the entry point that allocates nothing): one chain of
    // launches, the BVH walk on at most 128 CUs (its overflow stacks are sized by that), and queues as
    // large as what the other buffers leave.
    const bool fixed_mem = w.color.arena != nullptr;
    const uint32_t walk_cus_max = fixed_mem ? std::min(128u, (uint32_t)s->num_cus) : (uint32_t)s->num_cus;
    const size_t walk_overflow_bytes = (size_t)(fixed_mem ? 1u : 4u) * walk_cus_max * 2048u * (RT_BVH_STACK - RT_WF_LDS_STACK) * 4u;   // per chain: a walk launch with up to 2048 threads per CU
    // (the static scratch array also holds the tie lists: RT_WF_Q * RT_WF_TIE_WORDS words, within the slack below)
    const size_t wf_entry_bytes = 2 * RT_WF_RAY_BYTES + 2 * RT_WF_SHADOW_BYTES + 2 * (JQ_NF * 8 + JQ_NU * 4) +
                                  2 * sizeof(uint32_t) +
                                  ((s->ds.features & RTF_CSG) ? 2 * sizeof(uint32_t) : 0);
    // queue pipeline: s->wf_scale entries per pixel (and queue) of this rank's share, every queue cut into RT_WF_Q sub-queues
    uint32_t wf_cap = (uint32_t)std::min<size_t>(std::max<size_t>((size_t)W * num_rows * s->wf_scale, 1u << 20), WF_CAP_MAX);
    if (const char* ce = std::getenv("RT_WF_CAP")) { long v = std::atol(ce); if (v >= (long)RT_WF_Q) wf_cap = (uint32_t)std::min<long>(v, (long)WF_CAP_MAX); }   // tests: provoke an overflow
    if (fixed_mem) {
        const size_t fixed = acc_slots * (RT_ACC_LIMBS * 8 + 4) + walk_overflow_bytes + WF_SCALAR_BYTES + (size_t)RT_WF_Q * RT_WF_TIE_WORDS * 4 + 16 * 256;   // what is still to come, with alignment slack
        const size_t left = w.color.arena->capacity > w.color.arena->used + fixed ? w.color.arena->capacity - w.color.arena->used - fixed : 0;
        wf_cap = (uint32_t)std::min<size_t>(std::min<size_t>(left / wf_entry_bytes, (size_t)W * num_rows * 32), WF_CAP_MAX);
    }
    wf_cap = (wf_cap / RT_WF_Q) * RT_WF_Q;
    // The pipeline runs recursion_depth + 1 generations out of a scalar block sized for WF_MAX_GEN, and
    // generation 0 (one ray per pixel) must fit the queues: frames beyond either run on the persistent kernel.
    const bool wavefront = use_wavefront(s->ds.features) && (fixed_mem || !s->queue_overflowed) && rc.recursion_depth + 1 <= WF_MAX_GEN &&
                           (size_t)W * num_rows <= wf_cap && wf_cap >= RT_WF_Q;
    if (fixed_mem && !wavefront) {
        if (!use_wavefront(s->ds.features) || rc.recursion_depth + 1 > WF_MAX_GEN)
            return fail(RT_ERR_UNSUPPORTED, "this scene / recursion depth needs the persistent-kernel schedule, whose per-lane memory the static scratch array does not hold");
        return fail(RT_ERR_CAPACITY, "frame workspace exceeds the static scratch array");
    }
    if (!wavefront) for (auto& q : w.queue) if ((e = q.ensure((size_t)queue_cap * sizeof(LaneRay))) != RT_OK) return e;
    {
        bool fresh = w.acc.bytes < acc_slots * RT_ACC_LIMBS * 8;
        if ((e = w.acc.ensure(acc_slots * RT_ACC_LIMBS * 8)) != RT_OK) return e;
        if ((e = w.acc_flags.ensure(acc_slots * 4)) != RT_OK) return e;
        if (fresh || w.acc_dirty) {
            HIP_TRY(hipMemsetAsync(w.acc.p, 0, w.acc.bytes, stream));
            HIP_TRY(hipMemsetAsync(w.acc_flags.p, 0, w.acc_flags.bytes, stream));
        }
        w.acc_dirty = true;
    }
    HIP_TRY(hipMemsetAsync(w.mask.p, 0, npix, stream));
    if (rp.nranks > 1) {   // rows never rendered by this rank must not look like data (their colour resolves to 0)
        HIP_TRY(hipMemsetAsync(w.distance.p, 0, npix * sizeof(double), stream));
        HIP_TRY(hipMemsetAsync(w.hit_id.p, 0xFF, npix * sizeof(uint32_t), stream));
    }

    // persistent grid: enough waves to fill every CU several times over; each pulls tiles until none are left
    const uint32_t tiles = ((W + 7) / 8) * ((num_rows + 7) / 8);
    const uint32_t max_waves = (uint32_t)s->num_cus * trace_waves_per_cu();   // default: two waves per SIMD, all the register file allows
    if (!wavefront && (e = w.lane_mem.ensure((size_t)max_waves * 64 * RT_LANE_BYTES)) != RT_OK) return e;
    if (!fixed_mem) HIP_TRY(hipEventRecord(s->ev[0], stream));
    // ray-stack bound: every level pops one ray and pushes at most max_ray_spawn, plus the shadow tests of the last hit
    // Queries per trip.  Depth-first order (one ray per trip) needs at most per_level x depth + spawn + 1
    // stack entries; several rays per trip are taken only while what they may push still leaves that much room.
    uint32_t batch = batch_limit(s->ds.features);
    while (batch > 1 && batch * s->ds.max_shadow_spawn > RT_MAX_SHADOW_JOBS) --batch;
    if (batch * s->ds.max_shadow_spawn > RT_MAX_SHADOW_JOBS) return fail(RT_ERR_CAPACITY, "rt_render: too many deferred shadow tests per hit");
    uint32_t wide_sp_limit = 0;
    {
        uint64_t per_level = s->ds.max_ray_spawn > 0 ? s->ds.max_ray_spawn - 1 : 0;
        uint64_t need1 = per_level * rc.recursion_depth + s->ds.max_ray_spawn + 1;
        if (need1 > RT_MAX_STACK) return fail(RT_ERR_CAPACITY, "rt_render: recursion_depth x rays-per-hit exceeds the device ray stack");
        uint64_t wide = need1 + (uint64_t)batch * s->ds.max_ray_spawn;
        wide_sp_limit = wide <= RT_MAX_STACK ? (uint32_t)(RT_MAX_STACK - wide) : 0u;
    }
    TraceArgs ta{};
    ta.rows = (const uint32_t*)w.rows.p; ta.num_rows = num_rows;
    ta.distance = (double*)w.distance.p; ta.hit_id = (uint32_t*)w.hit_id.p;
    ta.list = nullptr; ta.count = aa_count; ta.counters = counters; ta.mode = 0;
    ta.refill_min = refill_min();
    { const char* te = std::getenv("RT_WF_TIES"); ta.exact_ties = (s->ties_possible && !(te && std::atoi(te) == 0)) ? 1u : 0u; }   // (read per frame, like the queue pipeline's)
    ta.batch = batch; ta.wide_sp_limit = wide_sp_limit; ta.lane_mem = (uint8_t*)w.lane_mem.p;
    ta.queue_cap = queue_cap; ta.acc = (unsigned long long*)w.acc.p; ta.acc_flags = (uint32_t*)w.acc_flags.p;
    // Rounds: a launch ends once its work has all been dealt out, and what the lanes still hold is
    // dealt out again by the next one.  A ray handed on is traced in the next round, so a chain of
    // secondary rays advances at least one level per round: recursion_depth + 1 rounds reach every
    // ray, and the last round runs to completion in any case.  Rounds with an empty queue return at once.
    const uint32_t rounds = trace_rounds(rc.recursion_depth, s->ds.features);
    auto run_rounds = [&](uint32_t mode, uint32_t first_grid, double* dst) -> int {
        ta.mode = mode;
        for (uint32_t r = 0; r < rounds; ++r) {
            ta.round = r;
            ta.last_round = r + 1 == rounds ? 1u : 0u;
            ta.next_item = scalars + (mode == 0 ? SC_NEXT0 : SC_NEXT1) + r;
            ta.idle_count = scalars + (mode == 0 ? SC_IDLE0 : SC_IDLE1) + r;
            ta.idle_limit = std::max(1u, (r == 0 ? first_grid : max_waves) * 64u / idle_divisor());
            ta.in_items = (const LaneRay*)w.queue[r & 1].p;
            ta.in_count = scalars + (mode == 0 ? SC_QUEUE0 : SC_QUEUE1) + r;
            ta.out_items = (LaneRay*)w.queue[(r + 1) & 1].p;
            ta.out_count = scalars + (mode == 0 ? SC_QUEUE0 : SC_QUEUE1) + r + 1;
            int lrc = launch_trace(s->ds.features, count_work, r == 0 ? first_grid : max_waves, stream, s->ds, rp, ta);
            if (lrc != RT_OK) return lrc;
        }
        if (mode == 0)
            hipLaunchKernelGGL(k_acc_resolve, dim3(2048), dim3(256), 0, stream, ta.acc, ta.acc_flags, (const unsigned int*)nullptr, 0u, (uint32_t)npix, dst);
        else
            hipLaunchKernelGGL(k_acc_resolve, dim3(2048), dim3(256), 0, stream, ta.acc, ta.acc_flags, (const unsigned int*)aa_count, rc.aa_rays, 0u, dst);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    };
    // ---- the other schedule: ray queues in HBM, one generation of rays per set of launches
    WfArgs wa{};
    uint32_t gen0_tiles = 0;               // generation 0 dealt out in tiles: bit 0 pass 1, bit 1 AA resolve
    if (wavefront) {
        const size_t n = (size_t)(wf_cap / RT_WF_Q) * RT_WF_Q;    // entries per array; doubles first, then the words
        for (auto& b : w.wf_rays) if ((e = b.ensure(n * RT_WF_RAY_BYTES)) != RT_OK) return e;
        for (auto& b : w.wf_shadows) if ((e = b.ensure(n * RT_WF_SHADOW_BYTES)) != RT_OK) return e;
        if ((e = w.wf_rjobs.ensure(n * (JQ_NF * 8 + JQ_NU * 4))) != RT_OK) return e;
        if ((e = w.wf_sjobs.ensure(n * (JQ_NF * 8 + JQ_NU * 4))) != RT_OK) return e;
        if ((e = w.wf_walk_list.ensure(2 * n * sizeof(uint32_t))) != RT_OK) return e;
        if ((s->ds.features & RTF_CSG) && (e = w.wf_csg_list.ensure(2 * n * sizeof(uint32_t))) != RT_OK) return e;
        if ((e = w.wf_overflow.ensure(walk_overflow_bytes)) != RT_OK) return e;
        if ((e = w.wf_scalars.ensure(WF_SCALAR_BYTES)) != RT_OK) return e;
        HIP_TRY(hipMemsetAsync(w.wf_scalars.p, 0, WF_SCALAR_BYTES, stream));
        if ((e = w.wf_ties.ensure((size_t)RT_WF_Q * RT_WF_TIE_WORDS * 4)) != RT_OK) return e;
        HIP_TRY(hipMemsetAsync(w.wf_ties.p, 0, (size_t)RT_WF_Q * RT_WF_TIE_WORDS * 4, stream));
        wa.tie_list = (uint32_t*)w.wf_ties.p;
        w.wf_cap = (uint32_t)n;
        auto queue = [](const DeviceBuf& b, size_t entries, size_t nf) { WfQueue q; q.f = (double*)b.p; q.u = (uint32_t*)((char*)b.p + entries * nf * 8); return q; };
        wa.rjobs = queue(w.wf_rjobs, n, JQ_NF); wa.sjobs = queue(w.wf_sjobs, n, JQ_NF);
        wa.sub = wf_cap / RT_WF_Q; wa.cap = (uint32_t)n;
        wa.csg_list = (uint32_t*)w.wf_csg_list.p;
        wa.walk_list = (uint32_t*)w.wf_walk_list.p; wa.walk_overflow = (uint32_t*)w.wf_overflow.p; wa.refill_min = std::getenv("RT_REFILL_MIN") ? refill_min() : 24u; wa.descend_frac = descend_frac();   // (measured: 24 free lanes before the walk kernel refills, 16 in the persistent kernel)
        wa.acc = (unsigned long long*)w.acc.p; wa.acc_flags = (uint32_t*)w.acc_flags.p;
        wa.distance = (double*)w.distance.p; wa.hit_id = (uint32_t*)w.hit_id.p; wa.counters = counters;
        wa.dropped = (unsigned int*)s->wf_sticky.p;
        wa.rows = (const uint32_t*)w.rows.p; wa.num_rows = num_rows; wa.aa_count = aa_count;
        { const char* dbg = std::getenv("RT_WF_DEBUG"); wa.debug = dbg ? (uint32_t)std::atoi(dbg) : 0u; }
        {   // Generation 0 in tiles of 8 x 8 pixels (coherent waves) for scenes whose queries walk a hierarchy over the
            // instances in the analytic kernel, one lane each: there neighbouring rays save most (many_balls_forest 4K);
            // elsewhere ray-by-ray dealing keeps the sub-queues more even (glass_monkey: tiles cost 3 %).  RT_WF_TILES overrides.
            uint32_t tiles = s->ds.tlas_nodes != nullptr ? 3u : 0u;                        // bit 0: pass 1, bit 1: AA resolve
            if (const char* te = std::getenv("RT_WF_TILES")) tiles = (uint32_t)std::atoi(te) & 3u;
            const uint64_t tile_runs = (uint64_t)((W + 7u) / 8u) * ((num_rows + 7u) / 8u);
            if (((tile_runs + RT_WF_Q - 1u) / RT_WF_Q) * 64u > wa.sub) tiles &= ~1u;      // the padded frame must fit generation 0's sub-queues
            gen0_tiles = tiles;
        }
    }
    auto run_wavefront = [&](uint32_t mode, double* dst) -> int {
        unsigned int* ws = (unsigned int*)w.wf_scalars.p + mode * WF_PASS_WORDS;
        const uint32_t generations = std::min(rc.recursion_depth + 1, WF_MAX_GEN);
        const uint32_t grid = std::max(1u, (uint32_t)s->num_cus * 16 / RT_WF_Q) * RT_WF_Q;   // a whole number of workgroups per sub-queue
        wa.mode = mode;
        wa.gen0_tiles = (gen0_tiles >> mode) & 1u;
        wa.list = (const uint32_t*)w.list.p;
        // generation 0 into queue set 0
        auto ray_queue = [](const DeviceBuf& b) { RayQueue q; q.f = (double*)b.p; q.u = nullptr; return q; };
        auto shadow_queue = [](const DeviceBuf& b) { ShadowQueue q; q.f = (double*)b.p; q.u = nullptr; return q; };
        const size_t n = wa.cap;
        wa.out_rays = ray_queue(w.wf_rays[0]); wa.out_shadows = shadow_queue(w.wf_shadows[0]); wa.n_out = ws + WF_N;
        hipLaunchKernelGGL(k_wf_generate, dim3(grid), dim3(256), 0, stream, rp, wa);
        HIP_TRY(hipGetLastError());
        uint32_t chains = fixed_mem ? 1u : wf_chains((size_t)W * num_rows);
        if (!fixed_mem) { const char* pe = std::getenv(mode == 0 ? "RT_WF_CHAINS_P0" : "RT_WF_CHAINS_P1"); if (pe) { long v = std::atol(pe); chains = v >= 4 ? 4u : (v >= 2 ? 2u : 1u); } }
        const char* te = std::getenv("RT_WF_TIES");                    // (read per frame; 0: ties stay as the 8-ulp rule decided them, as on the persistent kernel)
        const bool ties = s->ties_possible && !(te && std::atoi(te) == 0);
        uint32_t* const overflow0 = wa.walk_overflow;
        if (chains > 1) {                                             // the other chains start once generation 0 is in the queues
            HIP_TRY(hipEventRecord(s->chain_fork, stream));
            for (uint32_t c = 1; c < chains; ++c) HIP_TRY(hipStreamWaitEvent(s->chain_stream[c - 1], s->chain_fork, 0));
        }
        for (uint32_t g = 0; g < generations; ++g) {
            wa.rays = ray_queue(w.wf_rays[g & 1]); wa.shadows = shadow_queue(w.wf_shadows[g & 1]); wa.n_in = ws + WF_N + 2 * RT_WF_Q * g;
            wa.out_rays = ray_queue(w.wf_rays[(g + 1) & 1]); wa.out_shadows = shadow_queue(w.wf_shadows[(g + 1) & 1]); wa.n_out = ws + WF_N + 2 * RT_WF_Q * (g + 1);
            wa.walk_count = ws + WF_WALK_COUNT + RT_WF_Q * g;
            wa.csg_count = (unsigned int*)w.wf_scalars.p + WF_CSG_COUNT + (mode * (WF_MAX_GEN + 1) + g) * 2 * RT_WF_Q;
            wa.hit_count = (unsigned int*)w.wf_scalars.p + WF_HIT_COUNT + (mode * (WF_MAX_GEN + 1) + g) * 2 * RT_WF_Q;
            for (uint32_t c = 0; c < chains; ++c) {
                hipStream_t cs = c == 0 ? stream : s->chain_stream[c - 1];
                // CUs for the chain's walk kernel (one workgroup fills a CU's LDS).  Two chains: three
                // quarters each - the walks of the two chains rarely coincide, and when they do the
                // later one's surplus workgroups queue (measured 54.7 ms at one half, 53.7 ms at 3/4,
                // 56.4 ms at all CUs); four chains: a quarter each.
                static const long walk_cus = [] { const char* e = std::getenv("RT_WF_WALK_CUS"); return e ? std::atol(e) : 0L; }();
                const uint32_t share = chains == 2 ? (uint32_t)s->num_cus * 3u / 4u : (uint32_t)s->num_cus / chains;
                const uint32_t cus = std::min(walk_cus_max, std::max(1u, walk_cus > 0 && chains > 1 ? (uint32_t)std::min<long>(walk_cus, s->num_cus) : share));
                wa.sq_count = RT_WF_Q / chains; wa.sq_base = c * wa.sq_count;
                wa.walk_next = ws + WF_WALK_NEXT + RT_WF_Q * g + c;   // (a cursor per chain)
                wa.walk_overflow = overflow0 + (size_t)c * walk_cus_max * 2048u * (RT_BVH_STACK - RT_WF_LDS_STACK);   // (a walk launch has at most 2048 threads per CU)
                const uint32_t cgrid = grid / chains;
                int lrc = wf_launch(0, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                if (lrc != RT_OK) return lrc;
                if (ties) {                                              // surfaces that can meet at one distance: decide noted ties the reference's way
                    lrc = wf_launch(6, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                    if (lrc != RT_OK) return lrc;
                }
                if (s->ds.features & RTF_CSG) {                          // the queries that reach a composite shape
                    lrc = wf_launch(3, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                    if (lrc != RT_OK) return lrc;
                }
                if (s->ds.features & RTF_MESH) {
                    lrc = wf_launch_walk(count_work, cus, cs, s->ds, wa);
                    if (lrc != RT_OK) return lrc;
                }
                lrc = wf_launch(2, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                if (lrc != RT_OK) return lrc;
                if (s->shade_cls[0] == 0 && s->shade_cls[1] == 0) {       // both hit lists on the general kernel, in one launch
                    wa.shade_lists = 3u;
                    lrc = wf_launch(1, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                    if (lrc != RT_OK) return lrc;
                } else {                                                 // a launch per list, each on the kernel for its materials
                    // (a list no material of the scene belongs on stays empty: even so its launch would cost tens of
                    // microseconds per generation, the general kernel's 960 bytes of scratch per lane to be set up)
                    wa.shade_lists = 2u;
                    lrc = s->shade_none[1] ? RT_OK : wf_launch(s->shade_cls[1] ? 3 + s->shade_cls[1] : 1, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                    if (lrc != RT_OK) return lrc;
                    wa.shade_lists = 1u;
                    lrc = s->shade_none[0] ? RT_OK : wf_launch(s->shade_cls[0] ? 3 + s->shade_cls[0] : 1, s->ds.features, count_work, cgrid, cs, s->ds, rp, wa);
                    if (lrc != RT_OK) return lrc;
                }
            }
        }
        wa.walk_overflow = overflow0;
        for (uint32_t c = 1; c < chains; ++c) {                       // the frame goes on when every chain is through
            HIP_TRY(hipEventRecord(s->chain_join[c - 1], s->chain_stream[c - 1]));
            HIP_TRY(hipStreamWaitEvent(stream, s->chain_join[c - 1], 0));
        }
        if (mode == 0)
            hipLaunchKernelGGL(k_acc_resolve, dim3(2048), dim3(256), 0, stream, wa.acc, wa.acc_flags, (const unsigned int*)nullptr, 0u, (uint32_t)npix, dst);
        else
            hipLaunchKernelGGL(k_acc_resolve, dim3(2048), dim3(256), 0, stream, wa.acc, wa.acc_flags, (const unsigned int*)aa_count, rc.aa_rays, 0u, dst);
        HIP_TRY(hipGetLastError());
        return RT_OK;
    };
    if (num_rows > 0) {
        int lrc = wavefront ? run_wavefront(0, (double*)w.color.p) : run_rounds(0, std::min(tiles, max_waves), (double*)w.color.p);
        if (lrc != RT_OK) return lrc;
    }
    if (!fixed_mem) HIP_TRY(hipEventRecord(s->ev[1], stream));
    if (aa_pass) {
        uint32_t blocks = (uint32_t)((npix + 255) / 256);
        hipLaunchKernelGGL(k_aa_detect, dim3(blocks), dim3(256), 0, stream, rp, (const double*)w.color.p, (const double*)w.distance.p,
                           (uint32_t*)w.list.p, aa_count, (uint8_t*)w.mask.p);
        HIP_TRY(hipGetLastError());
        ta.list = (const uint32_t*)w.list.p;
        int lrc = wavefront ? run_wavefront(1, (double*)w.samples.p) : run_rounds(1, max_waves, (double*)w.samples.p);
        if (lrc != RT_OK) return lrc;
        hipLaunchKernelGGL(k_aa_combine, dim3(1024), dim3(256), 0, stream, rp, (const uint32_t*)w.list.p, aa_count,
                           (const double*)w.samples.p, (double*)w.color.p, (double*)w.distance.p);
        HIP_TRY(hipGetLastError());
    }
    if (!fixed_mem) HIP_TRY(hipEventRecord(s->ev[2], stream));
    if (d_rgb) {
        uint32_t blocks = (uint32_t)((npix + 255) / 256);
        hipLaunchKernelGGL(k_encode, dim3(blocks), dim3(256), 0, stream, rp, (const double*)w.color.p, d_rgb);
        HIP_TRY(hipGetLastError());
    }
    if (!fixed_mem) HIP_TRY(hipEventRecord(s->ev[3], stream));
    s->last_used_queues = wavefront;
    if (wavefront) HIP_TRY(hipMemcpyAsync(s->h_dropped, s->wf_sticky.p, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    // (a frame in the static scratch array is made of kernel launches, memsets and one copy to pinned memory only: no
    // events of the scene's, so that the sequence can be captured into a HIP graph)
    if (!fixed_mem) HIP_TRY(hipEventRecord(s->ev_done, stream));
    s->timed = true;
    w.acc_dirty = false;
    return RT_OK;
}

// Acts on entries the queue pipeline lost since the host last looked: the next frame gets queues twice
// as large, or, when they cannot grow any further, the persistent kernel (which hands rays on between
// rounds and cannot overflow).  Returns whether there were any.
bool note_drops(RtScene* s) {
    const unsigned int now = *s->h_dropped;
    if (now == s->dropped_seen) return false;
    s->dropped_seen = now;
    if (s->last_noalloc) return true;
    if (s->wf_scale < 32 && !std::getenv("RT_WF_CAP")) s->wf_scale *= 2;
    else s->queue_overflowed = true;
    return true;
}
// One frame whose completeness is known on return: enqueues it, waits, and redoes it while the queue
// pipeline reports lost entries (each attempt with larger queues, the last one on the persistent kernel).
int render_complete_frame(RtScene* s, const RtImageDim& dim, const RtCamera& cam, const RtRenderConfig& cfg, uint8_t* d_rgb,
                          bool count, hipStream_t stream) {
    for (int attempt = 0; attempt < 8; ++attempt) {
        s->last_attempts = attempt + 1;
        int rc = enqueue_frame(s, s->ws, dim, cam, cfg, d_rgb, count, stream);
        if (rc != RT_OK) return rc;
        s->last_noalloc = false;
        HIP_TRY(hipStreamSynchronize(stream));
        if (!s->last_used_queues || !note_drops(s)) { s->dropped_reported = s->dropped_seen; return RT_OK; }
    }
    return fail(RT_ERR_CAPACITY, "rt_render: the frame keeps overflowing the ray queues");
}

void fill_counters(RtCounters& out, const DCounters& c) {
    std::memset(&out, 0, sizeof out);
    out.rays_primary = c.rays_primary; out.rays_secondary = c.rays_secondary; out.rays_shadow = c.rays_shadow; out.rays_aa = c.rays_aa;
    out.aa_pixels = c.aa_pixels; out.instance_tests = c.instance_tests; out.box_tests = c.box_tests; out.tri_tests = c.tri_tests;
    out.shade_evals = c.shade_evals; out.stack_overflows = c.stack_overflows;
    out.box_tests_f32 = c.box_tests_f32; out.csg_events = c.csg_events;
    // Algorithmic flop estimate: fixed cost per unit of counted work -
    // ray->instance transform + analytic primitive test 48, BVH box slab test 24, mesh triangle
    // test 60, Phong per lit sample 40, hit->world + shading set-up 45 per closest-hit ray,
    // reflect/refract/Fresnel 30 per secondary ray, composite-shape event 20.  `flops` is the sum; the box
    // tests of the LDS-resident BVH walk run in packed f32 and are priced apart (flops_f32), against the
    // f32 vector rate.
    out.flops_f32 = 24 * c.box_tests_f32;
    out.flops = 48 * c.instance_tests + 24 * c.box_tests + 60 * c.tri_tests + 40 * c.shade_evals +
                45 * (c.rays_primary + c.rays_secondary + c.rays_aa) + 30 * c.rays_secondary + 20 * c.csg_events;
}

}  // namespace

extern "C" {

int rt_render_device(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam, const RtRenderConfig* cfg,
                     uint8_t* d_rgb_out, RtAux* d_aux, void* hip_stream) {
    int rc = check_args(scene, dim, cam, cfg);
    if (rc != RT_OK) return rc;
    RtScene* s = const_cast<RtScene*>(scene);
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    bool count = d_aux && d_aux->count_work;
    // an earlier asynchronous frame that lost rays (known once it has finished): larger queues from now on
    if (s->last_used_queues) (void)note_drops(s);
    if (count) rc = render_complete_frame(s, *dim, *cam, *cfg, d_rgb_out, true, stream);   // this variant of the call synchronises anyway
    else rc = enqueue_frame(s, s->ws, *dim, *cam, *cfg, d_rgb_out, false, stream);
    if (rc != RT_OK) return rc;
    s->last_noalloc = false;
    if (d_aux) {
        const size_t npix = (size_t)dim->width * dim->height;
        if (d_aux->color) HIP_TRY(hipMemcpyAsync(d_aux->color, s->ws.color.p, npix * 3 * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (d_aux->distance) HIP_TRY(hipMemcpyAsync(d_aux->distance, s->ws.distance.p, npix * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (d_aux->hit_id) HIP_TRY(hipMemcpyAsync(d_aux->hit_id, s->ws.hit_id.p, npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        if (d_aux->aa_mask) HIP_TRY(hipMemcpyAsync(d_aux->aa_mask, s->ws.mask.p, npix, hipMemcpyDeviceToDevice, stream));
        if (count) {   // counters are host-visible results: this variant of the call synchronises
            DCounters c{};
            unsigned int aa = 0;
            HIP_TRY(hipMemcpyAsync(&c, static_cast<char*>(s->ws.scalars.p) + SCALAR_BYTES, sizeof c, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&aa, static_cast<char*>(s->ws.scalars.p) + SC_AA_COUNT * 4, sizeof aa, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            c.aa_pixels = aa;
            fill_counters(d_aux->counters, c);
            d_aux->counters.schedule = s->last_used_queues ? 1 : 2; d_aux->counters.attempts = (uint64_t)s->last_attempts;
        }
    }
    return RT_OK;
}

int rt_render_device_noalloc(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam, const RtRenderConfig* cfg,
                             uint8_t* d_rgb_out, void* hip_stream) {
    int rc = check_args(scene, dim, cam, cfg);
    if (rc != RT_OK) return rc;
    if (!d_rgb_out) return fail(RT_ERR_INVALID_ARG, "rt_render_device_noalloc: d_rgb_out is null");
    if ((uint64_t)dim->width * dim->height > RT_NOALLOC_MAX_PIXELS)
        return fail(RT_ERR_CAPACITY, "rt_render_device_noalloc: frame larger than RT_NOALLOC_MAX_PIXELS (" + std::to_string(RT_NOALLOC_MAX_PIXELS) + " pixels)");
    RtScene* s = const_cast<RtScene*>(scene);
    std::lock_guard<std::mutex> lk(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    void* base = nullptr;
    { std::lock_guard<std::mutex> gl(g_scratch_mu); base = g_scratch_ptr[s->device]; }
    if (!base) return fail(RT_ERR_DEVICE, "rt_render_device_noalloc: the scratch array of this device was never looked up");
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(stream, &capture));
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (s->last_used_queues && !capturing) (void)note_drops(s);
    Arena arena;
    arena.base = static_cast<uint8_t*>(base); arena.capacity = RT_NOALLOC_SCRATCH_BYTES; arena.used = 0;
    Workspace w;                         // every buffer of the frame: bump-allocated in the arena, all of it re-initialised by the frame
    w.use(&arena);
    // Frames that use the array take turns on it: each waits for the event recorded behind the one before, whichever
    // scene and stream that was on.  (Not under stream capture, where an event of another stream cannot be waited for:
    // graphs that contain this call must be kept apart by their owner.)
    std::lock_guard<std::mutex> turn(g_scratch_mu);
    if (!capturing) HIP_TRY(hipStreamWaitEvent(stream, g_scratch_free[s->device], 0));
    rc = enqueue_frame(s, w, *dim, *cam, *cfg, d_rgb_out, false, stream);
    if (rc != RT_OK) return rc;
    if (!capturing) HIP_TRY(hipEventRecord(g_scratch_free[s->device], stream));
    s->last_noalloc = true;
    return RT_OK;
}

int rt_last_timings(const RtScene* scene, double* ms_pass1, double* ms_aa, double* ms_encode) {
    if (!scene) return fail(RT_ERR_INVALID_ARG, "rt_last_timings: null scene");
    RtScene* s = const_cast<RtScene*>(scene);
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->timed) return fail(RT_ERR_INVALID_ARG, "rt_last_timings: no frame rendered yet");
    if (s->last_noalloc) return fail(RT_ERR_UNSUPPORTED, "rt_last_timings: frames of rt_render_device_noalloc record no events");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev[3]));
    float a = 0, b = 0, c = 0;
    HIP_TRY(hipEventElapsedTime(&a, s->ev[0], s->ev[1]));
    HIP_TRY(hipEventElapsedTime(&b, s->ev[1], s->ev[2]));
    HIP_TRY(hipEventElapsedTime(&c, s->ev[2], s->ev[3]));
    if (ms_pass1) *ms_pass1 = a;
    if (ms_aa) *ms_aa = b;
    if (ms_encode) *ms_encode = c;
    return RT_OK;
}

int rt_frame_status(const RtScene* scene, uint64_t* dropped) {
    if (!scene) return fail(RT_ERR_INVALID_ARG, "rt_frame_status: null scene");
    RtScene* s = const_cast<RtScene*>(scene);
    std::lock_guard<std::mutex> lk(s->mu);
    if (dropped) *dropped = 0;
    if (!s->timed) return fail(RT_ERR_INVALID_ARG, "rt_frame_status: no frame rendered yet");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->last_noalloc) HIP_TRY(hipEventSynchronize(s->ev_done));   // (a frame of rt_render_device_noalloc records no event: its caller has synchronised the stream)
    if (s->last_used_queues) (void)note_drops(s);
    const unsigned int lost = s->dropped_seen - s->dropped_reported;
    s->dropped_reported = s->dropped_seen;
    if (dropped) *dropped = lost;
    if (lost) return fail(RT_ERR_CAPACITY, "a ray generation outgrew the queues: the frame lacks " + std::to_string(lost) +
                                           " rays / shadow tests; render it again (the queues have been enlarged)");
    return RT_OK;
}

int rt_render(const RtScene* scene, const RtImageDim* dim, const RtCamera* cam, const RtRenderConfig* cfg,
              uint8_t* rgb_out, RtAux* aux) {
    int rc = check_args(scene, dim, cam, cfg);
    if (rc != RT_OK) return rc;
    if (!rgb_out) return fail(RT_ERR_INVALID_ARG, "rt_render: rgb_out is null");
    RtScene* s = const_cast<RtScene*>(scene);
    const size_t npix = (size_t)dim->width * dim->height;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        HIP_TRY(hipSetDevice(s->device));
        uint32_t rows_cap = rt_owned_rows_capacity(dim->height, cfg->nranks, cfg->tile_rows);
        size_t rgb_bytes = (size_t)rows_cap * dim->width * 3;
        int e = s->ws.rgb_full.ensure(rgb_bytes);
        if (e != RT_OK) return e;
        HIP_TRY(hipMemsetAsync(s->ws.rgb_full.p, 0, rgb_bytes, nullptr));
        bool count = aux && aux->count_work;
        if (s->last_used_queues) (void)note_drops(s);
        rc = render_complete_frame(s, *dim, *cam, *cfg, static_cast<uint8_t*>(s->ws.rgb_full.p), count, nullptr);
        if (rc != RT_OK) return rc;
        HIP_TRY(hipMemcpy(rgb_out, s->ws.rgb_full.p, rgb_bytes, hipMemcpyDeviceToHost));
        if (aux) {
            if (aux->color) HIP_TRY(hipMemcpy(aux->color, s->ws.color.p, npix * 3 * sizeof(double), hipMemcpyDeviceToHost));
            if (aux->distance) HIP_TRY(hipMemcpy(aux->distance, s->ws.distance.p, npix * sizeof(double), hipMemcpyDeviceToHost));
            if (aux->hit_id) HIP_TRY(hipMemcpy(aux->hit_id, s->ws.hit_id.p, npix * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (aux->aa_mask) HIP_TRY(hipMemcpy(aux->aa_mask, s->ws.mask.p, npix, hipMemcpyDeviceToHost));
            DCounters c{};
            unsigned int aa = 0;
            HIP_TRY(hipMemcpy(&c, static_cast<char*>(s->ws.scalars.p) + SCALAR_BYTES, sizeof c, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(&aa, static_cast<char*>(s->ws.scalars.p) + SC_AA_COUNT * 4, sizeof aa, hipMemcpyDeviceToHost));
            c.aa_pixels = aa;
            fill_counters(aux->counters, c);
            aux->counters.schedule = s->last_used_queues ? 1 : 2; aux->counters.attempts = (uint64_t)s->last_attempts;
        }
        if (std::getenv("RT_DEBUG_ROUNDS") && aux && aux->count_work) {   // lane occupancy of the trace kernel's phases
            DCounters c{};
            HIP_TRY(hipMemcpy(&c, static_cast<char*>(s->ws.scalars.p) + SCALAR_BYTES, sizeof c, hipMemcpyDeviceToHost));
            const char* names[8] = {"trips", "phase A jobs", "walk descend", "walk refill", "walk leaf", "phase B rays", "", ""};
            for (int i = 0; i < 6; ++i)
                std::fprintf(stderr, "[rt] %-14s wave trips %12llu  lanes/trip %5.1f\n", names[i], c.prof[2 * i], c.prof[2 * i] ? (double)c.prof[2 * i + 1] / (double)c.prof[2 * i] : 0.0);
            double tot = (double)(c.prof[PF_CYC_A] + c.prof[PF_CYC_T] + c.prof[PF_CYC_B] + c.prof[PF_CYC_FETCH]) + 1e-30;
            std::fprintf(stderr, "[rt] wave time: A %.1f%% T %.1f%% B %.1f%% fetch/hand-on %.1f%%\n", 100.0 * c.prof[PF_CYC_A] / tot, 100.0 * c.prof[PF_CYC_T] / tot,
                         100.0 * c.prof[PF_CYC_B] / tot, 100.0 * c.prof[PF_CYC_FETCH] / tot);
        }
        if (std::getenv("RT_DEBUG_ROUNDS") && aux && aux->count_work && s->last_used_queues && std::getenv("RT_WF_DEBUG") && (std::atoi(std::getenv("RT_WF_DEBUG")) & 16)) {
            DCounters c{};
            HIP_TRY(hipMemcpy(&c, static_cast<char*>(s->ws.scalars.p) + SCALAR_BYTES, sizeof c, hipMemcpyDeviceToHost));
            const double tot = (double)(c.prof[PF_CYC_A] + c.prof[PF_CYC_T] + c.prof[PF_CYC_B]) + 1e-30;
            std::fprintf(stderr, "[rt] walk kernel, wave cycles: refill %.1f%% (%.0f per trip), descent %.1f%% (%.0f per step), leaves %.1f%% (%.0f per trip)\n",
                         100.0 * c.prof[PF_CYC_A] / tot, (double)c.prof[PF_CYC_A] / ((double)c.prof[PF_T_REFILLS] + 1e-30),
                         100.0 * c.prof[PF_CYC_T] / tot, (double)c.prof[PF_CYC_T] / ((double)c.prof[PF_T_ITERS] + 1e-30),
                         100.0 * c.prof[PF_CYC_B] / tot, (double)c.prof[PF_CYC_B] / ((double)c.prof[PF_T_LEAF_ITERS] + 1e-30));
        }
        if (std::getenv("RT_DEBUG_ROUNDS") && aux && aux->count_work && s->last_used_queues) {   // where the shade kernel's waves spend their time
            DCounters c{};
            HIP_TRY(hipMemcpy(&c, static_cast<char*>(s->ws.scalars.p) + SCALAR_BYTES, sizeof c, hipMemcpyDeviceToHost));
            const double it = (double)c.prof[PF_TRIPS] + 1e-30;
            std::fprintf(stderr, "[rt] shade: %llu wave trips, %.1f hits each; cycles per trip: list+instance %.0f, reservation %.0f, ray+world hit %.0f, material %.0f, holes %.0f, accumulation %.0f; start-up %llu\n",
                         c.prof[PF_TRIPS], (double)c.prof[PF_TRIP_LANES] / it, c.prof[PF_A_ITERS] / it, c.prof[PF_A_LANES] / it, c.prof[PF_B_ITERS] / it,
                         c.prof[PF_B_LANES] / it, c.prof[PF_CYC_A] / it, c.prof[PF_CYC_T] / it, c.prof[PF_CYC_FETCH]);
        }
        if (std::getenv("RT_DEBUG_ROUNDS") && s->ws.wf_scalars.p && s->last_used_queues) {   // queue fill per generation
            std::vector<unsigned int> ws(WF_SCALAR_BYTES / 4);
            HIP_TRY(hipMemcpy(ws.data(), s->ws.wf_scalars.p, WF_SCALAR_BYTES, hipMemcpyDeviceToHost));
            for (int pass = 0; pass < 2; ++pass) {
                std::fprintf(stderr, "[rt] pass %d rays+shadow tests per generation (largest sub-queue):", pass);
                for (uint32_t g = 0; g <= RT_MAX_ROUNDS; ++g) {
                    unsigned long long nr = 0, ns = 0; unsigned int mx = 0;
                    for (uint32_t q = 0; q < RT_WF_Q; ++q) {
                        unsigned int a0 = ws[pass * WF_PASS_WORDS + WF_N + 2 * RT_WF_Q * g + 2 * q], a1 = ws[pass * WF_PASS_WORDS + WF_N + 2 * RT_WF_Q * g + 2 * q + 1];
                        nr += a0; ns += a1; mx = std::max(mx, std::max(a0, a1));
                    }
                    if (nr + ns == 0) break;
                    std::fprintf(stderr, " %llu+%llu(%u)", nr, ns, mx);
                }
                std::fprintf(stderr, "\n");
            }
            std::fprintf(stderr, "[rt] sub-queue capacity %u, dropped %u\n", (unsigned)(s->ws.wf_cap / RT_WF_Q), *s->h_dropped);
        }
        if (std::getenv("RT_DEBUG_ROUNDS")) {   // how many rays each round handed on
            unsigned int sc[SCALAR_BYTES / 4];
            HIP_TRY(hipMemcpy(sc, s->ws.scalars.p, SCALAR_BYTES, hipMemcpyDeviceToHost));
            std::fprintf(stderr, "[rt] queue pass1:");
            for (uint32_t r = 1; r <= RT_MAX_ROUNDS && sc[SC_QUEUE0 + r]; ++r) std::fprintf(stderr, " %u", sc[SC_QUEUE0 + r]);
            std::fprintf(stderr, "\n[rt] queue aa:");
            for (uint32_t r = 1; r <= RT_MAX_ROUNDS && sc[SC_QUEUE1 + r]; ++r) std::fprintf(stderr, " %u", sc[SC_QUEUE1 + r]);
            std::fprintf(stderr, "\n");
        }
    }
    double ms_a = 0, ms_b = 0, ms_c = 0;
    const bool have_ms = rt_last_timings(scene, &ms_a, &ms_b, &ms_c) == RT_OK;
    if (aux && have_ms) { aux->ms_pass1 = ms_a; aux->ms_aa = ms_b; aux->ms_encode = ms_c; }
    RtProgressFn progress = nullptr;
    void* progress_user = nullptr;
    { std::lock_guard<std::mutex> lk(s->mu); progress = s->progress; progress_user = s->progress_user; }
    if (progress) {   // the stages of src/progress_tracker.rs:59-73, after the fact
        unsigned int flagged = 0;
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (cfg->anti_alias) HIP_TRY(hipMemcpy(&flagged, static_cast<char*>(s->ws.scalars.p) + SC_AA_COUNT * 4, sizeof flagged, hipMemcpyDeviceToHost));
        }
        const uint64_t traced = (uint64_t)rt_owned_rows_capacity(dim->height, cfg->nranks, cfg->tile_rows) * dim->width;
        progress(progress_user, RT_PROGRESS_START, 0, npix, 0.0);
        progress(progress_user, RT_PROGRESS_PASS1, std::min<uint64_t>(traced, npix), npix, ms_a * 1e-3);
        if (cfg->anti_alias) {
            progress(progress_user, RT_PROGRESS_START_AA, 0, flagged, ms_a * 1e-3);
            progress(progress_user, RT_PROGRESS_AA, flagged, flagged, (ms_a + ms_b) * 1e-3);
        }
        progress(progress_user, RT_PROGRESS_END, npix, npix, (ms_a + ms_b + ms_c) * 1e-3);
    }
    return RT_OK;
}

int rt_set_progress(RtScene* scene, RtProgressFn fn, void* user) {
    if (!scene) return fail(RT_ERR_INVALID_ARG, "rt_set_progress: null scene");
    std::lock_guard<std::mutex> lk(scene->mu);
    scene->progress = fn; scene->progress_user = user;
    return RT_OK;
}

}  // extern "C"
