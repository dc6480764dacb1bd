// rt_wavefront.h — the render path as a pipeline of small kernels over ray queues in HBM.
//
// Same arithmetic as the persistent trace kernel (rt_kernels.h), other schedule: the rays of a
// frame are processed generation by generation (camera rays, then everything they spawned, ...)
// and every generation runs through four launches
//     k_wf_analytic   one thread per query (ray or shadow test): gate of Scene::cast_ray, every
//                     primitive except the un-wrapped meshes; queries whose ray reaches a mesh's
//                     bounding box go onto the walk list
//     k_wf_walk       persistent lanes take queries off the walk list and walk the mesh BVHs
//                     (LDS-resident), replacing a finished query at once
//     k_wf_settle     one thread per query: the colour terms of shadow tests that nothing blocks
//                     and of rays that left the scene go to the accumulators; hits are listed
//     k_wf_shade      one thread per listed hit: hit -> world space, shader tree, new rays and
//                     shadow tests appended to the next generation's queues, the hit's own
//                     colour term to the accumulators
// (in sky-box scenes rays that failed the gate still see the background: the analytic kernel looks it up).
// Every lane of every kernel has a query of the same kind to work on, each kernel only holds
// the registers its phase needs (more waves per SIMD to hide the queue traffic), and the walk
// draws on millions of queries so that one long walk no longer holds up a wave.
// Used for scenes without area / directional lights (those need visibility answers inside
// the shader and stay on the persistent kernel).
#pragma once
#include "rt_kernels.h"

#define RT_WF_LDS_STACK 8      // BVH walk stack entries per thread kept in LDS by k_wf_walk (measured 6 / 8 / 12: 49.3 / 49.1 / 49.45 ms; the LDS the shorter stack leaves free lets other kernels' workgroups share the CU)
// What the walk needs to know of a mesh instance (DInstance -> DPrim -> DMesh), gathered into one record:
// the first RT_WF_WALK_TABLE of them are built once per workgroup and kept in LDS, so that a lane
// that takes a new query reads them with one hop instead of a chain of four dependent global loads.
struct WalkInst {
    double w2l[12];
    double uniform_s, uniform_inv_s;
    double lo[3], hi[3];
    double max_abs;
    uint32_t node_off, tri_off, lds_node_off, lds_tri_off, inst, is_root;
};
#define RT_WF_WALK_TABLE 8u
#define RT_WF_WALK_FIXED_BYTES(threads) ((threads) * RT_WF_LDS_STACK * 4u + 2048u + RT_WF_WALK_TABLE * (uint32_t)sizeof(WalkInst))   // k_wf_walk's LDS apart from the BVH blob: walk stacks, list starts, instance table
RT_DEV WalkInst walk_inst(const DScene& sc, uint32_t m) {
    WalkInst w;
    w.inst = sc.mesh_instances[m];
    const DInstance& in = sc.instances[w.inst];
    const DMesh& mesh = sc.meshes[sc.prims[in.prim].ref];
    for (int k = 0; k < 12; ++k) w.w2l[k] = in.x.w2l.r[k];
    w.uniform_s = in.x.uniform_s; w.uniform_inv_s = in.x.uniform_inv_s;
    for (int k = 0; k < 3; ++k) { w.lo[k] = mesh.lo[k]; w.hi[k] = mesh.hi[k]; }
    w.max_abs = mesh.max_abs;
    w.node_off = mesh.node_off; w.tri_off = mesh.tri_off; w.lds_node_off = mesh.lds_node_off; w.lds_tri_off = mesh.lds_tri_off;
    w.is_root = in.is_root;
    return w;
}
// Every queue is cut into RT_WF_Q sub-queues with a fill counter each.  Appending costs one
// atomic per wave on the counter; on a single counter those serialise (~12 ns each, 4-6 million
// per generation), spread over 256 counters they do not.  Workgroup b reads and appends to
// sub-queue b % RT_WF_Q; generation 0 deals the image out pixel by pixel, so every sub-queue
// holds an even sample of the frame (every 256th pixel) and stays about as long as the others.
#define RT_WF_Q 256u

// Queue entries.  The records of traced queries (TraceJob) are stored field by field (structure of arrays): entry e
// of a queue with `cap` entries keeps field k at f[k*cap + e] (doubles) / u[k*cap + e] (words); the kernels that
// read them densely take a field or two each.  Rays and shadow tests are stored as records, cut into the two
// parts their readers take together:
//   ray         part A, 64 bytes: origin, direction, depth, path, slot, flags   (analytic kernel, walk, shading)
//               part B, 48 bytes: weight, contribution                          (shading; misses)
//   shadow test part A, 64 bytes: origin, direction, distance, slot (RT_NO_SLOT marks a hole)   (analytic kernel, walk)
//               part B, 32 bytes: weight, slot                                  (k_wf_settle, for the tests nothing blocks)
// with all parts A of a queue first, then all parts B.  A lane reads or writes a whole part with 16-byte
// instructions; the shade kernels, which take their hits from lists (every third or fourth entry of a
// sub-queue), then fetch whole 64-byte sectors they use fully instead of a sector per field, and what a lane
// appends fills its sectors at once instead of leaving them half written until the next field's store.
enum { QK_JOB = 0, QK_RAY = 1, QK_SHADOW = 2 };
template <int KIND>
struct WfQueueT {
    double* f;                        // QK_JOB: the doubles' arrays; records: the buffer
    uint32_t* u;                      // QK_JOB: the words' arrays
    static constexpr uint32_t A_BYTES = 64u, B_BYTES = KIND == QK_RAY ? 48u : 32u, A_DOUBLES = KIND == QK_RAY ? 6u : 7u;
    RT_DEV uint8_t* part_a(uint32_t e) const { return reinterpret_cast<uint8_t*>(f) + (size_t)e * A_BYTES; }
    RT_DEV uint8_t* part_b(uint32_t cap, uint32_t e) const { return reinterpret_cast<uint8_t*>(f) + (size_t)cap * A_BYTES + (size_t)e * B_BYTES; }
    RT_DEV double& F(uint32_t cap, uint32_t k, uint32_t e) const {
        if constexpr (KIND == QK_JOB) return f[(size_t)k * cap + e];
        else return k < A_DOUBLES ? reinterpret_cast<double*>(part_a(e))[k] : reinterpret_cast<double*>(part_b(cap, e))[k - A_DOUBLES];
    }
    RT_DEV uint32_t& U(uint32_t cap, uint32_t k, uint32_t e) const {
        if constexpr (KIND == QK_JOB) return u[(size_t)k * cap + e];
        else if constexpr (KIND == QK_RAY) return reinterpret_cast<uint32_t*>(part_a(e) + 48u)[k];
        else return k == 0u ? reinterpret_cast<uint32_t*>(part_a(e) + 56u)[0] : reinterpret_cast<uint32_t*>(part_b(cap, e) + 24u)[0];
    }
    RT_DEV d3 ld3(uint32_t cap, uint32_t k, uint32_t e) const { return mk3(F(cap, k, e), F(cap, k + 1, e), F(cap, k + 2, e)); }
    RT_DEV void st3(uint32_t cap, uint32_t k, uint32_t e, d3 v) const { F(cap, k, e) = v.x; F(cap, k + 1, e) = v.y; F(cap, k + 2, e) = v.z; }
};
typedef WfQueueT<QK_JOB> WfQueue;     // (the records of traced queries)
typedef WfQueueT<QK_RAY> RayQueue;
typedef WfQueueT<QK_SHADOW> ShadowQueue;
#define RT_WF_RAY_BYTES 112u          // per entry of a ray queue (parts A + B)
#define RT_WF_SHADOW_BYTES 96u        // per entry of a shadow-test queue
// rays (the fields of LaneRay): 12 doubles, 4 words
enum { RQ_O = 0, RQ_D = 3, RQ_W = 6, RQ_C = 9, RQ_NF = 12, RQ_DEPTH = 0, RQ_PATH = 1, RQ_SLOT = 2, RQ_FLAGS = 3, RQ_NU = 4 };
// shadow tests (ShadowJob): 10 doubles, the slot twice (SQ_SLOT in part A, SQ_SLOT_B in part B)
enum { SQ_O = 0, SQ_D = 3, SQ_DIST = 6, SQ_W = 7, SQ_NF = 10, SQ_SLOT = 0, SQ_SLOT_B = 1, SQ_NU = 2 };
// query records (TraceJob): 2 doubles, 5 words
enum { JQ_T = 0, JQ_CT = 1, JQ_NF = 2, JQ_INST = 0, JQ_FOUND = 1, JQ_CA = 2, JQ_CB = 3, JQ_FLAGS = 4, JQ_NU = 5 };

struct WfArgs {
    // queries of the current generation; sub-queue s holds rays [s*sub, s*sub + n_in[2s]) and
    // shadow tests [s*sub, s*sub + n_in[2s+1]) of their arrays
    RayQueue rays; ShadowQueue shadows; WfQueue rjobs, sjobs;
    const unsigned int* n_in;
    uint32_t sub;                     // capacity of one sub-queue
    uint32_t cap;                     // RT_WF_Q * sub: entries per array
    // The sub-queues are independent of each other from generation 0 on, so a frame may be run as
    // several chains of launches side by side (one HIP stream each), every chain with a range of
    // sub-queues of its own: this launch works on sub-queues [sq_base, sq_base + sq_count).
    uint32_t sq_base, sq_count;
    // the next generation
    RayQueue out_rays; ShadowQueue out_shadows;
    unsigned int* n_out;
    // mesh walk
    uint32_t* walk_list;              // RT_WF_Q lists of 2*sub query ids: ray i -> i, shadow test i -> i | 0x80000000
    unsigned int* walk_count;         // per list
    unsigned int* walk_next;          // cursor into the lists read as one
    // hits to shade (k_wf_settle -> k_wf_shade): per sub-queue a list of ray entries in the first `sub` words of
    // its walk list (free again after the walk); materials that spawn no rays from the front, the
    // others from the back; hit_count[2s], hit_count[2s+1] (one 64-bit word) count the two kinds
    unsigned int* hit_count;
    // composite shapes: queries whose ray reaches the bounding box of a composite, per sub-queue a list of 2*sub
    // query ids (encoded like the walk list's) - those that reach only the scene's first composite from the front,
    // the others from the back - and csg_count[2s], csg_count[2s+1]; a second run of the analytic kernel answers them
    uint32_t* csg_list;
    unsigned int* csg_count;
    uint32_t* walk_overflow;          // (RT_BVH_STACK - RT_WF_LDS_STACK) words per walk thread
    uint32_t refill_min;
    uint32_t descend_frac;            // eighths: the descent pauses for the leaves once fewer than this share of the walking lanes still descend (0: never)
    // outputs
    unsigned long long* acc;
    uint32_t* acc_flags;
    double* distance;
    uint32_t* hit_id;
    DCounters* counters;
    unsigned int* dropped;            // rays / shadow tests that found no room in the next generation's queues (must stay 0)
    // generation 0
    const uint32_t* rows;
    uint32_t num_rows;
    const uint32_t* list;
    const unsigned int* aa_count;
    uint32_t mode;
    uint32_t gen0_tiles;              // generation 0 dealt out in runs of 64 neighbouring rays (8 x 8 pixel tiles) instead of ray by ray
    uint32_t* tie_list;               // per sub-queue RT_WF_TIE_WORDS words (k_wf_analytic -> k_wf_ties), all counts zero between launches
    uint32_t shade_lists;             // k_wf_shade: the hit lists this launch works on (bit 0: materials without rays, bit 1: the others)
    uint32_t debug;                   // timing experiments only (RT_WF_DEBUG): 1 skips hit reconstruction and shading, 2 shading, 4 the stores of new rays and shadow tests, 8 the shade kernels' own colour terms; 16: the walk kernel's cycles per section instead of the shade kernels'
};

RT_DEV void wave_accumulate(unsigned long long* stage, uint32_t slot, d3 c, unsigned long long* acc, uint32_t* acc_flags);   // (below, with the shading phase)

// The shader tree's rays go straight into the next generation's queues, at positions reserved beforehand.
struct QueueSink {
    RayQueue rays; ShadowQueue shadows;
    uint32_t cap;
    // Reserved entries: one run per lane and queue.  (Laying a wave's reservation out in planes —
    // first new ray of every lane, then the second, ... — makes the stores denser but was measured
    // slower overall: siblings next to each other mix long and short BVH walks within a wave.)
    uint32_t ray0, shadow0;
    uint32_t max_rays, max_shadows, nrays, nshadows;
    RT_DEV uint32_t ray_entry(uint32_t k) const { return ray0 + k; }
    RT_DEV uint32_t shadow_entry(uint32_t k) const { return shadow0 + k; }
    d3 origin;                        // unused: every new ray starts at the hit point it is given
    d3 parent_w;                      // weight of the hit's own ray: new weights are products with it
    uint32_t depth, parent_path, slot;
    bool dry;                         // timing experiments only (RT_WF_DEBUG bit 2): count the entries, store nothing
    bool drop_dead;                   // scene without a sky box: a ray that will fail the gate of Scene::cast_ray adds exactly nothing
    RT_DEV bool room() const { return true; }                      // the reservation covers the material's worst case
    RT_DEV uint32_t mark() const { return nrays | (nshadows << 16); }
    RT_DEV void ray(d3 o, d3 d, d3 w, d3 c, uint32_t ordinal) {
        if (drop_dead && !(depth > 0 && (c.x + c.y + c.z) > RT_MIN_CONTRIBUTION)) return;   // scene.rs:67-73 with a BLACK background
        if (nrays >= max_rays) return;
        const uint32_t e = ray_entry(nrays++);
        if (dry) return;
        rays.st3(cap, RQ_O, e, o); rays.st3(cap, RQ_D, e, d); rays.st3(cap, RQ_W, e, parent_w * w); rays.st3(cap, RQ_C, e, c);
        rays.U(cap, RQ_DEPTH, e) = depth; rays.U(cap, RQ_PATH, e) = child_path(parent_path, ordinal);
        rays.U(cap, RQ_SLOT, e) = slot; rays.U(cap, RQ_FLAGS, e) = 0u;
    }
    RT_DEV void shadow(d3 o, d3 d, double dist, d3 w) {
        if (nshadows >= max_shadows) return;
        const uint32_t e = shadow_entry(nshadows++);
        if (dry) return;
        shadows.st3(cap, SQ_O, e, o); shadows.st3(cap, SQ_D, e, d); shadows.F(cap, SQ_DIST, e) = dist;
        shadows.st3(cap, SQ_W, e, parent_w * w); shadows.U(cap, SQ_SLOT, e) = slot; shadows.U(cap, SQ_SLOT_B, e) = slot;
    }
    template <class K>
    RT_DEV void scale(uint32_t from, uint32_t to, K k) {
        for (uint32_t i = from & 0xFFFFu; i < (to & 0xFFFFu); ++i) rays.st3(cap, RQ_W, ray_entry(i), rays.ld3(cap, RQ_W, ray_entry(i)) * k);
        for (uint32_t i = from >> 16; i < (to >> 16); ++i) shadows.st3(cap, SQ_W, shadow_entry(i), shadows.ld3(cap, SQ_W, shadow_entry(i)) * k);
    }
};

// Takes what QueueSink takes and only counts it: a first run of a hit's material with this sink tells
// how many rays and shadow tests the second, real run will append, so that exactly that much room
// is reserved and the queues stay free of holes.
struct CountSink {
    uint32_t nrays, nshadows, depth;
    bool drop_dead;
    RT_DEV bool room() const { return true; }
    RT_DEV uint32_t mark() const { return nrays | (nshadows << 16); }
    RT_DEV void ray(d3, d3, d3, d3 c, uint32_t) {
        if (drop_dead && !(depth > 0 && (c.x + c.y + c.z) > RT_MIN_CONTRIBUTION)) return;
        nrays++;
    }
    RT_DEV void shadow(d3, d3, double, d3) { nshadows++; }
    template <class K>
    RT_DEV void scale(uint32_t, uint32_t, K) {}
};

template <bool COUNT>
RT_DEV void wf_flush(DCounters* g, const Work<COUNT>& wk, unsigned long long cam, uint32_t mode) {
    if (!COUNT) return;
    if (mode == 0) flush_counters<COUNT>(g, wk, cam, 0);
    else flush_counters<COUNT>(g, wk, 0, cam);
}

// ------------------------------------------------------------------ generation 0
#ifdef RT_FRAME_KERNELS   // (not templates: compiled once, by the file that launches them)
// Pass 1: one camera ray per pixel of the rows this rank renders (src/render.rs:163-170).
// AA resolve: aa_rays jittered camera rays per flagged pixel (src/render.rs:259-299).
__global__ void __launch_bounds__(256)
k_wf_generate(DRenderParams rp, WfArgs a) {
    const uint32_t W = rp.cam.width;
    const double theta = 2.0 * RT_PI / (double)rp.aa_rays;        // render.rs:227
    const d3 cam_o = ld3(rp.cam.origin);
    // (pass 1: the host has checked that the frame's pixels fit the queues.  AA resolve: the number of
    // flagged pixels is only known on the device; samples that do not fit are counted as lost, which
    // makes the host redo the frame with larger queues)
    const unsigned long long aa_want = a.mode == 0 ? 0ull : (unsigned long long)*a.aa_count * rp.aa_rays;
    const unsigned long long aa_room = a.gen0_tiles ? (unsigned long long)RT_WF_Q * (a.sub / 64u) * 64ull : (unsigned long long)a.cap;   // (whole runs of 64 per sub-queue)
    if (a.mode != 0 && aa_want > aa_room && blockIdx.x == 0 && threadIdx.x == 0)
        atomicAdd(a.dropped, (unsigned int)min(aa_want - aa_room, 0x7FFFFFFFull));
    auto emit = [&](uint32_t e, uint32_t item) {
        d3 dir;
        uint32_t slot;
        if (a.mode == 0) {
            uint32_t x = item % W, y = a.rows[item / W];
            slot = y * W + x;
            dir = camera_dir(rp.cam, (double)x + 0.5, (double)y + 0.5);                  // render.rs:167
        } else {
            uint32_t fi = item / rp.aa_rays, i = item % rp.aa_rays;
            uint32_t pixel = a.list[fi];
            uint32_t x = pixel % W, y = pixel / W;
            uint64_t stream0 = rng_stream(rp.seed, pixel, 0);
            double rand_rotation = 0.0 + (2.0 * RT_PI - 0.0) * rng_aa(stream0, 0);       // render.rs:266
            double ax = cos(rand_rotation + theta * (double)i);
            double ay = sin(rand_rotation + theta * (double)i);
            double rand_distance = 0.2 + (0.5 - 0.2) * rng_aa(stream0, 1 + i);           // render.rs:277
            dir = camera_dir(rp.cam, (double)x + 0.5 + ax * rand_distance, (double)y + 0.5 + ay * rand_distance);
            slot = item;
        }
        const RayQueue& r = a.out_rays;
        r.st3(a.cap, RQ_O, e, cam_o); r.st3(a.cap, RQ_D, e, dir);
        r.st3(a.cap, RQ_W, e, mk3(1.0, 1.0, 1.0)); r.st3(a.cap, RQ_C, e, mk3(1.0, 1.0, 1.0));
        r.U(a.cap, RQ_DEPTH, e) = rp.recursion_depth; r.U(a.cap, RQ_PATH, e) = 1u;
        r.U(a.cap, RQ_SLOT, e) = slot; r.U(a.cap, RQ_FLAGS, e) = a.mode == 0 ? 3u : 2u;
    };
    if (a.gen0_tiles) {
        // Coherent dealing: a wave's worth of neighbouring camera rays stays together.  Pass 1: the frame's rows
        // (in the order of `rows`) are cut into tiles of 8 x 8 pixels, tile t goes to sub-queue t % RT_WF_Q as 64
        // consecutive entries; pixels of a tile beyond the frame's edge become holes (flag 8).  AA resolve: runs
        // of 64 consecutive samples (neighbouring flagged pixels) are dealt out the same way.  Rays that start
        // together take the same way through a scene's bounding hierarchy, and so do the shadow tests and rays
        // their hits spawn, which sit side by side in the next generation.
        const uint32_t tw = (W + 7u) / 8u;
        const unsigned long long slots = a.mode == 0 ? (unsigned long long)tw * ((a.num_rows + 7u) / 8u) * 64ull : min(aa_want, aa_room);
        const uint32_t chunks = (uint32_t)((slots + 63ull) / 64ull);
        if (blockIdx.x == 0) {
            for (uint32_t sq = threadIdx.x; sq < RT_WF_Q; sq += blockDim.x) {
                uint32_t n = (chunks / RT_WF_Q + (sq < chunks % RT_WF_Q ? 1u : 0u)) * 64u;
                if (chunks && sq == (chunks - 1u) % RT_WF_Q) n -= (uint32_t)((unsigned long long)chunks * 64ull - slots);   // the last run may be short
                a.n_out[2 * sq] = n; a.n_out[2 * sq + 1] = 0;
            }
        }
        const uint32_t per = ((chunks + RT_WF_Q - 1u) / RT_WF_Q) * 64u;
        for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < per * RT_WF_Q; t += gridDim.x * blockDim.x) {
            const uint32_t entry = t % per, sq = t / per;
            const uint32_t chunk = (entry / 64u) * RT_WF_Q + sq;
            const unsigned long long u = (unsigned long long)chunk * 64ull + entry % 64u;
            if (chunk >= chunks || u >= slots) continue;
            const uint32_t e = sq * a.sub + entry;
            if (a.mode == 0) {
                const uint32_t p = entry % 64u, r = (chunk / tw) * 8u + p / 8u, x = (chunk % tw) * 8u + p % 8u;
                if (r >= a.num_rows || x >= W) { a.out_rays.U(a.cap, RQ_FLAGS, e) = 8u; continue; }
                emit(e, r * W + x);
            } else {
                emit(e, (uint32_t)u);
            }
        }
        return;
    }
    const uint32_t total = a.mode == 0 ? W * a.num_rows : (uint32_t)min(aa_want, (unsigned long long)a.cap);
    // item i goes to sub-queue i % RT_WF_Q
    if (blockIdx.x == 0) {
        for (uint32_t sq = threadIdx.x; sq < RT_WF_Q; sq += blockDim.x) {
            a.n_out[2 * sq] = total / RT_WF_Q + (sq < total % RT_WF_Q ? 1u : 0u); a.n_out[2 * sq + 1] = 0;
        }
    }
    // thread t writes entry t % per of sub-queue t / per: neighbouring lanes fill neighbouring entries
    const uint32_t per = (total + RT_WF_Q - 1u) / RT_WF_Q;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < per * RT_WF_Q; t += gridDim.x * blockDim.x) {
        const uint32_t item = (t % per) * RT_WF_Q + t / per;
        if (item >= total) continue;
        emit((item % RT_WF_Q) * a.sub + item / RT_WF_Q, item);
    }
}

#endif  // RT_FRAME_KERNELS

// ------------------------------------------------------------------ analytic phase
// One thread per queue entry: the sub-queue's rays first, then its shadow tests.
// Scenes with composite shapes run it twice.  REDO = false: every entry, composites left out; a query
// whose ray reaches a composite's bounding box is listed instead of answered.  REDO = true: the listed
// queries only, composites included.  The composite evaluation is by far the heaviest part of a query
// and differs from composite to composite (four spheres, a cube and a sphere, ...): listed side by
// side, queries with the same work fill whole waves, and the first run keeps neither its registers
// nor its scratch memory.
// LDS: [small scene tables][sky-box scenes: per wave 64 x 10 words of staging, per wave 128 list entries]
#define RT_WF_TIE_WORDS (4u + RT_TIE_LIST * 3u)   // a sub-queue's list of noted tie decisions: count, then (query, instance, instance) triples
// (The second run fits three waves per SIMD without spilling - 167 registers against 171 when left alone - and
// gains from the third: subtraction_outside 4K 57.9 -> 52.3 ms together with the empty shade launches left out.  The
// first run held to three waves spills 22 registers and loses: 53.9 ms.)
#define RT_REDO_WAVES(F) ((((F) & (RTF_MESH | RTF_WRAP | RTF_MISC)) == 0u) ? 3 : 1)   // (the variant with every feature would spill 264 registers: left alone)
template <uint32_t F, bool COUNT, bool HOT, bool REDO>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(REDO ? RT_REDO_WAVES(F) : 1)))
k_wf_analytic(DScene gsc, DRenderParams rp, WfArgs a) {
    constexpr bool HAS_MESH = (F & RTF_MESH) != 0;
    constexpr bool HAS_CSG = (F & RTF_CSG) != 0;
    static_assert(HAS_CSG || !REDO, "only scenes with composite shapes have anything to redo");
    const uint32_t sq = a.sq_base + blockIdx.x % a.sq_count;
    const uint32_t nr = min(a.n_in[2 * sq], a.sub), nsh = min(a.n_in[2 * sq + 1], a.sub);
    const uint32_t n0 = REDO ? min(a.csg_count[2 * sq], 2u * a.sub) : 0u, n1 = REDO ? min(a.csg_count[2 * sq + 1], 2u * a.sub - n0) : 0u;
    const uint32_t r0 = (n0 + 63u) & ~63u;                            // the second kind starts in a wave of its own
    const uint32_t nwork = REDO ? r0 + n1 : nr + nsh;
    if ((blockIdx.x / a.sq_count) * blockDim.x >= nwork) return;        // nothing for this workgroup: leave before the tables are copied
    const DScene sc = localise_scene<HOT>(gsc, rt_lds_dynamic);
    const uint32_t cap = a.cap;
    Work<COUNT> wk = {};
    unsigned long long cam = 0;
    const uint32_t* clist = a.csg_list + (size_t)sq * (2u * a.sub);
    // Closest-hit decisions between two surfaces at (all but) one distance are noted in the sub-queue's tie list and
    // decided again, the reference's way, by k_wf_ties (doing that here would cost this kernel its fifth wave per SIMD).
    uint32_t* const tie_count = a.tie_list + (size_t)sq * RT_WF_TIE_WORDS;
    uint32_t* const tie_entries = tie_count + 4;
    // One query (the body of the loop below).  Returns the entry of a ray that failed the gate of Scene::cast_ray in a
    // sky-box scene (what it sees is looked up afterwards), else RT_NO_SLOT.
    auto one_query = [&](uint32_t wi) -> uint32_t {
        uint32_t ql = wi;
        if constexpr (REDO) {
            if (wi >= n0 && wi < r0) return RT_NO_SLOT;
            const uint32_t e = wi < n0 ? clist[wi] : clist[2u * a.sub - 1u - (wi - r0)];
            ql = (e & 0x80000000u) ? nr + (e & 0x7FFFFFFFu) - sq * a.sub : e - sq * a.sub;
        }
        if (!REDO) {
            if (ql < nr) {
                const uint32_t q = sq * a.sub + ql;
                if (a.rays.U(cap, RQ_FLAGS, q) & 8u) { a.rjobs.U(cap, JQ_FLAGS, q) = 6u; return RT_NO_SLOT; }        // a hole (queue overflow only)
            } else {
                const uint32_t q = sq * a.sub + ql - nr;
                if (a.shadows.U(cap, SQ_SLOT, q) == RT_NO_SLOT) { a.sjobs.U(cap, JQ_FLAGS, q) = 7u; return RT_NO_SLOT; }   // a hole
            }
        }
        const bool shadow = ql >= nr;
        const uint32_t q = sq * a.sub + (shadow ? ql - nr : ql);      // entry of the shadow-test / ray arrays
        const WfQueue& jobs = shadow ? a.sjobs : a.rjobs;
        const Work<COUNT> wk_before = wk;                             // (a query handed to the second run is counted there)
        const unsigned long long cam_before = cam;
        d3 o, d;
        double max_distance = 0.0;
        if (shadow) {
            o = a.shadows.ld3(cap, SQ_O, q); d = a.shadows.ld3(cap, SQ_D, q); max_distance = a.shadows.F(cap, SQ_DIST, q);
            RT_COUNT(wk, rays_shadow);
        } else {
            const uint32_t flags = a.rays.U(cap, RQ_FLAGS, q);
            o = a.rays.ld3(cap, RQ_O, q); d = a.rays.ld3(cap, RQ_D, q);
            // gate of Scene::cast_ray (scene.rs:67): terminated rays still see the background
            // (a listed query has passed it already; without a sky box a ray that fails the gate is never queued,
            // QueueSink::ray, and camera rays pass it whenever recursion_depth > 0: its 28 bytes stay unread)
            const bool gated = !REDO && (sc.has_background != 0 || rp.recursion_depth == 0);
            d3 rc = mk3(1.0, 1.0, 1.0);
            if (gated) rc = a.rays.ld3(cap, RQ_C, q);
            if (gated && !(a.rays.U(cap, RQ_DEPTH, q) > 0 && (rc.x + rc.y + rc.z) > RT_MIN_CONTRIBUTION)) {
                // its colour term: the background, looked up below by full waves; exactly nothing without a sky box
                if (flags & 1u) { const uint32_t slot = a.rays.U(cap, RQ_SLOT, q); a.distance[slot] = INFINITY; a.hit_id[slot] = 0xFFFFFFFFu; }
                jobs.U(cap, JQ_FLAGS, q) = 6u;
                return ((F & RTF_IMAGES) != 0 && sc.has_background) ? q : RT_NO_SLOT;
            }
            if (COUNT) { if (flags & 2u) cam++; else wk.rays_secondary++; }
        }
        SceneHit h;
        h.inst = 0; h.c.t = 0.0; h.c.a = 0; h.c.b = 0;
        uint32_t deferred = 0;
        // (triangles, polygons and polyhedra met along an edge are noted as well, by shadow tests too - bit 31 of the
        // query - : k_wf_ties takes the reference's verdict on them)
        constexpr bool DOUBTING = !REDO && (F & RTF_MISC) != 0;
        TieLog ties;                                                  // (a null list: nothing is noted - shadow tests, the second run)
        ties.count = (REDO || (shadow && !DOUBTING)) ? nullptr : tie_count; ties.entries = tie_entries; ties.query = shadow ? (q | 0x80000000u) : q;
        bool found = trace_scene<F, COUNT, HAS_MESH, REDO || !HAS_CSG, TieLog, DOUBTING>(sc, o, d, shadow, max_distance, h, wk, &deferred, &ties);
        if constexpr (HAS_CSG && !REDO) {
            if (deferred && !(shadow && found)) {                     // (an occluded shadow test needs no composite any more)
                wk = wk_before; cam = cam_before;                     // its record is written, and its work counted, by the second run
                uint32_t* list = a.csg_list + (size_t)sq * (2u * a.sub);
                const uint32_t e = shadow ? (q | 0x80000000u) : q;
                if (deferred == 1u) list[wave_alloc(a.csg_count + 2 * sq)] = e;
                else list[2u * a.sub - 1u - wave_alloc(a.csg_count + 2 * sq + 1)] = e;
                return RT_NO_SLOT;
            }
        }
        const bool settled = shadow && found;                        // an occluded shadow test needs no more work
        jobs.U(cap, JQ_FOUND, q) = found ? 1u : 0u;
        jobs.U(cap, JQ_FLAGS, q) = (shadow ? 1u : 0u) | (settled ? 2u : 0u);
        if (!shadow) {                                                // (of a shadow test only the verdict is kept)
            jobs.F(cap, JQ_T, q) = h.t;
            jobs.F(cap, JQ_CT, q) = h.c.t;
            jobs.U(cap, JQ_INST, q) = h.inst;
            jobs.U(cap, JQ_CA, q) = h.c.a; jobs.U(cap, JQ_CB, q) = h.c.b;
        }
        if constexpr (HAS_MESH) {
            // Only queries whose ray reaches the bounding box of some mesh go on to the walk (the walk
            // makes the same test, with the same numbers, first thing for every instance).
            bool reaches = false;
            if (!settled) {
                const uint32_t nm = sc.num_mesh_instances;
                for (uint32_t m = 0; m < nm && !reaches; ++m) {
                    const DInstance& in = sc.instances[sc.mesh_instances[m]];
                    const DMesh& mesh = sc.meshes[sc.prims[in.prim].ref];
                    double s, inv_s, tn;
                    const LRay lr = to_local(in.x, o, d, s, inv_s);
                    const d3 inv = mk3(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
                    reaches = slab(mesh.lo, mesh.hi, lr.o, inv, (shadow ? max_distance : h.t) * s, tn);
                }
                if (COUNT && !reaches) { wk.instance_tests += nm; wk.box_tests += nm; }   // what the walk would have counted
            }
            if (reaches) {
                uint32_t w = wave_alloc(a.walk_count + sq);
                a.walk_list[(size_t)sq * (2u * a.sub) + w] = shadow ? (q | 0x80000000u) : q;
            }
        }
        return RT_NO_SLOT;
    };
    // Sky-box scenes queue the rays that fail the gate: they still see the background (scene.rs:67-73).  That lookup costs
    // hundreds of instructions and concerns a minority of a wave's queries, so those rays are set aside in the wave's own
    // list and looked up 64 at a time by full waves, their terms going straight to the accumulators.
    // LDS behind the tables: [per wave 64 x 10 words of staging][per wave 128 list entries]
    const bool sky = !REDO && (F & RTF_IMAGES) != 0 && sc.has_background != 0;   // (a variant without images has no sky box: none of this is compiled into it)
    uint8_t* sky_lds = rt_lds_dynamic + (HOT ? (gsc.hot_bytes + 15u) & ~15u : 0u);
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(sky_lds) + (threadIdx.x >> 6) * 640u;
    uint32_t* pending = reinterpret_cast<uint32_t*>(sky_lds + 4u * 640u * 8u) + (threadIdx.x >> 6) * 128u;
    uint32_t npending = 0;                                            // (the same on every lane of the wave)
    const uint32_t lane = threadIdx.x & 63u;
    auto look_up_sky = [&](bool all) {
        while (npending >= 64u || (all && npending > 0u)) {
            const uint32_t take = min(npending, 64u);
            uint32_t mslot = RT_NO_SLOT;
            d3 mval = mk3(0.0, 0.0, 0.0);
            if (lane < take) {
                const uint32_t mq = pending[npending - take + lane];
                mslot = a.rays.U(cap, RQ_SLOT, mq);
                mval = a.rays.ld3(cap, RQ_W, mq) * background_color<F>(sc, a.rays.ld3(cap, RQ_D, mq));
            }
            npending -= take;
            wave_accumulate(stage, mslot, mval, a.acc, a.acc_flags);
        }
    };
    const uint32_t bound = sky ? (nwork + 63u) & ~63u : nwork;       // (whole waves stay in the loop when they share a list)
    for (uint32_t wi = (blockIdx.x / a.sq_count) * blockDim.x + threadIdx.x; wi < bound; wi += (gridDim.x / a.sq_count) * blockDim.x) {
        const uint32_t sky_q = wi < nwork ? one_query(wi) : RT_NO_SLOT;
        if (sky) {
            const unsigned long long mm = __ballot(sky_q != RT_NO_SLOT);
            if (mm) {
                if (sky_q != RT_NO_SLOT) pending[npending + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = sky_q;
                npending += (uint32_t)__popcll(mm);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                look_up_sky(false);
            }
        }
    }
    if (sky) look_up_sky(true);
    wf_flush<COUNT>(a.counters, wk, cam, a.mode);
}

// ------------------------------------------------------------------ exact ties
// One 64-thread workgroup per sub-queue, right after the analytic kernel: the decisions it noted (none at all in
// most launches: the workgroup leaves at once).  A query's notes are taken in the order in which they were made, by
// the thread that holds the first of them; its record is rewritten if the answer changes hands.
template <uint32_t F>
__global__ void __launch_bounds__(64)
k_wf_ties(DScene sc, DRenderParams rp, WfArgs a) {
    const uint32_t sq = a.sq_base + blockIdx.x;
    uint32_t* const tie_count = a.tie_list + (size_t)sq * RT_WF_TIE_WORDS;
    const uint32_t* const tie_entries = tie_count + 4;
    const uint32_t nt = min(*tie_count, RT_TIE_LIST);
    if (nt == 0u) return;
    const uint32_t cap = a.cap;
    for (uint32_t e = threadIdx.x; e < nt; e += blockDim.x) {
        const uint32_t q = tie_entries[3 * e];
        bool first = true;
        for (uint32_t k = 0; k < e; ++k) first = first && tie_entries[3 * k] != q;
        if (!first) continue;
        // (i, i): instance i was met along an edge; a query's doubted instances are judged together (resolve_doubts), before its
        // ties.  Left as the loop decided them in scenes with meshes, whose walk lists have been made with the loop's distances by now.
        const bool doubts = (F & RTF_MISC) != 0 && sc.num_mesh_instances == 0u;
        uint32_t doubted[RT_DOUBT_SET];
        uint32_t nd = 0;
        if constexpr ((F & RTF_MISC) != 0) {
            if (doubts) {
                for (uint32_t k = e; k < nt; ++k) {
                    if (tie_entries[3 * k] != q || tie_entries[3 * k + 1] != tie_entries[3 * k + 2]) continue;
                    const uint32_t ti = tie_entries[3 * k + 1];
                    uint32_t at = 0;                              // ascending, each once
                    while (at < nd && doubted[at] < ti) ++at;
                    if ((at < nd && doubted[at] == ti) || nd == RT_DOUBT_SET) continue;
                    for (uint32_t m = nd; m > at; --m) doubted[m] = doubted[m - 1];
                    doubted[at] = ti; ++nd;
                }
            }
        }
        if (q & 0x80000000u) {                                        // a shadow test: the verdict with the noted instances judged by their chains
            if constexpr ((F & RTF_MISC) != 0) {
                if (nd == 0u) continue;
                const uint32_t qs = q & 0x7FFFFFFFu;
                const bool before = a.sjobs.U(cap, JQ_FOUND, qs) != 0u;
                const bool verdict = resolve_shadow_doubts<F>(sc, a.shadows.ld3(cap, SQ_O, qs), a.shadows.ld3(cap, SQ_D, qs), a.shadows.F(cap, SQ_DIST, qs), doubted, nd);
                if (verdict != before) {
                    a.sjobs.U(cap, JQ_FOUND, qs) = verdict ? 1u : 0u;
                    a.sjobs.U(cap, JQ_FLAGS, qs) = 1u | (verdict ? 2u : 0u);
                }
            }
            continue;
        }
        bool found = a.rjobs.U(cap, JQ_FOUND, q) != 0u;
        const d3 o = a.rays.ld3(cap, RQ_O, q), d = a.rays.ld3(cap, RQ_D, q);
        SceneHit h;
        h.t = a.rjobs.F(cap, JQ_T, q); h.inst = a.rjobs.U(cap, JQ_INST, q);
        h.c.t = a.rjobs.F(cap, JQ_CT, q); h.c.a = a.rjobs.U(cap, JQ_CA, q); h.c.b = a.rjobs.U(cap, JQ_CB, q);
        bool changed = false;
        if constexpr ((F & RTF_MISC) != 0) {
            if (nd) { resolve_doubts<F>(sc, o, d, doubted, nd, h, found); changed = true; }
        }
        for (uint32_t k = e; k < nt; ++k) {
            if (tie_entries[3 * k] != q) continue;
            const uint32_t ti = tie_entries[3 * k + 1], tj = tie_entries[3 * k + 2];
            if (ti != tj && found) changed = resolve_tie<F>(sc, o, d, ti, tj, h) || changed;
        }
        if (changed) {
            a.rjobs.U(cap, JQ_FOUND, q) = found ? 1u : 0u;
            a.rjobs.F(cap, JQ_T, q) = h.t; a.rjobs.F(cap, JQ_CT, q) = h.c.t;
            a.rjobs.U(cap, JQ_INST, q) = h.inst; a.rjobs.U(cap, JQ_CA, q) = h.c.a; a.rjobs.U(cap, JQ_CB, q) = h.c.b;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *tie_count = 0u;                            // (the list is the next generation's again)
}

// ------------------------------------------------------------------ mesh walk
// Persistent lanes; a query is walked through every mesh instance by the lane that took it and
// the result merged into its TraceJob: the closest hit over the instances already tested (ties
// go to the earlier instance, src/scene.rs:200-217), or the any-hit verdict.  A lane whose
// query ends waits until refill_min lanes of its wave are free (or nobody is walking), then
// they all take new queries from the list at once.  The walk itself is mesh_walk's
// ("while-while": descend to a leaf, then test its triangles); its stack is a column of LDS per
// thread, the overflow in global memory.
template <bool LDS, bool COUNT>
__global__ void __launch_bounds__(LDS ? 1024 : 256)
k_wf_walk(DScene sc, WfArgs a) {
    uint32_t* lstack = reinterpret_cast<uint32_t*>(rt_lds_dynamic) + threadIdx.x;
    const uint32_t lstride = blockDim.x;
    uint32_t* starts = reinterpret_cast<uint32_t*>(rt_lds_dynamic) + (size_t)blockDim.x * RT_WF_LDS_STACK;   // RT_WF_Q + 1 words, padded to 2 KB
    WalkInst* wtable = reinterpret_cast<WalkInst*>(rt_lds_dynamic + (size_t)blockDim.x * RT_WF_LDS_STACK * 4u + 2048u);
    const uint8_t* blob = rt_lds_dynamic + RT_WF_WALK_FIXED_BYTES(blockDim.x);
    if (threadIdx.x >= 64 && threadIdx.x - 64 < min(sc.num_mesh_instances, RT_WF_WALK_TABLE)) wtable[threadIdx.x - 64] = walk_inst(sc, threadIdx.x - 64);
    if (threadIdx.x < 64) {                                          // exclusive scan of the list lengths by the first wave
        uint32_t run = 0;
        for (uint32_t base = 0; base < a.sq_count; base += 64) {
            uint32_t n = base + threadIdx.x < a.sq_count ? min(a.walk_count[a.sq_base + base + threadIdx.x], 2u * a.sub) : 0u;
            uint32_t incl = n;
            for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)threadIdx.x >= off) incl += v; }
            starts[base + threadIdx.x] = run + incl - n;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) starts[a.sq_count] = run;
    }
    __syncthreads();
    // A short list is walked by few workgroups, in small portions: the others leave before they
    // copy the BVH (late generations and small frames hold a few thousand queries; every wave
    // still gets a portion while there are fewer portions than waves).  The largest portion weighs the wait for the
    // list counter, which a wave sits out once per portion, against the end of the launch, where the waves that took the
    // last portions walk on while the others have run dry (1080p frame with portions of at most 64 / 128 / 256 / 512
    // queries: 50.8 / 48.8 / 49.0 / 50.1 ms).
    const uint32_t total = starts[a.sq_count];
    const uint32_t waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t chunk = min(128u, max(64u, (total / waves + 63u) & ~63u));
    if (blockIdx.x * 4u * chunk >= total) return;                    // one portion per SIMD of a staying workgroup (also: nothing to walk at all)
    if constexpr (LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(sc.lds_blob);
        uint4* dst = reinterpret_cast<uint4*>(const_cast<uint8_t*>(blob));
        for (uint32_t i = threadIdx.x; i < sc.lds_blob_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t nm = sc.num_mesh_instances;
    const uint32_t lane = threadIdx.x & 63u;
    // (the thread's overflow stack in global memory; its address is worked out where it is needed, which is rare, instead of
    // being kept in two registers the kernel does not have)
    auto gstack = [&]() -> uint32_t* {
        uint32_t t = threadIdx.x;
        asm volatile("" : "+v"(t));
        return a.walk_overflow + (size_t)(blockIdx.x * blockDim.x + t) * (RT_BVH_STACK - RT_WF_LDS_STACK);
    };
    Work<COUNT> wk = {};
    // The chain's walk lists are read as one: `starts` holds where each begins in the joint numbering.
    uint32_t chunk_next = 0, chunk_end = 0, chunk_list = 0;          // the wave's reserved part of the joint list, and the list its first entry lies in
    bool dry = total == 0;                                           // wave-uniform: the joint list is used up
    bool have = false, walking = false;
    uint32_t qe = 0;                                                 // the query's entry in its arrays
    bool qshadow = false;
    d3 qo = mk3(0.0, 0.0, 0.0), qd = qo;                             // its ray, world space
    uint32_t m = 0;
    double qt = 0.0, qct = 0.0;
    uint32_t qinst = 0, qfound = 0, qca = 0;
    LRay lr; lr.o = mk3(0.0, 0.0, 0.0); lr.d = lr.o;
    d3 inv = lr.o;
    Slab32 s32 = slab32_setup(lr.o, lr.o, 0.0);                      // (LDS walk: the f32 box test)
    float best32 = 0.0f;
    double best = 0.0, tmax = 0.0, inv_s = 1.0;
    bool any = false, inclusive = false, found = false;
    uint32_t tri = 0, inst = 0, cur = RT_WALK_DONE;
    const DBvhNode* gnodes = sc.bvh_nodes;
    const DMeshTri* gtris = sc.mesh_tris;
    const LBvhNode* lnodes = reinterpret_cast<const LBvhNode*>(blob);
    const LMeshTri* ltris = reinterpret_cast<const LMeshTri*>(blob);
    int sp = 0;
    auto push = [&](uint32_t v) {
        if (sp < RT_WF_LDS_STACK) lstack[sp * lstride] = v;
        else if (sp < RT_BVH_STACK) gstack()[sp - RT_WF_LDS_STACK] = v;
        ++sp;
    };
    // (the LDS part of the stack through a pointer that says so: otherwise the compiler merges the
    // two reads of pop into one generic-pointer load, which is slower and waits on both memory counters)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32* const lstack3 = (lds_u32*)lstack;
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return RT_WALK_DONE;
        --sp;
        if (sp < RT_WF_LDS_STACK) return lstack3[sp * lstride];
        return gstack()[sp - RT_WF_LDS_STACK];
    };
    bool qchanged = false;                                           // a mesh took the query's answer over
    auto finish_query = [&]() {
        if (qchanged) {                                               // (else the analytic phase's record stands)
            const WfQueue& jobs = qshadow ? a.sjobs : a.rjobs;
            jobs.U(a.cap, JQ_FOUND, qe) = 1u;
            if (!qshadow) {                                           // (of a shadow test only the verdict is read)
                jobs.F(a.cap, JQ_T, qe) = qt; jobs.F(a.cap, JQ_CT, qe) = qct;
                jobs.U(a.cap, JQ_INST, qe) = qinst;
                jobs.U(a.cap, JQ_CA, qe) = qca; jobs.U(a.cap, JQ_CB, qe) = 0u;
            }
        }
        have = false;
    };
    const bool sections = COUNT && (a.debug & 16u) != 0;             // (timing experiments: where a wave's cycles go)
    long long t_sec = sections ? clock64() : 0;
    unsigned long long cyc_refill = 0, cyc_descend = 0, cyc_leaf = 0;
    for (;;) {
        const bool need = !walking && (have || !dry);
        const unsigned long long nmask = __ballot(need), wmask = __ballot(walking);
        if (nmask == 0ull && wmask == 0ull) break;
        if (nmask != 0ull && (wmask == 0ull || (uint32_t)__popcll(nmask) >= a.refill_min)) {
            // free lanes take the next queries off the list
            const bool want = need && !have;
            const unsigned long long jm = __ballot(want);
            if (jm) {
                const uint32_t nwant = (uint32_t)__popcll(jm);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(jm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)jm, 0u));   // wanting lanes below this one
                // (wave-uniform) a used-up chunk -> reserve the next portion of the joint list
                if (chunk_next >= chunk_end || chunk_next >= total) {
                    uint32_t base = 0;
                    const uint32_t leader = (uint32_t)(__ffsll((long long)jm) - 1);
                    if (lane == leader) base = atomicAdd(a.walk_next, chunk);
                    base = __shfl(base, leader);
                    chunk_next = base; chunk_end = base + chunk;
                    if (base >= total) dry = true;
                    else {                                            // which list does it start in?  (largest l with starts[l] <= base)
                        uint32_t lo = 0, hi = a.sq_count;
                        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= base) lo = mid; else hi = mid; }
                        chunk_list = lo;
                    }
                }
                const uint32_t avail = dry ? 0u : min(chunk_end, total) - chunk_next;
                const uint32_t idx = chunk_next + rank;
                chunk_next += min(nwant, avail);
                if (want && rank < avail) {
                    // the list joint entry idx belongs to: the chunk's, or one of the next few (starts[sq_count] = total > idx)
                    uint32_t lo = chunk_list;
                    while (starts[lo + 1] <= idx) ++lo;
                    const uint32_t e = a.walk_list[(size_t)(a.sq_base + lo) * (2u * a.sub) + (idx - starts[lo])];
                    qshadow = (e & 0x80000000u) != 0;
                    qe = e & 0x7FFFFFFFu;
                    if (qshadow) {                                    // (a shadow test on this list has found nothing yet)
                        qo = a.shadows.ld3(a.cap, SQ_O, qe); qd = a.shadows.ld3(a.cap, SQ_D, qe);
                        qt = a.shadows.F(a.cap, SQ_DIST, qe); qinst = 0; qfound = 0;
                    } else {
                        qo = a.rays.ld3(a.cap, RQ_O, qe); qd = a.rays.ld3(a.cap, RQ_D, qe);
                        qt = a.rjobs.F(a.cap, JQ_T, qe);
                        qinst = a.rjobs.U(a.cap, JQ_INST, qe); qfound = a.rjobs.U(a.cap, JQ_FOUND, qe);
                    }
                    any = qshadow;
                    have = true; m = 0; qchanged = false;
                }
            }
            if (need && have) {
                RT_PROF(wk, PF_T_REFILLS, PF_T_REFILL_LANES);
                do {   // next mesh instance whose bounding box the ray reaches
                    const WalkInst wi = m < RT_WF_WALK_TABLE ? wtable[m] : walk_inst(sc, m);
                    inst = wi.inst;
                    double s;
                    {   // to_local (rt_trace.h) on the record's copy of the instance transform
                        lr.o = mk3(((wi.w2l[0] * qo.x + wi.w2l[1] * qo.y) + wi.w2l[2] * qo.z) + wi.w2l[3],
                                   ((wi.w2l[4] * qo.x + wi.w2l[5] * qo.y) + wi.w2l[6] * qo.z) + wi.w2l[7],
                                   ((wi.w2l[8] * qo.x + wi.w2l[9] * qo.y) + wi.w2l[10] * qo.z) + wi.w2l[11]);
                        const d3 v = mk3((wi.w2l[0] * qd.x + wi.w2l[1] * qd.y) + wi.w2l[2] * qd.z,
                                         (wi.w2l[4] * qd.x + wi.w2l[5] * qd.y) + wi.w2l[6] * qd.z,
                                         (wi.w2l[8] * qd.x + wi.w2l[9] * qd.y) + wi.w2l[10] * qd.z);
                        if (wi.uniform_inv_s != 0.0) { inv_s = wi.uniform_inv_s; s = wi.uniform_s; }
                        else { s = length(v); inv_s = 1.0 / s; }
                        lr.d = v * inv_s;
                    }
                    RT_COUNT(wk, instance_tests);
                    inv = mk3(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
                    tmax = qt * s;
                    inclusive = (wi.is_root & RT_INST_ROOT) != 0;
                    RT_COUNT(wk, box_tests);
                    double tn;
                    if (slab(wi.lo, wi.hi, lr.o, inv, tmax, tn) && !(sc.has_clear_to_light && qshadow && (wi.is_root & RT_INST_CLEAR_TO_LIGHT))) {
                        walking = true; cur = 0; sp = 0; found = false; best = tmax;
                        if (LDS) { s32 = slab32_setup(lr.o, inv, wi.max_abs); best32 = slab32_limit(best); }
                        gnodes = sc.bvh_nodes + wi.node_off; gtris = sc.mesh_tris + wi.tri_off;
                        lnodes = reinterpret_cast<const LBvhNode*>(blob + wi.lds_node_off);
                        ltris = reinterpret_cast<const LMeshTri*>(blob + wi.lds_tri_off);
                    } else if (++m == nm) {
                        finish_query();
                    }
                } while (!walking && have);
            }
        }
        if (sections) { const long long t1 = clock64(); cyc_refill += (unsigned long long)(t1 - t_sec); t_sec = t1; }
        const uint32_t nwalk = (uint32_t)__popcll(__ballot(walking));
        while (walking && !(cur & 0x80000000u)) {                   // descend to a leaf
            // (lanes still here: those descending) most of the wave waits at its leaves -> test those first
            if ((uint32_t)__popcll(__ballot(true)) * 8u < nwalk * a.descend_frac) break;
            bool h0, h1, swap;
            uint32_t c0, c1;
            if (COUNT) { wk.box_tests += 2; if (LDS) wk.box_tests_f32 += 2; }
            RT_PROF(wk, PF_T_ITERS, PF_T_LANES);
            uint32_t top = 0;
            if (LDS) {
                float tn0, tn1;
                // (the stack's top is asked for together with the node: the branch-free step below may need it, and it
                // does not depend on the box tests)
                top = lstack3[__mul24(min(max(sp - 1, 0), RT_WF_LDS_STACK - 1), (int)lstride)];
                const LBvhNode n = lds_node(lnodes + cur);
                h0 = slab32_box(n.lo[0], n.hi[0], s32, best32, tn0);
                h1 = slab32_box(n.lo[1], n.hi[1], s32, best32, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
                asm volatile("" : "+v"(top));                         // (keeps the read where it was written: ahead of the node's)
            } else {
                double tn0, tn1;
                const DBvhNode& n = gnodes[cur];
                h0 = slab(n.lo[0], n.hi[0], lr.o, inv, best, tn0);
                h1 = slab(n.lo[1], n.hi[1], lr.o, inv, best, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
            }
            if (LDS && !__ballot(sp >= RT_WF_LDS_STACK)) {
                // Every lane's stack top lies in LDS (nearly always): the step without branches.  The far child is
                // stored above the top whether it will be needed or not, the top has been read whether it will be
                // popped or not, and selects decide.  A wave's step is then one straight run of instructions instead of four nested
                // divergent regions, each of which costs a round of scalar bookkeeping per trip.
                lstack3[__mul24(sp, (int)lstride)] = swap ? c0 : c1;
                const bool both = h0 && h1, none = !h0 && !h1;
                const uint32_t one = h0 ? c0 : c1;
                cur = both ? (swap ? c1 : c0) : (none ? (sp > 0 ? top : RT_WALK_DONE) : one);
                sp += both ? 1 : ((none && sp > 0) ? -1 : 0);
            } else if (h0 && h1) {
                cur = swap ? c1 : c0;
                push(swap ? c0 : c1);
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else cur = pop();
        }
        if (sections) { const long long t1 = clock64(); cyc_descend += (unsigned long long)(t1 - t_sec); t_sec = t1; }
        if (walking) {
            if (cur != RT_WALK_DONE && (cur & 0x80000000u)) {         // the leaf's triangles
                RT_PROF(wk, PF_T_LEAF_ITERS, PF_T_LEAF_LANES);
                uint32_t base = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    double t;
                    RT_COUNT(wk, tri_tests);
                    const double lim = any ? tmax : best; const bool incl = any && inclusive;
                    bool hit = LDS ? mesh_tri_test(lds_tri(ltris + base + i), lr, lim, incl, t) : mesh_tri_test(gtris[base + i], lr, lim, incl, t);
                    if (hit) {
                        if (any) {
                            if (inclusive ? t <= tmax : t < tmax) found = true;
                        } else if (t < best) {
                            best = t; tri = base + i; found = true;
                            if (LDS) best32 = slab32_limit(best);
                        }
                    }
                }
                cur = (any && found) ? RT_WALK_DONE : pop();
            }
            if (cur == RT_WALK_DONE) {                                // this instance is done: merge, move on
                bool settled = false;
                if (found) {
                    if (any) { qfound = 1u; settled = true; qchanged = true; }
                    else {
                        double tw = best * inv_s;
                        if (replaces_hit(tw, inst, qt, qinst, qfound != 0u)) {
                            qt = tw; qinst = inst; qfound = 1u; qchanged = true;
                            qct = best; qca = tri;
                        }
                    }
                }
                ++m;
                if (settled || m == nm) finish_query();
                walking = false;
            }
        }
        if (sections) { const long long t1 = clock64(); cyc_leaf += (unsigned long long)(t1 - t_sec); t_sec = t1; }
    }
    if (COUNT) {
        if (sections && (threadIdx.x & 63u) == 0u) { wk.prof[PF_CYC_A] = cyc_refill; wk.prof[PF_CYC_T] = cyc_descend; wk.prof[PF_CYC_B] = cyc_leaf; }
        flush_counters<COUNT>(a.counters, wk, 0, 0);
    }
}

// ------------------------------------------------------------------ shading phase
// Room for `nr` consecutive ray entries and `ns` consecutive shadow-test entries per calling lane;
// both fill counters share one 64-bit word (rays low, shadow tests high) so that one atomic per
// wave serves both queues.  Every lane of the wave must call (with zeros if it wants nothing).
// In two halves, so that work that does not need the answer can run while the atomic is under way.
struct WaveAlloc2 { unsigned long long before, base; };
RT_DEV WaveAlloc2 wave_alloc2_begin(unsigned int* counters, uint32_t nr, uint32_t ns) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long incl = (unsigned long long)nr | ((unsigned long long)ns << 32);
    const unsigned long long mine = incl;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    const unsigned long long sum = __shfl(incl, 63);
    WaveAlloc2 w;
    w.before = incl - mine;
    w.base = 0;
    if (lane == 63u && sum) w.base = atomicAdd(reinterpret_cast<unsigned long long*>(counters), sum);
    return w;
}
RT_DEV void wave_alloc2_end(const WaveAlloc2& w, uint32_t& rbase, uint32_t& sbase) {
    const unsigned long long at = __shfl(w.base, 63) + w.before;
    rbase = (uint32_t)at; sbase = (uint32_t)(at >> 32);
}
RT_DEV void wave_alloc2(unsigned int* counters, uint32_t nr, uint32_t ns, uint32_t& rbase, uint32_t& sbase) {
    wave_alloc2_end(wave_alloc2_begin(counters, nr, ns), rbase, sbase);
}

// Adds one colour term per lane (slot RT_NO_SLOT: none) to the accumulators, the whole wave at once:
// the terms' limbs go through LDS so that nine neighbouring lanes hold the nine limbs of one sample,
// whose atomics then fall into one or two cache lines (four times the rate of one lane per term).
// `stage` is the wave's own 64 x 10 words of LDS.  Every lane of the wave must call.
RT_DEV void wave_accumulate(unsigned long long* stage, uint32_t slot, d3 c, unsigned long long* acc, uint32_t* acc_flags) {
    const uint32_t lane = threadIdx.x & 63u;
    if (!__ballot(slot != RT_NO_SLOT)) return;
    const double v[3] = {c.x, c.y, c.z};
    for (int ch = 0; ch < 3; ++ch) {
        double l0 = 0.0, l1 = 0.0, l2 = 0.0;
        if (slot != RT_NO_SLOT) {
            if (!(fabs(v[ch]) < 0x1p40)) {
                uint32_t bit = (v[ch] != v[ch]) ? 4u : (v[ch] > 0.0 ? 1u : 2u);
                atomicOr(acc_flags + slot, bit << (3 * ch));
            } else {
                l0 = trunc(v[ch]);
                double r1 = (v[ch] - l0) * 0x1p40;
                l1 = trunc(r1);
                l2 = trunc((r1 - l1) * 0x1p40);
            }
        }
        stage[lane * 10u + 3 * ch] = (unsigned long long)(long long)l0;
        stage[lane * 10u + 3 * ch + 1] = (unsigned long long)(long long)l1;
        stage[lane * 10u + 3 * ch + 2] = (unsigned long long)(long long)l2;
    }
    stage[lane * 10u + 9u] = slot;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t t = lane; t < 64u * RT_ACC_LIMBS; t += 64u) {
        const uint32_t rec = t / RT_ACC_LIMBS, limb = t % RT_ACC_LIMBS;
        const uint32_t rslot = (uint32_t)stage[rec * 10u + 9u];
        // neighbouring lanes often hold terms of one sample (the shadow tests of one hit, the rays
        // it spawned): the first of such a run adds the whole run's limbs up (integers: any order
        // gives the same sum) and issues one atomic for them
        if (rslot == RT_NO_SLOT || (rec > 0u && (uint32_t)stage[(rec - 1u) * 10u + 9u] == rslot)) continue;
        unsigned long long l = stage[rec * 10u + limb];
        for (uint32_t r = rec + 1u; r < 64u && (uint32_t)stage[r * 10u + 9u] == rslot; ++r) l += stage[r * 10u + limb];
        if (l != 0ull) atomicAdd(acc + (size_t)rslot * RT_ACC_LIMBS + limb, l);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// First the queries that need no material are settled (k_wf_settle, a light kernel at high
// occupancy), then the hits are shaded (k_wf_shade, register-heavy) from a list in which they sit
// side by side, so that its waves are full and of one kind of material.
//
// k_wf_settle: one thread per queue entry.  Shadow tests that nothing blocks and rays that left
// the scene give their colour term to the accumulators at once (wave_accumulate); hits go onto the
// sub-queue's hit list.  LDS: [per wave 64 x 10 words of staging]
template <uint32_t F, bool COUNT>
__global__ void __launch_bounds__(256)
k_wf_settle(DScene sc, DRenderParams rp, WfArgs a) {
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(rt_lds_dynamic) + (threadIdx.x >> 6) * 640u;
    const uint32_t sq = a.sq_base + blockIdx.x % a.sq_count;
    const uint32_t nr = min(a.n_in[2 * sq], a.sub), nsh = min(a.n_in[2 * sq + 1], a.sub);
    const uint32_t cap = a.cap;
    const uint32_t nq = nr + nsh;
    const uint32_t rounded = (nq + 63u) & ~63u;                      // whole waves stay in the loop: reservation and accumulation are wave operations
    uint32_t* hits = a.walk_list + (size_t)sq * (2u * a.sub);
    Work<COUNT> wk = {};
    // Sky-box scenes: what a ray that left the scene sees costs hundreds of instructions (two inverse
    // trigonometric functions, four texel taps), and such rays are a minority of a wave's queries.  They are
    // set aside in the wave's own list and looked up 64 at a time, by full waves.
    uint32_t* pending = reinterpret_cast<uint32_t*>(rt_lds_dynamic + 4u * 640u * 8u) + (threadIdx.x >> 6) * 128u;
    uint32_t npending = 0;                                            // (the same on every lane of the wave)
    const uint32_t lane = threadIdx.x & 63u;
    const bool batch_misses = (F & RTF_IMAGES) != 0 && sc.has_background != 0;
    auto look_up_sky = [&](bool all) {
        while (npending >= 64u || (all && npending > 0u)) {
            const uint32_t take = min(npending, 64u);
            uint32_t mslot = RT_NO_SLOT;
            d3 mval = mk3(0.0, 0.0, 0.0);
            if (lane < take) {
                const uint32_t mq = pending[npending - take + lane];
                mslot = a.rays.U(cap, RQ_SLOT, mq);
                mval = a.rays.ld3(cap, RQ_W, mq) * background_color<F>(sc, a.rays.ld3(cap, RQ_D, mq));
                if (a.rays.U(cap, RQ_FLAGS, mq) & 1u) { a.distance[mslot] = INFINITY; a.hit_id[mslot] = 0xFFFFFFFFu; }
            }
            npending -= take;
            wave_accumulate(stage, mslot, mval, a.acc, a.acc_flags);
        }
    };
    for (uint32_t ql = (blockIdx.x / a.sq_count) * blockDim.x + threadIdx.x; ql < rounded; ql += (gridDim.x / a.sq_count) * blockDim.x) {
        const uint32_t q = sq * a.sub + (ql >= nr ? ql - nr : ql);   // entry of the shadow-test / ray arrays
        uint32_t kind = 0;
        uint32_t cslot = RT_NO_SLOT;                                  // this query's colour term
        d3 cval = mk3(0.0, 0.0, 0.0);
        bool missed = false;
        if (ql < nq) {
            if (ql >= nr) {                                           // shadow test: the light's term counts if nothing blocks it
                if (!(a.sjobs.U(cap, JQ_FLAGS, q) & 4u) && !a.sjobs.U(cap, JQ_FOUND, q)) {
                    cslot = a.shadows.U(cap, SQ_SLOT_B, q);
                    cval = a.shadows.ld3(cap, SQ_W, q);
                    RT_COUNT(wk, shade_evals);
                }
            } else if (!(a.rjobs.U(cap, JQ_FLAGS, q) & 4u)) {
                if (!a.rjobs.U(cap, JQ_FOUND, q)) {
                    if (batch_misses) missed = true;
                    else {
                        cslot = a.rays.U(cap, RQ_SLOT, q);
                        cval = a.rays.ld3(cap, RQ_W, q) * background_color<F>(sc, a.rays.ld3(cap, RQ_D, q));
                        if (a.rays.U(cap, RQ_FLAGS, q) & 1u) { a.distance[cslot] = INFINITY; a.hit_id[cslot] = 0xFFFFFFFFu; }
                    }
                } else {
                    const uint32_t bounds = sc.shader_bounds[sc.instances[a.rjobs.U(cap, JQ_INST, q)].shader];
                    kind = (bounds & 0xFFFFu) ? 2u : 1u;
                }
            }
        }
        uint32_t at1, at2;
        wave_alloc2(a.hit_count + 2 * sq, kind == 1u ? 1u : 0u, kind == 2u ? 1u : 0u, at1, at2);
        if (kind == 1u) hits[at1] = ql;                               // (hits are rays: ql < nr <= sub, and there are at most nr of them)
        if (kind == 2u) hits[a.sub - 1u - at2] = ql;
        wave_accumulate(stage, cslot, cval, a.acc, a.acc_flags);
        if (batch_misses) {
            const unsigned long long mm = __ballot(missed);
            if (mm) {
                if (missed) pending[npending + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = q;
                npending += (uint32_t)__popcll(mm);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                look_up_sky(false);
            }
        }
    }
    if (batch_misses) look_up_sky(true);
    if (COUNT) flush_counters<COUNT>(a.counters, wk, 0, 0);
}

// k_wf_shade: one thread per hit.  A hit's new rays and shadow tests are written straight into the
// next generation's queues, into room reserved beforehand (one atomic per wave): what the
// material's structure says for materials without rays, what a counting run of the material says
// for the others.  Only when a sub-queue is full do reserved entries stay unused; those are marked
// as holes (ray flags bit 3 / shadow slot RT_NO_SLOT).  The hit's own colour term goes to the
// accumulators at once.
// LDS: [small scene tables][per wave 64 x 10 words of staging]
//
// CLS picks the code for a class of materials; a.shade_lists the hit lists a launch works on (bit 0: materials
// without rays, bit 1: the others).  The general kernel (CLS 0) carries the whole shader interpreter and
// holds 229 registers, two waves per SIMD; most of its time it waits for memory.  When all materials of a
// list are of a simple shape the host launches that list on a kernel built for the shape, which fits three
// waves per SIMD without spilling:
//   CLS 1  every material is a plain PhongShader (the walls of a room): no interpreter, no counting run
//   CLS 2  Reflection / Translucent, alone or in one Composite (mirrors, glass): no Phong code, no frame array
#define RT_SHADE_CLS_KINDS(CLS) ((CLS) == 1 ? (1u << DS_PHONG) : (CLS) == 2 ? ((1u << DS_REFLECTION) | (1u << DS_TRANSLUCENT) | (1u << DS_COMPOSITE)) : RT_SHADER_KINDS_ALL)
#define RT_SHADE_CLS_DEPTH(CLS) ((CLS) == 1 ? 0 : (CLS) == 2 ? 1 : RT_SHADE_MAX_FRAMES)
// (three waves per SIMD are asked for where they are known to fit without spills: the variants without feature bits beyond meshes)
#define RT_SHADE_WAVES(F, CLS) (((CLS) != 0 && ((F) & ~RTF_MESH) == 0u) ? 3 : 2)
template <uint32_t F, bool COUNT, bool HOT, int CLS = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES(F, CLS))))
k_wf_shade(DScene gsc, DRenderParams rp, WfArgs a) {
    constexpr uint32_t KINDS = RT_SHADE_CLS_KINDS(CLS);
    constexpr int DEPTH = RT_SHADE_CLS_DEPTH(CLS);
    const uint32_t sq = a.sq_base + blockIdx.x % a.sq_count;
    const uint32_t cap = a.cap;
    const uint32_t listed1 = min(a.hit_count[2 * sq], a.sub);
    const uint32_t n1 = (a.shade_lists & 1u) ? listed1 : 0u, n2 = (a.shade_lists & 2u) ? min(a.hit_count[2 * sq + 1], a.sub - listed1) : 0u;
    const uint32_t r1 = (n1 + 63u) & ~63u;                            // the second kind starts in a wave of its own
    const uint32_t rounded = (r1 + n2 + 63u) & ~63u;
    if ((blockIdx.x / a.sq_count) * blockDim.x >= rounded) return;      // nothing for this workgroup: leave before the tables are copied
    const DScene sc = localise_scene<HOT>(gsc, rt_lds_dynamic);
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(rt_lds_dynamic + (HOT ? (gsc.hot_bytes + 15u) & ~15u : 0u)) + (threadIdx.x >> 6) * 640u;
    const uint32_t* hits = a.walk_list + (size_t)sq * (2u * a.sub);
    Work<COUNT> wk = {};
    long long t_prof = COUNT ? clock64() : 0;
    RT_PROF_CYC(wk, PF_CYC_FETCH, t_prof);                            // (start-up: the scene tables into LDS)
    for (uint32_t v = (blockIdx.x / a.sq_count) * blockDim.x + threadIdx.x; v < rounded; v += (gridDim.x / a.sq_count) * blockDim.x) {
        const bool shade = v < n1 || (v >= r1 && v - r1 < n2);
        uint32_t want_rays = 0, want_shadows = 0, inst = 0, q = 0;
        // everything the hit needs from the queues is asked for at once, ahead of the reservation,
        // so that the round trips overlap
        d3 ro = mk3(0.0, 0.0, 0.0), rd = ro, rw = ro, rc = ro;
        uint32_t rdepth = 0, rpath = 0, rslot = 0, rflags = 0;
        SceneHit h;
        h.t = 0.0; h.inst = 0; h.c.t = 0.0; h.c.a = 0; h.c.b = 0;
        if (shade) {
            q = sq * a.sub + (v < n1 ? hits[v] : hits[a.sub - 1u - (v - r1)]);
            inst = a.rjobs.U(cap, JQ_INST, q);
            ro = a.rays.ld3(cap, RQ_O, q); rd = a.rays.ld3(cap, RQ_D, q); rw = a.rays.ld3(cap, RQ_W, q); rc = a.rays.ld3(cap, RQ_C, q);
            rdepth = a.rays.U(cap, RQ_DEPTH, q); rpath = a.rays.U(cap, RQ_PATH, q);
            rslot = a.rays.U(cap, RQ_SLOT, q); rflags = a.rays.U(cap, RQ_FLAGS, q);
            h.t = a.rjobs.F(cap, JQ_T, q); h.inst = inst;
            h.c.t = a.rjobs.F(cap, JQ_CT, q); h.c.a = a.rjobs.U(cap, JQ_CA, q); h.c.b = a.rjobs.U(cap, JQ_CB, q);
            const uint32_t bounds = sc.shader_bounds[sc.instances[inst].shader];
            want_rays = bounds & 0xFFFFu; want_shadows = bounds >> 16;
        }
        RT_PROF(wk, PF_TRIPS, PF_TRIP_LANES);
        RT_PROF_CYC(wk, PF_A_ITERS, t_prof);                              // hit list, instance, material bounds
        WorldHit wh;
        wh.p = ro; wh.n = rd; wh.tangent = rd; wh.u = 0.0; wh.v = 0.0; wh.inst = inst;
        if (shade && !(a.debug & 1u)) world_hit<F, COUNT>(sc, ro, rd, h, wh, wk);
        RT_PROF_CYC(wk, PF_B_ITERS, t_prof);                              // the ray, the hit in world space
        // A material that spawns rays is run twice: first only to count what it will append (how
        // many rays depends on the hit: total reflection, contributions below the threshold), ...
        if (CLS != 1 && shade && want_rays && !(a.debug & 3u)) {
            CountSink cs;
            cs.nrays = 0; cs.nshadows = 0; cs.depth = rdepth - 1; cs.drop_dead = sc.has_background == 0;
            Work<false> nowk = {};
            (void)shade_hit<F, false, CountSink, KINDS, DEPTH>(sc, sc.instances[inst].shader, wh, rd, rc, 0ull, rpath, cs, nowk);
            want_rays = min(want_rays, cs.nrays); want_shadows = min(want_shadows, cs.nshadows);
        }
        // ... then, with exactly that much room reserved in the next generation, for real
        uint32_t rbase, sbase;
        wave_alloc2(a.n_out + 2 * sq, want_rays, want_shadows, rbase, sbase);
        RT_PROF_CYC(wk, PF_A_LANES, t_prof);                              // counting run, reservation
        uint32_t cslot = RT_NO_SLOT;
        d3 cval = mk3(0.0, 0.0, 0.0);
        if (shade) {
            if (rbase + want_rays > a.sub) { atomicAdd(a.dropped, want_rays); want_rays = rbase < a.sub ? a.sub - rbase : 0u; }
            if (sbase + want_shadows > a.sub) { atomicAdd(a.dropped, want_shadows); want_shadows = sbase < a.sub ? a.sub - sbase : 0u; }
            QueueSink sink;
            sink.ray0 = sq * a.sub + rbase; sink.shadow0 = sq * a.sub + sbase;
            const DInstance& in = sc.instances[inst];
            if (rflags & 1u) { a.distance[rslot] = h.t; a.hit_id[rslot] = in.node_id; }
            sink.rays = a.out_rays; sink.shadows = a.out_shadows; sink.cap = cap;
            sink.max_rays = want_rays; sink.max_shadows = want_shadows; sink.nrays = 0; sink.nshadows = 0;
            sink.parent_w = rw; sink.depth = rdepth - 1; sink.parent_path = rpath; sink.slot = rslot; sink.drop_dead = sc.has_background == 0; sink.dry = (a.debug & 4u) != 0;
            d3 local = wh.p;
            if (!(a.debug & 3u)) {
                // (a plain Phong material is what the interpreter would call for it at once, with weight 1)
                if constexpr (CLS == 1) local = phong_color<F, COUNT>(sc, sc.shaders[in.shader], wh.p, wh.n, rd, mk3(1.0, 1.0, 1.0), 0ull, rpath, sink, wk);
                else local = shade_hit<F, COUNT, QueueSink, KINDS, DEPTH>(sc, in.shader, wh, rd, rc, 0ull, rpath, sink, wk);
            }
            cslot = rslot;
            cval = rw * local;
            RT_PROF_CYC(wk, PF_B_LANES, t_prof);                          // material
            for (uint32_t i = sink.nrays; i < want_rays; ++i) a.out_rays.U(cap, RQ_FLAGS, sink.ray_entry(i)) = 8u;          // holes
            for (uint32_t i = sink.nshadows; i < want_shadows; ++i) a.out_shadows.U(cap, SQ_SLOT, sink.shadow_entry(i)) = RT_NO_SLOT;
        }
        if (!(a.debug & 16u)) RT_PROF_CYC(wk, PF_CYC_A, t_prof);          // holes
        if (a.debug & 8u) cslot = RT_NO_SLOT;
        wave_accumulate(stage, cslot, cval, a.acc, a.acc_flags);
        if (!(a.debug & 16u)) RT_PROF_CYC(wk, PF_CYC_T, t_prof);          // accumulation
    }
    if (COUNT) flush_counters<COUNT>(a.counters, wk, 0, 0);
}
