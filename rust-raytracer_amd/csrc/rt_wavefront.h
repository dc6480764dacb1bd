// rt_wavefront.h — the render path as a pipeline of small kernels over ray queues in HBM.
//
// Same arithmetic as the persistent trace kernel (rt_kernels.h), other schedule: the rays of a
// frame are processed generation by generation (camera rays, then everything they spawned, ...)
// and every generation runs through four launches
//     k_wf_analytic   one thread per query (ray or shadow test): gate of Scene::cast_ray, every
//                     primitive except the un-wrapped meshes; queries that still need the
//                     meshes go onto the walk list
//     k_wf_walk       persistent lanes take queries off the walk list and walk the mesh BVHs
//                     (LDS-resident), replacing a finished query at once
//     k_wf_shade      one thread per query: hit -> world space, shader tree, new rays and
//                     shadow tests appended to the next generation's queues
//     k_wf_accumulate the colour contributions of the generation into the per-sample
//                     fixed-point accumulators
// Every lane of every kernel has a query of the same kind to work on, each kernel only holds
// the registers its phase needs (more waves per SIMD to hide the queue traffic), and the walk
// draws on millions of queries so that one long walk no longer holds up a wave.
// Used for scenes without area / directional lights (those need visibility answers inside
// the shader and stay on the persistent kernel).
#pragma once
#include "rt_kernels.h"

#define RT_WF_LDS_STACK 12     // BVH walk stack entries per thread kept in LDS by k_wf_walk
// Every queue is cut into RT_WF_Q sub-queues with a fill counter each.  Appending costs one
// atomic per wave on the counter; on a single counter those serialise (~12 ns each, 4-6 million
// per generation), spread over 256 counters they do not.  Workgroup b reads and appends to
// sub-queue b % RT_WF_Q; generation 0 deals the image out pixel by pixel, so every sub-queue
// holds an even sample of the frame (every 256th pixel) and stays about as long as the others.
#define RT_WF_Q 256u

struct alignas(16) WfContribution {   // one colour term for one sample
    uint32_t slot;                    // RT_NO_SLOT: nothing
    uint32_t _pad;
    double c[3];
};

struct WfArgs {
    // queries of the current generation; sub-queue s holds rays [s*sub_rays, s*sub_rays + n_in[2s]) and
    // shadow tests [s*sub_shadows, s*sub_shadows + n_in[2s+1])
    LaneRay* rays;
    TraceJob* rjobs;
    ShadowJob* shadows;
    TraceJob* sjobs;
    const unsigned int* n_in;
    uint32_t sub_rays, sub_shadows;   // capacity of one sub-queue
    uint32_t cap_rays;                // RT_WF_Q * sub_rays
    // the next generation
    LaneRay* out_rays;
    ShadowJob* out_shadows;
    unsigned int* n_out;
    // mesh walk
    uint32_t* walk_list;              // RT_WF_Q lists of (sub_rays + sub_shadows) query ids: ray i -> i, shadow test i -> i | 0x80000000
    unsigned int* walk_count;         // per list
    unsigned int* walk_next;          // per list
    uint32_t* walk_overflow;          // (RT_BVH_STACK - RT_WF_LDS_STACK) words per walk thread
    uint32_t refill_min;
    // outputs
    WfContribution* contrib;          // one slot per query: rays first, shadow tests from cap_rays on
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
};

template <bool COUNT>
RT_DEV void wf_flush(DCounters* g, const Work<COUNT>& wk, unsigned long long cam, uint32_t mode) {
    if (!COUNT) return;
    if (mode == 0) flush_counters<COUNT>(g, wk, cam, 0);
    else flush_counters<COUNT>(g, wk, 0, cam);
}

// ------------------------------------------------------------------ generation 0
// Pass 1: one camera ray per pixel of the rows this rank renders (src/render.rs:163-170).
// AA resolve: aa_rays jittered camera rays per flagged pixel (src/render.rs:259-299).
__global__ void __launch_bounds__(256)
k_wf_generate(DRenderParams rp, WfArgs a) {
    const uint32_t W = rp.cam.width;
    const uint32_t total = a.mode == 0 ? W * a.num_rows : min(*a.aa_count * rp.aa_rays, a.cap_rays / 2);
    // item i goes to sub-queue i % RT_WF_Q
    if (blockIdx.x == 0) {
        for (uint32_t sq = threadIdx.x; sq < RT_WF_Q; sq += blockDim.x) {
            a.n_out[2 * sq] = total / RT_WF_Q + (sq < total % RT_WF_Q ? 1u : 0u); a.n_out[2 * sq + 1] = 0;
        }
    }
    const double theta = 2.0 * RT_PI / (double)rp.aa_rays;        // render.rs:227
    const d3 cam_o = ld3(rp.cam.origin);
    for (uint32_t item = blockIdx.x * blockDim.x + threadIdx.x; item < total; item += gridDim.x * blockDim.x) {
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
        LaneRay r;
        st3(r.o, cam_o); st3(r.d, dir);
        st3(r.w, mk3(1.0, 1.0, 1.0)); st3(r.c, mk3(1.0, 1.0, 1.0));
        r.depth = rp.recursion_depth; r.path = 1u; r.slot = slot; r.flags = a.mode == 0 ? 3u : 2u;
        a.out_rays[(size_t)(item % RT_WF_Q) * a.sub_rays + item / RT_WF_Q] = r;
    }
}

// ------------------------------------------------------------------ analytic phase
template <uint32_t F, bool COUNT>
__global__ void __launch_bounds__(256)
k_wf_analytic(DScene sc, DRenderParams rp, WfArgs a) {
    constexpr bool HAS_MESH = (F & RTF_MESH) != 0;
    const uint32_t sq = blockIdx.x % RT_WF_Q;
    const uint32_t nr = min(a.n_in[2 * sq], a.sub_rays), nsh = min(a.n_in[2 * sq + 1], a.sub_shadows);
    const uint32_t rbase = sq * a.sub_rays, sbase = sq * a.sub_shadows;
    Work<COUNT> wk = {};
    unsigned long long cam = 0;
    for (uint32_t ql = (blockIdx.x / RT_WF_Q) * blockDim.x + threadIdx.x; ql < nr + nsh; ql += (gridDim.x / RT_WF_Q) * blockDim.x) {
        const bool shadow = ql >= nr;
        const uint32_t q = shadow ? sbase + (ql - nr) : rbase + ql;   // index into the shadow-test / ray arrays
        d3 o, d;
        double max_distance = 0.0;
        TraceJob* job;
        if (shadow) {
            const ShadowJob& s = a.shadows[q];
            job = a.sjobs + q;
            a.contrib[a.cap_rays + q].slot = RT_NO_SLOT;
            if (s.slot == RT_NO_SLOT) { job->flags = 7u; continue; }        // a hole
            o = ld3(s.o); d = ld3(s.d); max_distance = s.dist;
            RT_COUNT(wk, rays_shadow);
        } else {
            const LaneRay& r = a.rays[q];
            job = a.rjobs + q;
            if (r.flags & 8u) { a.contrib[q].slot = RT_NO_SLOT; job->flags = 6u; continue; }   // a hole
            o = ld3(r.o); d = ld3(r.d);
            const d3 rc = ld3(r.c);
            // gate of Scene::cast_ray (scene.rs:67): terminated rays still see the background
            if (!(r.depth > 0 && (rc.x + rc.y + rc.z) > RT_MIN_CONTRIBUTION)) {
                WfContribution& c = a.contrib[q];
                c.slot = r.slot;
                st3(c.c, ld3(r.w) * background_color<F>(sc, d));
                if (r.flags & 1u) { a.distance[r.slot] = INFINITY; a.hit_id[r.slot] = 0xFFFFFFFFu; }
                job->flags = 6u;
                continue;
            }
            a.contrib[q].slot = RT_NO_SLOT;
            if (COUNT) { if (r.flags & 2u) cam++; else wk.rays_secondary++; }
        }
        SceneHit h;
        h.inst = 0; h.c.t = 0.0; h.c.a = 0; h.c.b = 0;
        bool found = trace_scene<F, COUNT, HAS_MESH>(sc, o, d, shadow, max_distance, h, wk);
        TraceJob j;
        j.t = shadow ? max_distance : h.t;
        j.inst = h.inst; j.found = found ? 1u : 0u;
        j.ct = h.c.t; j.ca = h.c.a; j.cb = h.c.b;
        j.src = 0;
        const bool settled = shadow && found;                        // an occluded shadow test needs no more work
        j.flags = (shadow ? 1u : 0u) | (settled ? 2u : 0u);
        j._pad[0] = 0; j._pad[1] = 0;
        *job = j;
        if constexpr (HAS_MESH) {
            if (!settled && sc.num_mesh_instances > 0) {
                uint32_t w = wave_alloc(a.walk_count + sq);
                a.walk_list[(size_t)sq * (a.sub_rays + a.sub_shadows) + w] = shadow ? (q | 0x80000000u) : q;
            }
        }
    }
    wf_flush<COUNT>(a.counters, wk, cam, a.mode);
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
    const uint8_t* blob = rt_lds_dynamic + (size_t)blockDim.x * RT_WF_LDS_STACK * 4u + 2048u;
    if (threadIdx.x < 64) {                                          // exclusive scan of the list lengths by the first wave
        uint32_t run = 0;
        for (uint32_t base = 0; base < RT_WF_Q; base += 64) {
            uint32_t n = min(a.walk_count[base + threadIdx.x], a.sub_rays + a.sub_shadows);
            uint32_t incl = n;
            for (int off = 1; off < 64; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)threadIdx.x >= off) incl += v; }
            starts[base + threadIdx.x] = run + incl - n;
            run += __shfl(incl, 63);
        }
        if (threadIdx.x == 0) starts[RT_WF_Q] = run;
    }
    if constexpr (LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(sc.lds_blob);
        uint4* dst = reinterpret_cast<uint4*>(const_cast<uint8_t*>(blob));
        for (uint32_t i = threadIdx.x; i < sc.lds_blob_bytes / 16; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    if (starts[RT_WF_Q] == 0) return;
    const uint32_t nm = sc.num_mesh_instances;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t* gstack = a.walk_overflow + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * (RT_BVH_STACK - RT_WF_LDS_STACK);
    Work<COUNT> wk = {};
    // The RT_WF_Q walk lists are read as one: `starts` holds where each begins in the joint numbering.
    const uint32_t total = starts[RT_WF_Q];
    uint32_t chunk_next = 0, chunk_end = 0;                          // the wave's reserved part of the joint list
    bool dry = total == 0;                                           // wave-uniform: the joint list is used up
    bool have = false, walking = false;
    TraceJob* q = nullptr;
    const double* od = nullptr;
    uint32_t m = 0;
    double qt = 0.0, qct = 0.0;
    uint32_t qinst = 0, qfound = 0, qca = 0;
    LRay lr; lr.o = mk3(0.0, 0.0, 0.0); lr.d = lr.o;
    d3 inv = lr.o;
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
        else if (sp < RT_BVH_STACK) gstack[sp - RT_WF_LDS_STACK] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return RT_WALK_DONE;
        --sp;
        return sp < RT_WF_LDS_STACK ? lstack[sp * lstride] : gstack[sp - RT_WF_LDS_STACK];
    };
    auto finish_query = [&]() {
        q->t = qt; q->inst = qinst; q->found = qfound; q->ct = qct; q->ca = qca; q->cb = 0u;
        have = false;
    };
    for (;;) {
        const bool need = !walking && (have || !dry);
        const unsigned long long nmask = __ballot(need), wmask = __ballot(walking);
        if (nmask == 0ull && wmask == 0ull) break;
        if (nmask != 0ull && (wmask == 0ull || (uint32_t)__popcll(nmask) >= a.refill_min)) {
            // free lanes take the next queries off the list
            const bool want = need && !have;
            const unsigned long long jm = __ballot(want);
            if (jm) {
                const uint32_t nwant = (uint32_t)__popcll(jm), rank = (uint32_t)__popcll(jm & lt);
                // (wave-uniform) a used-up chunk -> reserve the next 256 entries of the joint list
                if (chunk_next >= chunk_end || chunk_next >= total) {
                    uint32_t base = 0;
                    const uint32_t leader = (uint32_t)(__ffsll((long long)jm) - 1);
                    if (lane == leader) base = atomicAdd(a.walk_next, 256u);
                    base = __shfl(base, leader);
                    chunk_next = base; chunk_end = base + 256u;
                    if (base >= total) dry = true;
                }
                const uint32_t avail = dry ? 0u : min(chunk_end, total) - chunk_next;
                const uint32_t idx = chunk_next + rank;
                chunk_next += min(nwant, avail);
                if (want && rank < avail) {
                    // which list does joint entry idx belong to?  (largest l with starts[l] <= idx)
                    uint32_t lo = 0, hi = RT_WF_Q;
                    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= idx) lo = mid; else hi = mid; }
                    const uint32_t e = a.walk_list[(size_t)lo * (a.sub_rays + a.sub_shadows) + (idx - starts[lo])];
                    if (e & 0x80000000u) { q = a.sjobs + (e & 0x7FFFFFFFu); od = a.shadows[e & 0x7FFFFFFFu].o; }
                    else { q = a.rjobs + e; od = a.rays[e].o; }
                    any = (q->flags & 1u) != 0;
                    qt = q->t; qinst = q->inst; qfound = q->found; qct = q->ct; qca = q->ca;
                    have = true; m = 0;
                }
            }
            if (need && have) {
                do {   // next mesh instance whose bounding box the ray reaches
                    inst = sc.mesh_instances[m];
                    const DInstance& in = sc.instances[inst];
                    const DMesh& mesh = sc.meshes[sc.prims[in.prim].ref];
                    double s;
                    lr = to_local(in.x, ld3(od), ld3(od + 3), s, inv_s);
                    RT_COUNT(wk, instance_tests);
                    inv = mk3(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
                    tmax = qt * s;
                    inclusive = in.is_root != 0;
                    RT_COUNT(wk, box_tests);
                    double tn;
                    if (slab(mesh.lo, mesh.hi, lr.o, inv, tmax, tn)) {
                        walking = true; cur = 0; sp = 0; found = false; best = tmax;
                        gnodes = sc.bvh_nodes + mesh.node_off; gtris = sc.mesh_tris + mesh.tri_off;
                        lnodes = reinterpret_cast<const LBvhNode*>(blob + mesh.lds_node_off);
                        ltris = reinterpret_cast<const LMeshTri*>(blob + mesh.lds_tri_off);
                    } else if (++m == nm) {
                        finish_query();
                    }
                } while (!walking && have);
            }
        }
        while (walking && !(cur & 0x80000000u)) {                   // descend to a leaf
            double tn0, tn1;
            bool h0, h1;
            uint32_t c0, c1;
            if (COUNT) wk.box_tests += 2;
            if (LDS) {
                const LBvhNode& n = lnodes[cur];
                h0 = slab_f32(n.lo[0], n.hi[0], lr.o, inv, best, tn0);
                h1 = slab_f32(n.lo[1], n.hi[1], lr.o, inv, best, tn1);
                c0 = n.child[0]; c1 = n.child[1];
            } else {
                const DBvhNode& n = gnodes[cur];
                h0 = slab(n.lo[0], n.hi[0], lr.o, inv, best, tn0);
                h1 = slab(n.lo[1], n.hi[1], lr.o, inv, best, tn1);
                c0 = n.child[0]; c1 = n.child[1];
            }
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                cur = swap ? c1 : c0;
                push(swap ? c0 : c1);
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else cur = pop();
        }
        if (walking) {
            if (cur != RT_WALK_DONE) {                                // the leaf's triangles
                uint32_t base = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    double t;
                    RT_COUNT(wk, tri_tests);
                    bool hit = LDS ? mesh_tri_test(ltris[base + i], lr, t) : mesh_tri_test(gtris[base + i], lr, t);
                    if (hit) {
                        if (any) {
                            if (inclusive ? t <= tmax : t < tmax) found = true;
                        } else if (t < best) {
                            best = t; tri = base + i; found = true;
                        }
                    }
                }
                cur = (any && found) ? RT_WALK_DONE : pop();
            }
            if (cur == RT_WALK_DONE) {                                // this instance is done: merge, move on
                bool settled = false;
                if (found) {
                    if (any) { qfound = 1u; settled = true; }
                    else {
                        double tw = best * inv_s;
                        if (tw < qt || (tw == qt && qfound && inst < qinst)) {
                            qt = tw; qinst = inst; qfound = 1u;
                            qct = best; qca = tri;
                        }
                    }
                }
                ++m;
                if (settled || m == nm) finish_query();
                walking = false;
            }
        }
    }
    if (COUNT) flush_counters<COUNT>(a.counters, wk, 0, 0);
}

// ------------------------------------------------------------------ shading phase
// Room for `n` consecutive entries per calling lane in a queue; one atomic per wave.
RT_DEV uint32_t wave_alloc_n(unsigned int* counter, uint32_t n) {
    // inclusive scan of n over the active lanes
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long active = __ballot(1);
    uint32_t incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    // inactive lanes hand on garbage in the scan: every caller passes n = 0 for lanes without a request instead of leaving the call
    const uint32_t last = 63u - (uint32_t)__builtin_clzll(active);
    const uint32_t sum = __shfl(incl, last);
    uint32_t base = 0;
    if (lane == last && sum) base = atomicAdd(counter, sum);
    base = __shfl(base, last);
    return base + incl - n;
}

// A hit's new rays and shadow tests are written straight into the next generation's queues.  Room
// for the most its material can ask for is reserved first (one atomic per wave and queue); entries
// the hit did not use after all are marked as holes (ray flags bit 3 / shadow slot RT_NO_SLOT) and
// cost the next generation one skipped thread each.
template <uint32_t F, bool COUNT>
__global__ void __launch_bounds__(256)
k_wf_shade(DScene sc, DRenderParams rp, WfArgs a) {
    const uint32_t sq = blockIdx.x % RT_WF_Q;
    const uint32_t nr = min(a.n_in[2 * sq], a.sub_rays), nsh = min(a.n_in[2 * sq + 1], a.sub_shadows);
    const uint32_t rb = sq * a.sub_rays, sb = sq * a.sub_shadows;
    const uint32_t nq = nr + nsh;
    const uint32_t rounded = (nq + 63u) & ~63u;                      // whole waves stay in the loop: the queue allocation is a wave operation
    Work<COUNT> wk = {};
    for (uint32_t ql = (blockIdx.x / RT_WF_Q) * blockDim.x + threadIdx.x; ql < rounded; ql += (gridDim.x / RT_WF_Q) * blockDim.x) {
        const uint32_t q = ql >= nr ? sb + (ql - nr) : rb + ql;      // index into the shadow-test / ray arrays
        bool shade = false;
        TraceJob j;
        uint32_t want_rays = 0, want_shadows = 0;
        if (ql < nq) {
            if (ql >= nr) {                                           // shadow test: the light's term counts if nothing blocks it
                if (!(a.sjobs[q].flags & 4u) && !a.sjobs[q].found) {
                    WfContribution& c = a.contrib[a.cap_rays + q];
                    c.slot = a.shadows[q].slot;
                    st3(c.c, ld3(a.shadows[q].w));
                    RT_COUNT(wk, shade_evals);
                }
            } else {
                j = a.rjobs[q];
                if (!(j.flags & 4u)) {
                    if (!j.found) {
                        const LaneRay& r = a.rays[q];
                        WfContribution& c = a.contrib[q];
                        c.slot = r.slot;
                        st3(c.c, ld3(r.w) * background_color<F>(sc, ld3(r.d)));
                        if (r.flags & 1u) { a.distance[r.slot] = INFINITY; a.hit_id[r.slot] = 0xFFFFFFFFu; }
                    } else {
                        shade = true;
                        const uint32_t bounds = sc.shader_bounds[sc.instances[j.inst].shader];
                        want_rays = bounds & 0xFFFFu; want_shadows = bounds >> 16;
                    }
                }
            }
        }
        // room in the next generation for what the hit may ask for
        uint32_t rbase = wave_alloc_n(a.n_out + 2 * sq, want_rays);
        uint32_t sbase = wave_alloc_n(a.n_out + 2 * sq + 1, want_shadows);
        if (!shade) continue;
        if (rbase + want_rays > a.sub_rays) { atomicAdd(a.dropped, want_rays); want_rays = rbase < a.sub_rays ? a.sub_rays - rbase : 0u; }
        if (sbase + want_shadows > a.sub_shadows) { atomicAdd(a.dropped, want_shadows); want_shadows = sbase < a.sub_shadows ? a.sub_shadows - sbase : 0u; }
        const LaneRay r = a.rays[q];
        const d3 rd = ld3(r.d), rw = ld3(r.w);
        SceneHit h;
        h.t = j.t; h.inst = j.inst; h.c.t = j.ct; h.c.a = j.ca; h.c.b = j.cb;
        WorldHit wh;
        world_hit<F, COUNT>(sc, ld3(r.o), rd, h, wh, wk);
        const DInstance& in = sc.instances[h.inst];
        if (r.flags & 1u) { a.distance[r.slot] = h.t; a.hit_id[r.slot] = in.node_id; }
        QueueSink sink;
        sink.rays = a.out_rays + rb + rbase; sink.shadows = a.out_shadows + sb + sbase;
        sink.max_rays = want_rays; sink.max_shadows = want_shadows; sink.nrays = 0; sink.nshadows = 0;
        sink.parent_w = rw; sink.depth = r.depth - 1; sink.parent_path = r.path; sink.slot = r.slot;
        d3 local = shade_hit<F, COUNT>(sc, in.shader, wh, rd, ld3(r.c), 0ull, r.path, sink, wk);
        WfContribution& c = a.contrib[q];
        c.slot = r.slot;
        st3(c.c, rw * local);
        for (uint32_t i = sink.nrays; i < want_rays; ++i) sink.rays[i].flags = 8u;          // holes
        for (uint32_t i = sink.nshadows; i < want_shadows; ++i) sink.shadows[i].slot = RT_NO_SLOT;
    }
    if (COUNT) flush_counters<COUNT>(a.counters, wk, 0, 0);
}

// ------------------------------------------------------------------ accumulation
// One thread per (contribution, limb): the nine limbs of one sample are adjacent words, so the
// atomics of nine neighbouring lanes fall into one or two cache lines.
__global__ void __launch_bounds__(256)
k_wf_accumulate(WfArgs a) {
    const uint32_t sq = blockIdx.x % RT_WF_Q;
    const uint32_t nr = min(a.n_in[2 * sq], a.sub_rays), nsh = min(a.n_in[2 * sq + 1], a.sub_shadows);
    const uint32_t tasks = (nr + nsh) * RT_ACC_LIMBS;
    for (uint32_t t = (blockIdx.x / RT_WF_Q) * blockDim.x + threadIdx.x; t < tasks; t += (gridDim.x / RT_WF_Q) * blockDim.x) {
        const uint32_t ql = t / RT_ACC_LIMBS, limb = t % RT_ACC_LIMBS;
        const WfContribution& c = a.contrib[ql < nr ? (size_t)sq * a.sub_rays + ql : (size_t)a.cap_rays + (size_t)sq * a.sub_shadows + (ql - nr)];
        const uint32_t slot = c.slot;
        if (slot == RT_NO_SLOT) continue;
        const uint32_t channel = limb / 3u, part = limb % 3u;
        const double v = c.c[channel];
        if (!(fabs(v) < 0x1p40)) {
            if (part == 0) {
                uint32_t bit = (v != v) ? 4u : (v > 0.0 ? 1u : 2u);
                atomicOr(a.acc_flags + slot, bit << (3u * channel));
            }
            continue;
        }
        double l0 = trunc(v);
        double r1 = (v - l0) * 0x1p40;
        double l1 = trunc(r1);
        double l2 = trunc((r1 - l1) * 0x1p40);
        double l = part == 0 ? l0 : (part == 1 ? l1 : l2);
        if (l != 0.0) atomicAdd(a.acc + (size_t)slot * RT_ACC_LIMBS + limb, (unsigned long long)(long long)l);
    }
}
