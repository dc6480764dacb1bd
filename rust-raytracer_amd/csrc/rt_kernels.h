// rt_kernels.h — the kernels of the render path: the trace kernel (pass 1 and the
// anti-alias resolve), AA detect / combine, and the clamp-gamma-encode stage.
//
// Frame driver they replace: render_with_config (src/render.rs:99-312) with its
// thread pool (src/multithread.rs).
//
// k_trace is a persistent kernel of ray processors.  Every lane keeps a private stack of
// pending rays (in its own block of global memory) and, whenever the stack runs low, takes
// more work from a global counter: a pixel or AA sample (round 0), or a ray another lane
// handed on (later rounds).  The rays on a lane's stack may belong to different samples;
// each carries the slot of its sample, and every colour contribution is added to that
// slot's fixed-point accumulator (integer adds commute, so the image does not depend on
// which lane traced what or in which order; k_acc_resolve turns the sums into f64).
//
// Each trip round the main loop takes a few rays off the stack plus the shadow tests
// the previous trip's hits asked for, and runs them through three phases shared by all 64
// lanes: (A) the analytic primitives, (T) the mesh BVH walk for all of the trip's queries
// in one loop, (B) shading, which pushes the reflected / refracted rays and queues the
// next shadow tests.
//
// Load balancing across launches ("rounds").  The cost of a sample is wildly uneven (a wall
// pixel is 4 rays, a pixel through the glass mesh can be ten thousand), so one launch does
// not run every sample to completion: as soon as the launch's work counter is exhausted
// every lane finishes the trip it is in and writes what is left on its stack to a queue;
// the next launch deals those rays out again over all lanes.  A ray handed on is always
// traced in the next round, so recursion_depth + 1 rounds finish every sample; the last
// round runs to completion regardless.
#pragma once
#include "rt_shade.h"

#define RT_ACC_LIMBS 9    // 3 channels x 3 signed limbs of weight 2^0, 2^-40, 2^-80

// One unique index per calling lane from a shared counter, one atomic per wave.
RT_DEV uint32_t wave_alloc(unsigned int* counter) {
    unsigned long long m = __ballot(1);
    uint32_t lane = threadIdx.x & 63u;
    uint32_t leader = (uint32_t)(__ffsll((long long)m) - 1);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// A lane's pending contributions to one sample: per channel three integer-valued doubles
// (limbs of weight 2^0, 2^-40, 2^-80).  Splitting a colour into limbs is exact for |v| in
// [2^-27, 2^40) (smaller parts are cut at 2^-80); limb sums stay exact in f64 as long as fewer
// than 2^12 terms are pending, so they are flushed to the global 64-bit integer accumulators
// before that.  Non-finite and absurdly large values are recorded as flags instead.
struct LaneAcc {
    uint32_t slot, terms, flags;
    double l[RT_ACC_LIMBS];
};
#define RT_NO_SLOT 0xFFFFFFFFu
RT_DEV void acc_flush(LaneAcc& la, unsigned long long* acc, uint32_t* acc_flags) {
    if (la.slot != RT_NO_SLOT) {
        unsigned long long* limbs = acc + (size_t)la.slot * RT_ACC_LIMBS;
        for (int i = 0; i < RT_ACC_LIMBS; ++i)
            if (la.l[i] != 0.0) atomicAdd(limbs + i, (unsigned long long)(long long)la.l[i]);
        if (la.flags) atomicOr(acc_flags + la.slot, la.flags);
    }
    for (int i = 0; i < RT_ACC_LIMBS; ++i) la.l[i] = 0.0;
    la.terms = 0; la.flags = 0; la.slot = RT_NO_SLOT;
}
RT_DEV void acc_term(LaneAcc& la, int channel, double v) {
    if (!(fabs(v) < 0x1p40)) {
        uint32_t bit = (v != v) ? 4u : (v > 0.0 ? 1u : 2u);
        la.flags |= bit << (3 * channel);
        return;
    }
    double l0 = trunc(v);
    double r1 = (v - l0) * 0x1p40;
    double l1 = trunc(r1);
    double l2 = trunc((r1 - l1) * 0x1p40);
    la.l[3 * channel] += l0; la.l[3 * channel + 1] += l1; la.l[3 * channel + 2] += l2;
}
RT_DEV void acc_contribute(LaneAcc& la, uint32_t slot, d3 c, unsigned long long* acc, uint32_t* acc_flags) {
    if (la.slot != slot || la.terms >= 4000u) { acc_flush(la, acc, acc_flags); la.slot = slot; }
    la.terms++;
    acc_term(la, 0, c.x); acc_term(la, 1, c.y); acc_term(la, 2, c.z);
}
RT_DEV double acc_value(const unsigned long long* limbs, uint32_t flags, uint32_t channel) {
    uint32_t f = (flags >> (3u * channel)) & 7u;
    if (f) {
        if ((f & 4u) || (f & 3u) == 3u) return NAN;
        return (f & 1u) ? INFINITY : -INFINITY;
    }
    const unsigned long long* a = limbs + 3u * channel;
    return ((double)(long long)a[2] * 0x1p-80 + (double)(long long)a[1] * 0x1p-40) + (double)(long long)a[0];
}

// calculate_pixel_location + Ray::from_destination (src/render.rs:127-132,167-168).
// (px, py) already include the caller's +0.5; the closure adds another 0.5.
RT_DEV d3 camera_dir(const DCamera& cam, double px, double py) {
    double cx = (2.0 * ((px + 0.5) / (double)cam.width) - 1.0) * cam.x_factor;
    double cy = (1.0 - 2.0 * (py + 0.5) / (double)cam.height) * cam.fov_factor;
    d3 o = ld3(cam.origin), side = ld3(cam.side), up = ld3(cam.up), view = ld3(cam.view);
    // camera_to_world * (cx, cy, 1, 1): columns side, up, view, origin
    d3 loc = mk3(((side.x * cx + up.x * cy) + view.x * 1.0) + o.x * 1.0,
                 ((side.y * cx + up.y * cy) + view.y * 1.0) + o.y * 1.0,
                 ((side.z * cx + up.z * cy) + view.z * 1.0) + o.z * 1.0);
    return normalize(loc - o);
}

template <bool COUNT>
RT_DEV void flush_counters(DCounters* g, const Work<COUNT>& wk, unsigned long long cam_primary, unsigned long long cam_aa) {
    if (!COUNT) return;
    atomicAdd(&g->rays_primary, cam_primary);
    atomicAdd(&g->rays_aa, cam_aa);
    atomicAdd(&g->rays_secondary, wk.rays_secondary);
    atomicAdd(&g->rays_shadow, wk.rays_shadow);
    atomicAdd(&g->instance_tests, wk.instance_tests);
    atomicAdd(&g->box_tests, wk.box_tests);
    if (wk.box_tests_f32) atomicAdd(&g->box_tests_f32, wk.box_tests_f32);
    if (wk.csg_events) atomicAdd(&g->csg_events, wk.csg_events);
    atomicAdd(&g->tri_tests, wk.tri_tests);
    atomicAdd(&g->shade_evals, wk.shade_evals);
    atomicAdd(&g->stack_overflows, wk.stack_overflows);
    for (int i = 0; i < 16; ++i) if (wk.prof[i]) atomicAdd(&g->prof[i], wk.prof[i]);
}

struct TraceArgs {
    // pass 1 (mode 0): rows[] lists the image rows this rank renders (its own row blocks plus the
    // one-row halo the AA stencil needs); work items are pixels in 8x8 tiles over (width x num_rows)
    const uint32_t* rows;
    uint32_t num_rows;
    double* distance;         // W*H
    uint32_t* hit_id;         // W*H
    // AA resolve (mode 1): one work item per (flagged pixel, extra ray), src/render.rs:259-299
    const uint32_t* list;
    const unsigned int* count;
    unsigned int* next_item;  // global work counter
    DCounters* counters;
    uint32_t mode;
    // rounds: round 0 takes its items from the image / the AA list, round r > 0 from in_items
    uint32_t round;
    uint32_t last_round;      // 1: run every ray to completion, hand nothing on
    unsigned int* idle_count; // lanes of this launch that have run out of work
    uint32_t idle_limit;      // ... and how many of them end the round
    uint32_t batch;           // scene queries per lane per trip (<= RT_BATCH)
    uint32_t refill_min;      // free lanes it takes to interrupt the BVH walk for new queries
    uint32_t wide_sp_limit;   // deepest stack from which a lane may still take several rays in one trip
    uint8_t* lane_mem;        // RT_LANE_BYTES per launched thread
    const LaneRay* in_items;
    const unsigned int* in_count;
    LaneRay* out_items;
    unsigned int* out_count;
    uint32_t queue_cap;
    unsigned long long* acc;  // one accumulator (RT_ACC_LIMBS words) per sample slot, all zero between frames
    uint32_t* acc_flags;
    uint32_t exact_ties;      // noted closest-hit ties are decided again the reference's way (scenes in which two surfaces can meet at one distance)
};

// HOT: the scene's small tables (instances, primitives, shaders, lights ...) are copied into the
// workgroup's LDS first and read from there; only for variants without meshes, whose LDS is free.
template <uint32_t F, bool COUNT, bool HOT = false>
// At least two waves per SIMD, i.e. at most 256 registers: left alone the area-light variants take 448
// and run one wave per SIMD, which was measured 15 % (light_room) to 40 % (monkey_1) slower; three
// waves per SIMD spill too much and are slower again.
__global__ void __launch_bounds__((F & RTF_LDS_BVH) ? 512 : 64) __attribute__((amdgpu_waves_per_eu(2)))
k_trace(DScene gsc, DRenderParams rp, TraceArgs a) {
    static_assert(!HOT || !(F & RTF_MESH), "the mesh variants keep their walk stacks and BVH in LDS");
    const uint32_t queued = a.round > 0 ? min(*a.in_count, a.queue_cap) : 0u;
    if (a.round > 0 && queued == 0) return;                         // nothing was handed on: uniform exit
    const DScene sc = localise_scene<HOT>(gsc, rt_lds_dynamic);
    if constexpr (F & RTF_LDS_BVH) {
        // stage every mesh's BVH nodes and triangles into LDS once per workgroup
        const uint4* src = reinterpret_cast<const uint4*>(sc.lds_blob);
        uint4* dst = reinterpret_cast<uint4*>(const_cast<uint8_t*>(rt_lds_base()));
        for (uint32_t i = threadIdx.x; i < sc.lds_blob_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    constexpr bool BATCHED = (F & RTF_MESH) != 0;    // scenes with meshes: the trip's queries walk the BVH together
    constexpr int KB = RT_BATCH;
    const uint32_t W = rp.cam.width;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tiles_x = (W + 7) / 8;
    uint32_t total;
    if (a.round > 0) total = queued;
    else if (a.mode == 0) total = tiles_x * ((a.num_rows + 7) / 8) * 64;
    else total = *a.count * rp.aa_rays;
    const double theta = 2.0 * RT_PI / (double)rp.aa_rays;        // render.rs:227
    const d3 cam_o = ld3(rp.cam.origin);

    // this lane's working memory
    uint8_t* lane_mem = a.lane_mem + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * RT_LANE_BYTES;
    LaneRay* stack = reinterpret_cast<LaneRay*>(lane_mem + RT_LANE_OFF_STACK);
    TraceJob* jobs = reinterpret_cast<TraceJob*>(lane_mem + RT_LANE_OFF_JOBS);
    ShadowJob* shadows = reinterpret_cast<ShadowJob*>(lane_mem + RT_LANE_OFF_SHADOW);

    Work<COUNT> wk = {};
    unsigned long long cam = 0;
    int sp = 0;                         // pending camera / secondary rays
    uint32_t ns = 0, sh_head = 0;       // pending shadow tests: a ring of RT_MAX_SHADOW_JOBS entries starting at sh_head
    bool exhausted = false;             // this lane asked for work and there was none left
    bool draining = false;              // enough lanes are idle: finish this trip, hand the rest on
    bool idle_counted = false;
    LaneAcc la;
    la.slot = RT_NO_SLOT; la.terms = 0; la.flags = 0;
    for (int i = 0; i < RT_ACC_LIMBS; ++i) la.l[i] = 0.0;
    // the wave's reserved chunk of the work counter (wave-uniform)
    uint32_t chunk_next = 0, chunk_end = 0;

    auto hand_on = [&](const LaneRay& r) -> bool {                 // write a ray to the next round's queue
        uint32_t qi = wave_alloc(a.out_count);
        if (qi >= a.queue_cap) return false;
        a.out_items[qi] = r;
        return true;
    };

    long long t_prof = COUNT ? clock64() : 0;
    for (;;) {
        // ---------------- take work until a.batch rays are pending
        auto take = [&](bool need) {                                // one item for every lane that asks
            unsigned long long needy = __ballot(need);
            if (!needy) return false;
            uint32_t item = 0;
            uint32_t want = (uint32_t)__popcll(needy);
            uint32_t rank = (uint32_t)__popcll(needy & ((1ull << lane) - 1ull));
            uint32_t avail = chunk_end - chunk_next;
            if (avail < want) {
                // hand out the remainder, then reserve a new chunk for the rest
                uint32_t base = 0;
                if (lane == (uint32_t)(__ffsll((long long)needy) - 1)) base = atomicAdd(a.next_item, 64u);
                base = __shfl(base, __ffsll((long long)needy) - 1);
                if (need) item = rank < avail ? chunk_next + rank : base + (rank - avail);
                chunk_next = base + (want - avail);
                chunk_end = base + 64u;
            } else {
                if (need) item = chunk_next + rank;
                chunk_next += want;
            }
            if (!need) return true;
            if (item >= total) { exhausted = true; return true; }
            LaneRay& s0 = stack[sp];
            if (a.round > 0) {
                s0 = a.in_items[item];                                 // a ray handed on by the previous round
                ++sp;
                return true;
            }
            d3 dir;
            uint32_t slot;
            if (a.mode == 0) {
                uint32_t tile = item >> 6, within = item & 63u;
                uint32_t x = (tile % tiles_x) * 8 + (within & 7u), ry = (tile / tiles_x) * 8 + (within >> 3);
                if (x >= W || ry >= a.num_rows) return true;
                uint32_t y = a.rows[ry];
                slot = y * W + x;
                dir = camera_dir(rp.cam, (double)x + 0.5, (double)y + 0.5);              // render.rs:167
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
            st3(s0.o, cam_o); st3(s0.d, dir);
            st3(s0.w, mk3(1.0, 1.0, 1.0)); st3(s0.c, mk3(1.0, 1.0, 1.0));
            s0.depth = rp.recursion_depth; s0.path = 1u; s0.slot = slot; s0.flags = a.mode == 0 ? 3u : 2u;
            ++sp;
            return true;
        };
        if (!draining) {
            for (int f = 0; f < KB; ++f)
                if (!take((uint32_t)sp + ns < a.batch && !exhausted)) break;
            // Once enough lanes of the launch have run out of work, the others stop after the trip
            // they are in and hand their rays on, so that the next round can deal them out evenly.
            if (!a.last_round) {
                const bool now_idle = exhausted && sp == 0 && ns == 0 && !idle_counted;
                const unsigned long long im = __ballot(now_idle);
                if (im && lane == (uint32_t)(__ffsll((long long)im) - 1)) atomicAdd(a.idle_count, (unsigned int)__popcll(im));
                if (now_idle) idle_counted = true;
                if (__hip_atomic_load(a.idle_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.idle_limit) {
                    draining = true;
                    // items this wave had reserved but not yet taken must not get lost (wave-uniform test)
                    if (chunk_next < chunk_end && chunk_next < total && sp < RT_MAX_STACK) take(!exhausted);
                }
            }
        }
        if (__ballot(sp > 0 || ns > 0 || !(exhausted || draining)) == 0ull) break;
        if (sp == 0 && ns == 0) continue;
        RT_PROF(wk, PF_TRIPS, PF_TRIP_LANES);
        RT_PROF_CYC(wk, PF_CYC_FETCH, t_prof);

        // ---------------- one trip: a.batch queries — pending shadow tests first, then rays off the
        // stack (they stay in their stack slots until phase B).  The same number for every lane
        // that has them, so that the lanes of a wave finish the phases together.
        const uint32_t nsh = min(ns, a.batch);
        // (a stack too full for a whole batch of hits to push onto falls back to one ray per trip,
        // plain depth-first order, whose need the host has checked against the stack's size)
        const uint32_t npop = min((uint32_t)sp <= a.wide_sp_limit ? a.batch - nsh : min(a.batch - nsh, 1u), (uint32_t)sp);
        if (draining) {
            // everything below the rays of this trip goes to the next round now
            int keep = 0;
            for (int i = 0; i < sp - (int)npop; ++i) {
                if (!hand_on(stack[i])) { if (keep != i) stack[keep] = stack[i]; ++keep; }
            }
            if (keep != sp - (int)npop) for (uint32_t k = 0; k < npop; ++k) stack[keep + k] = stack[sp - (int)npop + k];
            sp = keep + (int)npop;
        }
        const int sp0 = sp;
        const uint32_t njobs = nsh + npop;
        // phase A: gate, every primitive except the un-wrapped meshes
        uint32_t live_mask = 0;                                     // queries that still need the meshes
        for (uint32_t j = 0; j < njobs; ++j) {
            RT_PROF(wk, PF_A_ITERS, PF_A_LANES);
            TraceJob& q = jobs[j];
            const bool shadow = j < nsh;
            d3 o, d;
            double max_distance = 0.0;
            if (shadow) {
                const uint32_t si = (sh_head + j) % RT_MAX_SHADOW_JOBS;
                const ShadowJob& s = shadows[si];
                o = ld3(s.o); d = ld3(s.d); max_distance = s.dist;
                q.src = RT_LANE_OFF_SHADOW + si * (uint32_t)sizeof(ShadowJob);
                RT_COUNT(wk, rays_shadow);
            } else {
                const uint32_t slot = (uint32_t)sp0 - 1u - (j - nsh);
                const LaneRay& r = stack[slot];
                o = ld3(r.o); d = ld3(r.d);
                const d3 rc = ld3(r.c);
                // gate of Scene::cast_ray (scene.rs:67): terminated rays still see the background
                if (!(r.depth > 0 && (rc.x + rc.y + rc.z) > RT_MIN_CONTRIBUTION)) {
                    acc_contribute(la, r.slot, ld3(r.w) * background_color<F>(sc, d), a.acc, a.acc_flags);
                    if (r.flags & 1u) { a.distance[r.slot] = INFINITY; a.hit_id[r.slot] = 0xFFFFFFFFu; }
                    q.flags = 6u;                                   // resolved here, slot becomes free
                    continue;
                }
                q.src = RT_LANE_OFF_STACK + slot * (uint32_t)sizeof(LaneRay);
                if (COUNT) { if (r.flags & 2u) cam++; else wk.rays_secondary++; }
            }
            SceneHit h;
            h.inst = 0; h.c.t = 0.0; h.c.a = 0; h.c.b = 0;
            // (closest-hit decisions between two surfaces at all but one distance are noted and decided again the
            // reference's way, level by level: exact_tie_order in rt_trace.h)
            LaneTieLog tl;
            tl.count = shadow ? 0u : 1u; tl.i0 = 0; tl.j0 = 0; tl.i1 = 0; tl.j1 = 0;
            bool found = trace_scene<F, COUNT, BATCHED, true, LaneTieLog>(sc, o, d, shadow, max_distance, h, wk, nullptr, &tl);
            if (tl.count > 1u && found && a.exact_ties) {
                (void)resolve_tie<RT_TIE_FEATURES(F)>(sc, o, d, tl.i0, tl.j0, h);
                if (tl.count > 2u) (void)resolve_tie<RT_TIE_FEATURES(F)>(sc, o, d, tl.i1, tl.j1, h);
            }
            q.t = shadow ? max_distance : h.t;
            q.inst = h.inst; q.found = found ? 1u : 0u;
            q.ct = h.c.t; q.ca = h.c.a; q.cb = h.c.b;
            q.flags = (shadow ? 1u : 0u) | ((shadow && found) ? 2u : 0u);   // an occluded shadow test is settled
            if (!(shadow && found)) live_mask |= 1u << j;
        }
        RT_PROF_CYC(wk, PF_CYC_A, t_prof);
        // phase T: the meshes, for all of the trip's queries
        if constexpr (BATCHED) {
            // (lanes without work this trip have skipped it; the others run the walk together)
            mesh_walk_batch<(F & RTF_LDS_BVH) != 0, COUNT>(sc, lane_mem, live_mask, a.refill_min, wk);
        }
        RT_PROF_CYC(wk, PF_CYC_T, t_prof);
        // phase B: add the visible lights, shade the hits, push what they spawn
        for (uint32_t j = 0; j < nsh; ++j) {
            if (!jobs[j].found) {
                const ShadowJob& s = shadows[(sh_head + j) % RT_MAX_SHADOW_JOBS];
                acc_contribute(la, s.slot, ld3(s.w), a.acc, a.acc_flags);
                RT_COUNT(wk, shade_evals);
            }
        }
        sh_head = (sh_head + nsh) % RT_MAX_SHADOW_JOBS;
        ns -= nsh;
        int top = sp0;
        for (uint32_t k = 0; k < npop; ++k) {
            RT_PROF(wk, PF_B_ITERS, PF_B_LANES);
            const int slot = sp0 - 1 - (int)k;
            const TraceJob q = jobs[nsh + k];
            bool reused = false;                                    // the slot holds a new ray
            if (!(q.flags & 4u)) {
                const LaneRay r = stack[slot];
                const d3 rd = ld3(r.d), rw = ld3(r.w);
                if (!q.found) {
                    acc_contribute(la, r.slot, rw * background_color<F>(sc, rd), a.acc, a.acc_flags);
                    if (r.flags & 1u) { a.distance[r.slot] = INFINITY; a.hit_id[r.slot] = 0xFFFFFFFFu; }
                } else {
                    SceneHit h;
                    h.t = q.t; h.inst = q.inst; h.c.t = q.ct; h.c.a = q.ca; h.c.b = q.cb;
                    WorldHit wh;
                    world_hit<F, COUNT>(sc, ld3(r.o), rd, h, wh, wk);
                    const DInstance& in = sc.instances[h.inst];
                    if (r.flags & 1u) { a.distance[r.slot] = h.t; a.hit_id[r.slot] = in.node_id; }
                    uint64_t stream = 0;
                    if constexpr (F & RTF_INLINE_LIGHTS) {          // only the area lights draw random numbers
                        if (a.mode == 0) stream = rng_stream(rp.seed, r.slot, 0);
                        else stream = rng_stream(rp.seed, a.list[r.slot / rp.aa_rays], 1 + r.slot % rp.aa_rays);
                    }
                    Spawn spawn[RT_SHADE_MAX_SPAWN];
                    ArraySink sink;
                    sink.init(spawn, rw, r.depth - 1, sc.has_background == 0);
                    d3 local = shade_hit<F, COUNT>(sc, in.shader, wh, rd, ld3(r.c), stream, r.path, sink, wk);
                    const uint32_t nspawn = sink.n;
                    acc_contribute(la, r.slot, rw * local, a.acc, a.acc_flags);
                    for (uint32_t i = 0; i < nspawn; ++i) {
                        const Spawn& sn = spawn[i];
                        if (sn.shadow) {
                            if (ns >= RT_MAX_SHADOW_JOBS) { RT_COUNT(wk, stack_overflows); continue; }
                            ShadowJob& s = shadows[(sh_head + ns) % RT_MAX_SHADOW_JOBS];
                            ++ns;
                            st3(s.o, sn.o); st3(s.d, sn.d); s.dist = sn.c.x; st3(s.w, sn.w); s.slot = r.slot;
                        } else {
                            // the first new ray takes over its parent's slot, the others go on top
                            LaneRay nr;
                            st3(nr.o, sn.o); st3(nr.d, sn.d); st3(nr.w, sn.w); st3(nr.c, sn.c);
                            nr.depth = r.depth - 1; nr.path = child_path(r.path, sn.ordinal); nr.slot = r.slot; nr.flags = 0u;
                            if (!reused) { stack[slot] = nr; reused = true; }
                            else if (top < RT_MAX_STACK) stack[top++] = nr;
                            else if (a.last_round || !hand_on(nr)) RT_COUNT(wk, stack_overflows);
                        }
                    }
                }
            }
            if (!reused) {                                          // free slot: the top entry moves down into it
                --top;
                if (slot != top) stack[slot] = stack[top];
            }
        }
        sp = top;
        RT_PROF_CYC(wk, PF_CYC_B, t_prof);
        if (draining) {
            // the rays this trip spawned start the next round
            int keep = 0;
            for (int i = 0; i < sp; ++i) {
                if (!hand_on(stack[i])) { if (keep != i) stack[keep] = stack[i]; ++keep; }
            }
            sp = keep;
        }
    }
    acc_flush(la, a.acc, a.acc_flags);
    if (a.mode == 0) flush_counters<COUNT>(a.counters, wk, cam, 0);
    else flush_counters<COUNT>(a.counters, wk, 0, cam);
}

// ------------------------------------------------------------------ accumulators -> f64
// Does this rank render image row y?  Its own row blocks (block b belongs to rank b % nranks), plus - when the AA
// detector will run - the row above and below each, so that the 8-neighbour stencil needs no exchange.
__host__ __device__ inline bool rt_rank_renders_row(uint32_t y, uint32_t H, uint32_t rank, uint32_t nranks, uint32_t tile_rows, bool halo) {
    if (nranks <= 1) return true;
    if ((y / tile_rows) % nranks == rank) return true;
    if (!halo) return false;
    return (y > 0 && ((y - 1) / tile_rows) % nranks == rank) || (y + 1 < H && ((y + 1) / tile_rows) % nranks == rank);
}

#ifdef RT_FRAME_KERNELS   // (the kernels from here on are not templates: compiled once, by the file that launches them)
// The rows this rank renders, ascending, written by the device itself (one workgroup; a frame has a few thousand
// rows): the frame needs no upload and no host synchronisation for them.
__global__ void __launch_bounds__(256)
k_frame_rows(uint32_t H, uint32_t rank, uint32_t nranks, uint32_t tile_rows, uint32_t halo, uint32_t* __restrict__ rows) {
    __shared__ uint32_t wave_total[4];
    uint32_t base = 0;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t y0 = 0; y0 < H; y0 += 256u) {
        const uint32_t y = y0 + threadIdx.x;
        const bool need = y < H && rt_rank_renders_row(y, H, rank, nranks, tile_rows, halo != 0u);
        const unsigned long long m = __ballot(need);
        if (lane == 0) wave_total[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 4u; ++w) { if (w < wave) before += wave_total[w]; all += wave_total[w]; }
        if (need) rows[base + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = y;
        base += all;
        __syncthreads();
    }
}

// One accumulator per sample slot; dst is the colour buffer (pass 1) or the AA sample buffer.
// Leaves every accumulator zero again.
__global__ void __launch_bounds__(256)
k_acc_resolve(unsigned long long* __restrict__ acc, uint32_t* __restrict__ acc_flags, const unsigned int* __restrict__ count,
              uint32_t per_count, uint32_t fixed_n, double* __restrict__ dst) {
    const uint32_t n = count ? *count * per_count : fixed_n;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        unsigned long long* limbs = acc + (size_t)slot * RT_ACC_LIMBS;
        uint32_t f = acc_flags[slot];
        for (uint32_t ch = 0; ch < 3; ++ch) dst[3 * (size_t)slot + ch] = acc_value(limbs, f, ch);
        for (uint32_t i = 0; i < RT_ACC_LIMBS; ++i) limbs[i] = 0ull;
        acc_flags[slot] = 0u;
    }
}

// ------------------------------------------------------------------ AA detect (src/render.rs:202-220)
RT_DEV bool owns_row(const DRenderParams& rp, uint32_t y) { return rp.nranks <= 1 || (y / rp.tile_rows) % rp.nranks == rp.rank; }

__global__ void __launch_bounds__(256)
k_aa_detect(DRenderParams rp, const double* __restrict__ color, const double* __restrict__ distance,
            uint32_t* __restrict__ list, unsigned int* __restrict__ count, uint8_t* __restrict__ mask) {
    const uint32_t W = rp.cam.width, H = rp.cam.height;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = idx % W, y = idx / W;
    bool flag = false;
    if (y < H && x >= 1 && x + 1 < W && y >= 1 && y + 1 < H && owns_row(rp, y)) {
        size_t ci = (size_t)y * W + x;
        double r = color[3 * ci], g = color[3 * ci + 1], b = color[3 * ci + 2];
        if (distance[ci] < INFINITY) {
            for (int dy = -1; dy <= 1 && !flag; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dx == 0 && dy == 0) continue;
                    size_t ni = (size_t)((long long)ci + (long long)dy * W + dx);
                    // Color::diff: |dr| + |db| + |dg| (color.rs:42-46)
                    double diff = (fabs(r - color[3 * ni]) + fabs(b - color[3 * ni + 2])) + fabs(g - color[3 * ni + 1]);
                    if (diff > rp.aa_threshold) { flag = true; break; }
                }
        }
    }
    // wave-aggregated append: one atomic per wave, ranks from the ballot
    unsigned long long ballot = __ballot(flag);
    if (ballot) {
        uint32_t lane = threadIdx.x & 63;
        uint32_t rank = __popcll(ballot & ((1ull << lane) - 1ull));
        uint32_t leader = __ffsll((long long)ballot) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count, (unsigned int)__popcll(ballot));
        base = __shfl(base, leader);
        if (flag) { list[base + rank] = y * W + x; mask[(size_t)y * W + x] = 1; }
    }
}

// mean of the original sample and the extra ones, distance := 0 (src/render.rs:291-297, 306)
__global__ void __launch_bounds__(256)
k_aa_combine(DRenderParams rp, const uint32_t* __restrict__ list, const unsigned int* __restrict__ count,
             const double* __restrict__ samples, double* __restrict__ color, double* __restrict__ distance) {
    uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    for (; f < *count; f += gridDim.x * blockDim.x) {
        uint32_t pixel = list[f];
        d3 total = mk3(0.0, 0.0, 0.0);
        total = total + ld3(color + 3 * (size_t)pixel);
        for (uint32_t i = 0; i < rp.aa_rays; ++i) total = total + ld3(samples + 3 * ((size_t)f * rp.aa_rays + i));
        d3 c = total / (double)(rp.aa_rays + 1);
        color[3 * (size_t)pixel] = c.x; color[3 * (size_t)pixel + 1] = c.y; color[3 * (size_t)pixel + 2] = c.z;
        distance[pixel] = 0.0;
    }
}

// ------------------------------------------------------------------ make_image (src/render.rs:361-370, src/color.rs:50-75)
RT_DEV uint8_t encode_channel(double v) {
    v = fmax(fmin(v, 1.0), 0.0);                                  // clamp: min(1.0) then max(0.0)
    double e = pow(v, 1.0 / 2.2) * 255.0;                         // gamma_encode, then `as u8` truncation
    if (!(e > 0.0)) return 0;
    if (e >= 255.0) return 255;
    return (uint8_t)e;
}
// Writes the rows this rank owns; with a multi-rank split the output is compacted:
// owned block k (k = block / nranks) starts at row k*tile_rows.
__global__ void __launch_bounds__(256)
k_encode(DRenderParams rp, const double* __restrict__ color, uint8_t* __restrict__ rgb) {
    const uint32_t W = rp.cam.width, H = rp.cam.height;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x = idx % W, y = idx / W;
    if (y >= H || !owns_row(rp, y)) return;
    uint32_t out_row = y;
    if (rp.nranks > 1) out_row = ((y / rp.tile_rows) / rp.nranks) * rp.tile_rows + y % rp.tile_rows;
    size_t ci = (size_t)y * W + x, oi = ((size_t)out_row * W + x) * 3;
    rgb[oi] = encode_channel(color[3 * ci]);
    rgb[oi + 1] = encode_channel(color[3 * ci + 1]);
    rgb[oi + 2] = encode_channel(color[3 * ci + 2]);
}
#endif  // RT_FRAME_KERNELS
