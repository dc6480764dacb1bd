// rt_kernels.h — the kernels of the render path: the trace kernel (pass 1 and the
// anti-alias resolve), AA detect / combine, and the clamp-gamma-encode stage.
//
// Frame driver they replace: render_with_config (src/render.rs:99-312) with its
// thread pool (src/multithread.rs).
//
// k_trace is a persistent kernel.  Every lane owns one camera sample at a time and a
// private stack of pending rays (in its own block of global memory); each trip round the main loop pops a few entries and runs
// the scene queries for them — closest hit for camera / reflected / refracted rays, any-hit
// for the deferred point-light shadow tests — through the same code, so all 64 lanes of a
// wave sit in the same traversal loop whatever kind of ray they carry.  A lane whose stack runs empty writes
// its sample and immediately takes the next one from the wave's chunk of the global work
// counter: cheap wall pixels (4 rays) and expensive glass pixels (hundreds) no longer
// wait for each other inside a wave, which is what the reference's 500 dynamic row-chunks
// achieve for CPU threads.
#pragma once
#include "rt_shade.h"


// Load balancing across launches.  The cost of a camera sample is wildly uneven (a wall pixel
// is 4 rays, a pixel through the glass mesh can be ten thousand, all serial in one lane), so a
// launch gives every work item a budget of ray steps.  Once the budget is used up the lane stops
// tracing camera/secondary rays of that item: each one still pending on its stack is written to
// a queue in global memory and becomes a work item of its own in the next launch ("round").
// Every part of a split sample adds its share of the colour into a fixed-point accumulator
// (integer adds commute, so the sum does not depend on the order in which lanes finish);
// k_acc_resolve turns the accumulators back into f64.
struct SpillItem {        // 112 bytes
    double o[3], d[3], w[3], c[3];
    uint32_t depth, path;
    uint32_t out;         // output slot: pixel index (pass 1) or sample index (AA resolve)
    uint32_t acc_slot;
};
#define RT_NO_SLOT 0xFFFFFFFFu
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

// Adds v into three signed 64-bit limbs (exact for |v| in [2^-27, 2^40); smaller parts are cut
// at 2^-80).  Non-finite and absurdly large values are recorded as flags instead.
RT_DEV void acc_add(unsigned long long* limbs, uint32_t* flags, uint32_t channel, double v) {
    if (!(fabs(v) < 0x1p40)) {
        uint32_t bit = (v != v) ? 4u : (v > 0.0 ? 1u : 2u);
        atomicOr(flags, bit << (3u * channel));
        return;
    }
    double l0 = trunc(v);
    double r1 = (v - l0) * 0x1p40;
    double l1 = trunc(r1);
    double l2 = trunc((r1 - l1) * 0x1p40);
    unsigned long long* a = limbs + 3u * channel;
    if (l0 != 0.0) atomicAdd(a, (unsigned long long)(long long)l0);
    if (l1 != 0.0) atomicAdd(a + 1, (unsigned long long)(long long)l1);
    if (l2 != 0.0) atomicAdd(a + 2, (unsigned long long)(long long)l2);
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
    atomicAdd(&g->tri_tests, wk.tri_tests);
    atomicAdd(&g->shade_evals, wk.shade_evals);
    atomicAdd(&g->stack_overflows, wk.stack_overflows);
}

struct TraceArgs {
    // pass 1 (mode 0): rows[] lists the image rows this rank renders (its own row blocks plus the
    // one-row halo the AA stencil needs); work items are pixels in 8x8 tiles over (width x num_rows)
    const uint32_t* rows;
    uint32_t num_rows;
    double* color;            // W*H*3
    double* distance;         // W*H
    uint32_t* hit_id;         // W*H
    // AA resolve (mode 1): one work item per (flagged pixel, extra ray), src/render.rs:259-299
    const uint32_t* list;
    const unsigned int* count;
    double* samples;          // count*aa_rays*3
    unsigned int* next_item;  // global work counter
    DCounters* counters;
    uint32_t mode;
    // rounds: round 0 takes its items from the image / the AA list, round r > 0 from in_items
    uint32_t round;
    uint32_t batch;           // rays a lane takes off its stack per trip (<= RT_BATCH; 1 without meshes)
    uint8_t* lane_mem;        // RT_LANE_BYTES per launched thread
    uint32_t budget;          // ray steps per item before its pending rays are handed to the next round
    const SpillItem* in_items;
    const unsigned int* in_count;
    SpillItem* out_items;
    unsigned int* out_count;
    uint32_t queue_cap;
    unsigned long long* acc;  // acc_cap x RT_ACC_LIMBS, all zero between frames
    uint32_t* acc_flags;      // acc_cap
    uint32_t* acc_out;        // acc_cap: output slot of each accumulator
    unsigned int* acc_alloc;
    uint32_t acc_cap;
};

template <uint32_t F, bool COUNT>
__global__ void __launch_bounds__((F & RTF_LDS_BVH) ? 512 : 64)
k_trace(DScene sc, DRenderParams rp, TraceArgs a) {
    const uint32_t queued = a.round > 0 ? min(*a.in_count, a.queue_cap) : 0u;
    if (a.round > 0 && queued == 0) return;                         // nothing was handed on: uniform exit
    if constexpr (F & RTF_LDS_BVH) {
        // stage every mesh's BVH nodes and triangles into LDS once per workgroup
        const uint4* src = reinterpret_cast<const uint4*>(sc.lds_blob);
        uint4* dst = reinterpret_cast<uint4*>(const_cast<uint8_t*>(rt_lds_base()));
        for (uint32_t i = threadIdx.x; i < sc.lds_blob_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    constexpr bool BATCHED = (F & RTF_MESH) != 0;    // scenes with meshes: several rays per trip, walked together
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
    uint32_t ns = 0;                    // pending shadow tests (spawned by the previous trip)
    // the sample this lane is working on
    bool active = false, exhausted = false, first = true;
    uint32_t item = 0;
    uint32_t steps = 0;                 // ray steps spent on the current item
    uint32_t acc_slot = RT_NO_SLOT;     // accumulator of the current item's sample once it has been split
    size_t out_index = 0;
    uint64_t stream = 0;
    d3 color = mk3(0.0, 0.0, 0.0);
    double distance = INFINITY;
    uint32_t hit_id = 0xFFFFFFFFu;
    // the wave's reserved chunk of the work counter (wave-uniform)
    uint32_t chunk_next = 0, chunk_end = 0;

    for (;;) {
        // ---------------- finish / refill
        if (active && sp == 0 && ns == 0) {
            if (acc_slot != RT_NO_SLOT) {
                unsigned long long* limbs = a.acc + (size_t)acc_slot * RT_ACC_LIMBS;
                acc_add(limbs, a.acc_flags + acc_slot, 0, color.x);
                acc_add(limbs, a.acc_flags + acc_slot, 1, color.y);
                acc_add(limbs, a.acc_flags + acc_slot, 2, color.z);
            } else if (a.mode == 0) {
                a.color[3 * out_index] = color.x; a.color[3 * out_index + 1] = color.y; a.color[3 * out_index + 2] = color.z;
            } else {
                a.samples[3 * out_index] = color.x; a.samples[3 * out_index + 1] = color.y; a.samples[3 * out_index + 2] = color.z;
            }
            if (a.mode == 0 && a.round == 0) {
                a.distance[out_index] = distance;
                a.hit_id[out_index] = hit_id;
            }
            active = false;
        }
        bool need = !active && !exhausted;
        unsigned long long needy = __ballot(need);
        if (needy) {
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
            if (need) {
                if (item >= total) {
                    exhausted = true;
                } else {
                    // ---- set up the work item
                    d3 dir;
                    bool valid = true;
                    steps = 0;
                    if (a.round > 0) {
                        // a ray handed on by the previous round: a work item of its own
                        const SpillItem& it = a.in_items[item];
                        out_index = it.out;
                        acc_slot = it.acc_slot;
                        if (a.mode == 0) stream = rng_stream(rp.seed, out_index, 0);
                        else stream = rng_stream(rp.seed, a.list[it.out / rp.aa_rays], 1 + it.out % rp.aa_rays);
                        active = true; first = false; valid = false;
                        color = mk3(0.0, 0.0, 0.0);
                        LaneRay& s0 = stack[0];
                        for (int k = 0; k < 3; ++k) { s0.o[k] = it.o[k]; s0.d[k] = it.d[k]; s0.w[k] = it.w[k]; s0.c[k] = it.c[k]; }
                        s0.depth = it.depth; s0.path = it.path;
                        sp = 1; ns = 0;
                    } else if (a.mode == 0) {
                        uint32_t tile = item >> 6, within = item & 63u;
                        uint32_t x = (tile % tiles_x) * 8 + (within & 7u), ry = (tile / tiles_x) * 8 + (within >> 3);
                        if (x >= W || ry >= a.num_rows) valid = false;
                        else {
                            uint32_t y = a.rows[ry];
                            out_index = (size_t)y * W + x;
                            dir = camera_dir(rp.cam, (double)x + 0.5, (double)y + 0.5);      // render.rs:167
                            stream = rng_stream(rp.seed, out_index, 0);
                        }
                    } else {
                        uint32_t f = item / rp.aa_rays, i = item % rp.aa_rays;
                        uint32_t pixel = a.list[f];
                        uint32_t x = pixel % W, y = pixel / W;
                        uint64_t stream0 = rng_stream(rp.seed, pixel, 0);
                        double rand_rotation = 0.0 + (2.0 * RT_PI - 0.0) * rng_aa(stream0, 0);   // render.rs:266
                        double ax = cos(rand_rotation + theta * (double)i);
                        double ay = sin(rand_rotation + theta * (double)i);
                        double rand_distance = 0.2 + (0.5 - 0.2) * rng_aa(stream0, 1 + i);       // render.rs:277
                        dir = camera_dir(rp.cam, (double)x + 0.5 + ax * rand_distance, (double)y + 0.5 + ay * rand_distance);
                        stream = rng_stream(rp.seed, pixel, 1 + i);
                        out_index = item;
                    }
                    if (valid) {
                        active = true; first = true; acc_slot = RT_NO_SLOT;
                        color = mk3(0.0, 0.0, 0.0); distance = INFINITY; hit_id = 0xFFFFFFFFu;
                        LaneRay& s0 = stack[0];
                        st3(s0.o, cam_o); st3(s0.d, dir);
                        st3(s0.w, mk3(1.0, 1.0, 1.0)); st3(s0.c, mk3(1.0, 1.0, 1.0));
                        s0.depth = rp.recursion_depth; s0.path = 1u;
                        sp = 1; ns = 0;
                    }
                }
            }
        }
        if (__ballot(active || !exhausted) == 0ull) break;
        if (!active) continue;

        // ---------------- one trip: the shadow tests spawned by the previous trip plus up to
        // a.batch rays off the stack (they stay in their stack slots until phase B)
        const bool primary_trip = first;
        first = false;
        const int sp0 = sp;
        const uint32_t npop = min(a.batch, (uint32_t)sp);
        const uint32_t njobs = ns + npop;
        // phase A: gate, budget, every primitive except the un-wrapped meshes
        for (uint32_t j = 0; j < njobs; ++j) {
            TraceJob& q = jobs[j];
            const bool shadow = j < ns;
            d3 o, d;
            double max_distance = 0.0;
            if (shadow) {
                const ShadowJob& s = shadows[j];
                o = ld3(s.o); d = ld3(s.d); max_distance = s.dist;
                q.src = RT_LANE_OFF_SHADOW + j * (uint32_t)sizeof(ShadowJob);
                RT_COUNT(wk, rays_shadow);
            } else {
                const uint32_t slot = (uint32_t)sp0 - 1u - (j - ns);
                const LaneRay& r = stack[slot];
                o = ld3(r.o); d = ld3(r.d);
                const d3 rc = ld3(r.c);
                // gate of Scene::cast_ray (scene.rs:67): terminated rays still see the background
                if (!(r.depth > 0 && (rc.x + rc.y + rc.z) > RT_MIN_CONTRIBUTION)) {
                    color = color + ld3(r.w) * background_color<F>(sc, d);
                    q.flags = 6u;                                   // resolved here, slot becomes free
                    continue;
                }
                if (steps >= a.budget) {
                    // out of budget: this ray and its whole subtree go to the next round
                    if (acc_slot == RT_NO_SLOT) {
                        uint32_t as = wave_alloc(a.acc_alloc);
                        if (as < a.acc_cap) { acc_slot = as; a.acc_out[as] = (uint32_t)out_index; }
                    }
                    bool handed = false;
                    if (acc_slot != RT_NO_SLOT) {
                        uint32_t qi = wave_alloc(a.out_count);
                        if (qi < a.queue_cap) {
                            SpillItem& it = a.out_items[qi];
                            for (int k = 0; k < 3; ++k) { it.o[k] = r.o[k]; it.d[k] = r.d[k]; it.w[k] = r.w[k]; it.c[k] = r.c[k]; }
                            it.depth = r.depth; it.path = r.path;
                            it.out = (uint32_t)out_index; it.acc_slot = acc_slot;
                            handed = true;
                        }
                    }
                    if (handed) { q.flags = 6u; continue; }
                    // no room: trace it here after all
                }
                q.src = RT_LANE_OFF_STACK + slot * (uint32_t)sizeof(LaneRay);
                if (COUNT) { if (primary_trip) cam++; else wk.rays_secondary++; }
            }
            ++steps;
            SceneHit h;
            h.inst = 0; h.c.t = 0.0; h.c.a = 0; h.c.b = 0;
            bool found = trace_scene<F, COUNT, BATCHED>(sc, o, d, shadow, max_distance, h, wk);
            q.t = shadow ? max_distance : h.t;
            q.inst = h.inst; q.found = found ? 1u : 0u;
            q.ct = h.c.t; q.ca = h.c.a; q.cb = h.c.b;
            q.flags = (shadow ? 1u : 0u) | ((shadow && found) ? 2u : 0u);   // an occluded shadow test is settled
        }
        // phase T: the meshes, for all of the trip's queries
        if constexpr (BATCHED) mesh_walk_batch<(F & RTF_LDS_BVH) != 0, COUNT>(sc, lane_mem, jobs, njobs, wk);
        // phase B: add the visible lights, shade the hits, push what they spawn
        for (uint32_t j = 0; j < ns; ++j) {
            if (!jobs[j].found) { color = color + ld3(shadows[j].w); RT_COUNT(wk, shade_evals); }
        }
        uint32_t ns_new = 0;
        int top = sp0;
        for (uint32_t k = 0; k < npop; ++k) {
            const int slot = sp0 - 1 - (int)k;
            const TraceJob q = jobs[ns + k];
            bool reused = false;                                    // the slot holds a new ray
            if (!(q.flags & 4u)) {
                const LaneRay r = stack[slot];
                const d3 rd = ld3(r.d), rw = ld3(r.w);
                if (!q.found) {
                    color = color + rw * background_color<F>(sc, rd);
                } else {
                    SceneHit h;
                    h.t = q.t; h.inst = q.inst; h.c.t = q.ct; h.c.a = q.ca; h.c.b = q.cb;
                    WorldHit wh;
                    world_hit<F, COUNT>(sc, ld3(r.o), rd, h, wh, wk);
                    const DInstance& in = sc.instances[h.inst];
                    if (primary_trip) { distance = h.t; hit_id = in.node_id; }
                    Spawn spawn[RT_SHADE_MAX_SPAWN];
                    uint32_t nspawn;
                    d3 local = shade_hit<F, COUNT>(sc, in.shader, wh, rd, ld3(r.c), stream, r.path, spawn, nspawn, wk);
                    color = color + rw * local;
                    for (uint32_t i = 0; i < nspawn; ++i) {
                        const Spawn& sn = spawn[i];
                        if (sn.shadow) {
                            if (ns_new >= RT_MAX_SHADOW_JOBS) { RT_COUNT(wk, stack_overflows); continue; }
                            ShadowJob& s = shadows[ns_new++];
                            st3(s.o, sn.o); st3(s.d, sn.d); s.dist = sn.c.x; st3(s.w, rw * sn.w);
                        } else {
                            // the first new ray takes over its parent's slot, the others go on top
                            int dst;
                            if (!reused) { dst = slot; reused = true; }
                            else {
                                if (top >= RT_MAX_STACK) { RT_COUNT(wk, stack_overflows); continue; }
                                dst = top++;
                            }
                            LaneRay& s = stack[dst];
                            st3(s.o, sn.o); st3(s.d, sn.d); st3(s.w, rw * sn.w); st3(s.c, sn.c);
                            s.depth = r.depth - 1; s.path = child_path(r.path, sn.ordinal);
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
        ns = ns_new;
    }
    if (a.mode == 0) flush_counters<COUNT>(a.counters, wk, cam, 0);
    else flush_counters<COUNT>(a.counters, wk, 0, cam);
}

// ------------------------------------------------------------------ split samples: accumulators -> f64
// dst is the colour buffer (pass 1) or the AA sample buffer; leaves every accumulator zero again.
__global__ void __launch_bounds__(256)
k_acc_resolve(unsigned long long* __restrict__ acc, uint32_t* __restrict__ acc_flags, const uint32_t* __restrict__ acc_out,
              const unsigned int* __restrict__ acc_alloc, uint32_t acc_cap, double* __restrict__ dst) {
    uint32_t n = min(*acc_alloc, acc_cap);
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        unsigned long long* limbs = acc + (size_t)slot * RT_ACC_LIMBS;
        uint32_t f = acc_flags[slot];
        size_t out = acc_out[slot];
        for (uint32_t ch = 0; ch < 3; ++ch) dst[3 * out + ch] = acc_value(limbs, f, ch);
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
