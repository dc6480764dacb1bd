// rt_kernels.h — the kernels of the render path: pass 1 over all pixels, anti-alias
// detect / resolve / combine, and the clamp-gamma-encode stage.
//
// Frame driver they replace: render_with_config (src/render.rs:99-312) with its
// thread pool (src/multithread.rs).  Work is handed out dynamically: every wave of
// a fixed-size grid pulls 8x8-pixel tiles (pass 1) or 64-sample chunks (AA) from a
// global counter until none are left, which balances the very uneven cost of glass
// pixels against wall pixels the way the reference's 500 row-chunks do for CPU threads.
#pragma once
#include "rt_shade.h"

struct StackRay {         // one pending secondary ray (Ray of src/geometry/ray.rs:4-9 plus its pixel weight)
    d3 o, d;
    d3 w;                 // product of the shader weights from the camera to this ray
    d3 c;                 // Ray::contribution
    uint32_t depth, path;
};

struct SampleResult {
    d3 color;
    double distance;
    uint32_t hit_id;
};

// Scene::cast_ray_get_distance for one camera sample, with the recursion of
// reflect / refract turned into an explicit stack (src/scene.rs:66-85 + shader/*.rs).
template <bool COUNT>
RT_DEV SampleResult trace_sample(const DScene& sc, d3 origin, d3 dir, uint32_t max_depth, uint64_t stream,
                                 Work<COUNT>& wk, unsigned long long& camera_rays) {
    StackRay stack[RT_MAX_STACK];
    int sp = 0;
    stack[0].o = origin; stack[0].d = dir;
    stack[0].w = mk3(1.0, 1.0, 1.0); stack[0].c = mk3(1.0, 1.0, 1.0);
    stack[0].depth = max_depth; stack[0].path = 1u;
    sp = 1;
    SampleResult res;
    res.color = mk3(0.0, 0.0, 0.0); res.distance = INFINITY; res.hit_id = 0xFFFFFFFFu;
    bool primary = true;
    while (sp > 0) {
        StackRay r = stack[--sp];
        bool is_primary = primary;
        primary = false;
        // gate of Scene::cast_ray (scene.rs:67): terminated rays still see the background
        if (!(r.depth > 0 && (r.c.x + r.c.y + r.c.z) > RT_MIN_CONTRIBUTION)) {
            res.color = res.color + r.w * background_color(sc, r.d);
            continue;
        }
        if (COUNT) { if (is_primary) camera_rays++; else wk.rays_secondary++; }
        SceneHit h;
        if (!trace_closest<COUNT>(sc, r.o, r.d, h, wk)) {
            res.color = res.color + r.w * background_color(sc, r.d);
            continue;
        }
        WorldHit wh;
        world_hit<COUNT>(sc, r.o, r.d, h, wh, wk);
        const DInstance& in = sc.instances[h.inst];
        if (is_primary) { res.distance = h.t; res.hit_id = in.node_id; }
        Spawn spawn[RT_SHADE_MAX_SPAWN];
        uint32_t nspawn;
        d3 local = shade_hit<COUNT>(sc, in.shader, wh, r.d, r.c, stream, r.path, spawn, nspawn, wk);
        res.color = res.color + r.w * local;
        for (uint32_t i = nspawn; i-- > 0;) {
            if (sp >= RT_MAX_STACK) { RT_COUNT(wk, stack_overflows); continue; }
            StackRay& s = stack[sp++];
            s.o = spawn[i].o; s.d = spawn[i].d;
            s.w = r.w * spawn[i].w; s.c = spawn[i].c;
            s.depth = r.depth - 1; s.path = child_path(r.path, spawn[i].ordinal);
        }
    }
    return res;
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

// ------------------------------------------------------------------ pass 1 (src/render.rs:152-181)
// rows[] lists the image rows this rank renders (its own row blocks plus the one-row
// halo the AA stencil needs); tiles are 8x8 pixels over (width x num_rows).
template <bool COUNT>
__global__ void __launch_bounds__(64)
k_pass1(DScene sc, DRenderParams rp, const uint32_t* __restrict__ rows, uint32_t num_rows,
        double* __restrict__ color, double* __restrict__ distance, uint32_t* __restrict__ hit_id,
        unsigned int* __restrict__ next_tile, DCounters* __restrict__ counters) {
    const uint32_t W = rp.cam.width;
    const uint32_t tiles_x = (W + 7) / 8, tiles_y = (num_rows + 7) / 8;
    const uint32_t total = tiles_x * tiles_y;
    const uint32_t lane = threadIdx.x;
    Work<COUNT> wk = {};
    unsigned long long cam = 0;
    for (;;) {
        uint32_t tile = 0;
        if (lane == 0) tile = atomicAdd(next_tile, 1u);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= total) break;
        uint32_t x = (tile % tiles_x) * 8 + (lane & 7), ry = (tile / tiles_x) * 8 + (lane >> 3);
        if (x >= W || ry >= num_rows) continue;
        uint32_t y = rows[ry];
        size_t pixel = (size_t)y * W + x;
        d3 dir = camera_dir(rp.cam, (double)x + 0.5, (double)y + 0.5);
        SampleResult r = trace_sample<COUNT>(sc, ld3(rp.cam.origin), dir, rp.recursion_depth, rng_stream(rp.seed, pixel, 0), wk, cam);
        color[3 * pixel] = r.color.x; color[3 * pixel + 1] = r.color.y; color[3 * pixel + 2] = r.color.z;
        distance[pixel] = r.distance;
        hit_id[pixel] = r.hit_id;
    }
    flush_counters<COUNT>(counters, wk, cam, 0);
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

// ------------------------------------------------------------------ AA resolve (src/render.rs:259-299)
// One work item per (flagged pixel, extra ray).
template <bool COUNT>
__global__ void __launch_bounds__(64)
k_aa_resolve(DScene sc, DRenderParams rp, const uint32_t* __restrict__ list, const unsigned int* __restrict__ count,
             double* __restrict__ samples, unsigned int* __restrict__ next_chunk, DCounters* __restrict__ counters) {
    const uint32_t total = *count * rp.aa_rays;
    const uint32_t lane = threadIdx.x;
    const uint32_t W = rp.cam.width;
    const double theta = 2.0 * RT_PI / (double)rp.aa_rays;        // :227
    Work<COUNT> wk = {};
    unsigned long long cam = 0;
    for (;;) {
        uint32_t chunk = 0;
        if (lane == 0) chunk = atomicAdd(next_chunk, 1u);
        chunk = __builtin_amdgcn_readfirstlane(chunk);
        if ((unsigned long long)chunk * 64ull >= total) break;
        uint32_t item = chunk * 64 + lane;
        if (item >= total) continue;
        uint32_t f = item / rp.aa_rays, i = item % rp.aa_rays;
        uint32_t pixel = list[f];
        uint32_t x = pixel % W, y = pixel / W;
        uint64_t stream0 = rng_stream(rp.seed, pixel, 0);
        double rand_rotation = 0.0 + (2.0 * RT_PI - 0.0) * rng_aa(stream0, 0);      // :266
        double ax = cos(rand_rotation + theta * (double)i);
        double ay = sin(rand_rotation + theta * (double)i);
        double rand_distance = 0.2 + (0.5 - 0.2) * rng_aa(stream0, 1 + i);          // :277
        double x_pos = (double)x + 0.5 + ax * rand_distance;
        double y_pos = (double)y + 0.5 + ay * rand_distance;
        d3 dir = camera_dir(rp.cam, x_pos, y_pos);
        SampleResult r = trace_sample<COUNT>(sc, ld3(rp.cam.origin), dir, rp.recursion_depth, rng_stream(rp.seed, pixel, 1 + i), wk, cam);
        samples[3 * (size_t)item] = r.color.x; samples[3 * (size_t)item + 1] = r.color.y; samples[3 * (size_t)item + 2] = r.color.z;
    }
    flush_counters<COUNT>(counters, wk, 0, cam);
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
