// rt_ingest.hip — the stage before the render path, on the device: Wavefront OBJ text -> unified-vertex
// triangle mesh (what Mesh::from_path gets from tobj 0.1.7's load_obj, src/mesh.rs:383-414), the
// meshes' bounding hierarchies (the reference builds its kd-tree on the host, src/mesh.rs:124-144,
// 281-332), and the vertical flip of texture images (src/texture.rs:46, src/normal_map.rs:52,116).
//
// Everything here is integer / byte work over arrays that are about to live in HBM anyway, organised
// as data-parallel passes with prefix sums between them:
//   OBJ     bytes -> line starts -> per-line kind and corner count -> prefix sums give every line its
//           place in the v / vt / vn pools and in the corner and triangle arrays -> per-line parse ->
//           the (model, v, vt, vn) corners are de-duplicated through a hash table that keeps the first
//           corner of every key -> a prefix sum over "first of its key" numbers the unified vertices
//           in first-appearance order, which is the order a sequential reader gives them.
//   BVH     binned SAH, top down, one launch per level and one workgroup per node: box and centroid
//           bounds, 3 x 16 bins with LDS atomics on order-preserving keys, the sweep, a stable
//           partition.  Every step is either exact (min / max, counts) or evaluated in the host
//           builder's order (the sweep), so the tree is the host builder's tree, node for node.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_flatten.hpp"

int rt_set_error(int code, const std::string& msg);   // rt_api.hip

namespace {

int ing_fail(int code, const std::string& msg) { return rt_set_error(code, msg); }

struct HipFailure { std::string what; };
#define ING_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) throw HipFailure{std::string(#expr) + ": " + hipGetErrorString(e_)}; \
    } while (0)

// Device buffers of one ingest call, freed together.
struct Pool {
    std::vector<void*> all;
    template <class T> T* get(size_t n) {
        void* p = nullptr;
        ING_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        all.push_back(p);
        return static_cast<T*>(p);
    }
    ~Pool() { for (void* p : all) (void)hipFree(p); }
};

// ------------------------------------------------------------------ exclusive prefix sum (u32, in place)
constexpr uint32_t SCAN_ITEMS = 8, SCAN_BLOCK = 256, SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

__device__ uint32_t block_exclusive(uint32_t v, uint32_t* lds, uint32_t& total) {   // 256 threads; lds: 5 words
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(incl, off); if ((int)lane >= off) incl += o; }
    __syncthreads();
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    return before + incl - v;
}

__global__ void __launch_bounds__(256) k_scan_tiles(uint32_t* data, uint32_t n, uint32_t* tile_sums) {
    __shared__ uint32_t lds[4];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) { v[k] = base + k < n ? data[base + k] : 0u; sum += v[k]; }
    uint32_t total;
    uint32_t run = block_exclusive(sum, lds, total);
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) data[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(256) k_scan_sums(uint32_t* sums, uint32_t n, uint32_t* grand_total) {   // one workgroup
    __shared__ uint32_t lds[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += SCAN_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive(v, lds, total);
        if (i < n) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
__global__ void __launch_bounds__(256) k_scan_add(uint32_t* data, uint32_t n, const uint32_t* tile_sums) {
    const uint32_t add = tile_sums[blockIdx.x];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) data[base + k] += add;
}
// data[i] <- sum of data[0..i); returns the sum of all (waits for the stream).
uint32_t exclusive_scan(Pool& pool, uint32_t* data, uint32_t n, hipStream_t st) {
    if (n == 0) return 0;
    const uint32_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint32_t* sums = pool.get<uint32_t>(tiles + 1);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(SCAN_BLOCK), 0, st, data, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, st, sums, tiles, sums + tiles);
    hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(SCAN_BLOCK), 0, st, data, n, sums);
    ING_HIP(hipGetLastError());
    uint32_t total = 0;
    ING_HIP(hipMemcpyAsync(&total, sums + tiles, 4, hipMemcpyDeviceToHost, st));
    ING_HIP(hipStreamSynchronize(st));
    return total;
}

// ================================================================== bounding hierarchy
struct GTri { double lo[3], hi[3], c[3]; uint32_t face, pad; };   // a triangle's record, in registers
// ... and in memory, component by component (component k of triangle i at f[k * n + i]): the lanes of a wave work on
// neighbouring triangles, so every load and store covers whole cache lines.
struct TriArray {
    double* f; uint32_t* face; uint32_t n;
    __device__ GTri load(uint32_t i) const {
        GTri t;
        for (int k = 0; k < 3; ++k) { t.lo[k] = f[(size_t)k * n + i]; t.hi[k] = f[(size_t)(3 + k) * n + i]; t.c[k] = f[(size_t)(6 + k) * n + i]; }
        t.face = face[i]; t.pad = 0;
        return t;
    }
    __device__ void store(uint32_t i, const GTri& t) const {
        for (int k = 0; k < 3; ++k) { f[(size_t)k * n + i] = t.lo[k]; f[(size_t)(3 + k) * n + i] = t.hi[k]; f[(size_t)(6 + k) * n + i] = t.c[k]; }
        face[i] = t.face;
    }
    __device__ double centre(uint32_t i, int axis) const { return f[(size_t)(6 + axis) * n + i]; }
};
struct GNode { double lo[3], hi[3]; uint32_t first, count; int left, right; uint32_t depth, pad; };
constexpr int BINS = 16;                        // as the host builder (rt_flatten.hpp, build_bvh)
constexpr uint32_t LEAF = 2;

__global__ void __launch_bounds__(256) k_tri_boxes(const double* P, const uint32_t* F, uint32_t n, TriArray out) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    GTri t;
    const double* a = P + 3 * (size_t)F[3 * f];
    const double* b = P + 3 * (size_t)F[3 * f + 1];
    const double* c = P + 3 * (size_t)F[3 * f + 2];
    for (int k = 0; k < 3; ++k) {
        t.lo[k] = fmin(a[k], fmin(b[k], c[k]));
        t.hi[k] = fmax(a[k], fmax(b[k], c[k]));
        t.c[k] = (t.lo[k] + t.hi[k]) * 0.5;
    }
    t.face = f; t.pad = 0;
    out.store(f, t);
}

// doubles as unsigned keys with the same order (for integer min / max atomics in LDS)
__device__ inline unsigned long long order_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ inline double block_reduce(double v, bool take_min, double* lds) {   // up to 16 waves; the result on every thread
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(v, off); v = take_min ? fmin(v, o) : fmax(v, o); }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = lds[0];
    for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) r = take_min ? fmin(r, lds[w]) : fmax(r, lds[w]);
    return r;
}
__device__ inline double box_area(const double* lo, const double* hi) {
    const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}

// The sweep over the bins of the three axes, in the host builder's order of evaluation (rt_flatten.hpp, build_bvh):
// the first split of least cost in (axis, bin) order, or axis -1 when no axis has an extent.
__device__ inline void sweep_bins(const unsigned long long (*blo)[BINS][3], const unsigned long long (*bhi)[BINS][3], const unsigned int (*bcount)[BINS],
                                  const double* clo, const double* chi, int& out_axis, int& out_bin) {
    const double inf = INFINITY;
    double best_cost = inf; int best_axis = -1, best_bin = -1;
    for (int a = 0; a < 3; ++a) {
        if (!(chi[a] > clo[a])) continue;
        double right_area[BINS]; uint32_t right_n[BINS];
        double rl[3] = {inf, inf, inf}, rh[3] = {-inf, -inf, -inf}; uint32_t rn = 0;
        for (int b = BINS - 1; b > 0; --b) {
            rn += bcount[a][b];
            for (int k = 0; k < 3; ++k) { rl[k] = fmin(rl[k], key_value(blo[a][b][k])); rh[k] = fmax(rh[k], key_value(bhi[a][b][k])); }
            right_area[b] = rn ? box_area(rl, rh) : 0.0; right_n[b] = rn;
        }
        double ll[3] = {inf, inf, inf}, lh[3] = {-inf, -inf, -inf}; uint32_t ln = 0;
        for (int b = 0; b < BINS - 1; ++b) {
            ln += bcount[a][b];
            for (int k = 0; k < 3; ++k) { ll[k] = fmin(ll[k], key_value(blo[a][b][k])); lh[k] = fmax(lh[k], key_value(bhi[a][b][k])); }
            if (ln == 0 || right_n[b + 1] == 0) continue;
            const double cost = ln * box_area(ll, lh) + right_n[b + 1] * right_area[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
    }
    out_axis = best_axis; out_bin = best_bin;
}

// Nodes of at most 64 triangles (most nodes of a hierarchy): one wave per node, four nodes per workgroup, a triangle
// per lane, no workgroup barriers.  Same steps, same arithmetic and same order as k_build_level, which leaves these
// nodes alone when `small_apart` is set.
__global__ void __launch_bounds__(256)
k_build_level_small(TriArray src, TriArray dst, GNode* nodes, uint32_t level_start, uint32_t level_end, uint32_t node_cap,
                    uint32_t* node_count, uint32_t* order, int max_depth, uint32_t* error) {
    __shared__ unsigned long long blo_all[4][3][BINS][3], bhi_all[4][3][BINS][3];
    __shared__ unsigned int bcount_all[4][3][BINS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t ni = level_start + blockIdx.x * 4u + wave;
    if (ni >= level_end) return;
    GNode& nd = nodes[ni];
    const uint32_t first = nd.first, count = nd.count, depth = nd.depth;
    if (count > 64u) return;
    unsigned long long (*blo)[BINS][3] = blo_all[wave];
    unsigned long long (*bhi)[BINS][3] = bhi_all[wave];
    unsigned int (*bcount)[BINS] = bcount_all[wave];
    const double inf = INFINITY;
    const bool mine = lane < count;
    GTri t{};
    if (mine) t = src.load(first + lane);
    double lo[3], hi[3], clo[3], chi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = mine ? t.lo[k] : inf; hi[k] = mine ? t.hi[k] : -inf; clo[k] = mine ? t.c[k] : inf; chi[k] = mine ? t.c[k] : -inf;
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fmin(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], off));
            clo[k] = fmin(clo[k], __shfl_xor(clo[k], off)); chi[k] = fmax(chi[k], __shfl_xor(chi[k], off));
        }
    }
    if (lane == 0) for (int k = 0; k < 3; ++k) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
    if (count <= LEAF || (int)depth >= max_depth) {
        if (count > LEAF && count > 7u && lane == 0) atomicOr(error, 1u);
        if (mine) order[first + lane] = t.face;
        if (lane == 0) { nd.left = -1; nd.right = -1; }
        return;
    }
    for (uint32_t i = lane; i < 3 * BINS; i += 64u) {
        bcount[i / BINS][i % BINS] = 0;
        for (int k = 0; k < 3; ++k) { blo[i / BINS][i % BINS][k] = order_key(inf); bhi[i / BINS][i % BINS][k] = order_key(-inf); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double scale[3];
    for (int a = 0; a < 3; ++a) scale[a] = chi[a] > clo[a] ? (double)BINS / (chi[a] - clo[a]) : 0.0;
    if (mine) {
        for (int a = 0; a < 3; ++a) {
            if (!(chi[a] > clo[a])) continue;
            const int b = min(BINS - 1, (int)((t.c[a] - clo[a]) * scale[a]));
            atomicAdd(&bcount[a][b], 1u);
            for (int k = 0; k < 3; ++k) { atomicMin(&blo[a][b][k], order_key(t.lo[k])); atomicMax(&bhi[a][b][k], order_key(t.hi[k])); }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int axis = -1, bin = -1;
    if (lane == 0) sweep_bins(blo, bhi, bcount, clo, chi, axis, bin);
    axis = __shfl(axis, 0); bin = __shfl(bin, 0);
    uint32_t mid;
    if (axis < 0) {
        mid = first + count / 2;
        if (mine) dst.store(first + lane, t);
    } else {
        const bool left = mine && min(BINS - 1, (int)((t.c[axis] - clo[axis]) * scale[axis])) <= bin;
        const unsigned long long ml = __ballot(left), mr = __ballot(mine && !left), lt = (1ull << lane) - 1ull;
        const uint32_t n_left = (uint32_t)__popcll(ml);
        if (mine) dst.store(first + (left ? (uint32_t)__popcll(ml & lt) : n_left + (uint32_t)__popcll(mr & lt)), t);
        mid = first + n_left;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    if (lane == 0) {
        const uint32_t c = atomicAdd(node_count, 2u);
        if (c + 2u > node_cap) { atomicOr(error, 2u); nd.left = -1; nd.right = -1; return; }
        GNode l{}, r{};
        l.first = first; l.count = mid - first; l.depth = depth + 1; l.left = l.right = -1;
        r.first = mid; r.count = first + count - mid; r.depth = depth + 1; r.left = r.right = -1;
        nodes[c] = l; nodes[c + 1] = r;
        nd.left = (int)c; nd.right = (int)(c + 1);
        const uint32_t large = (l.count > 64u ? 1u : 0u) + (r.count > 64u ? 1u : 0u);
        if (large) atomicAdd(node_count + 2, large);             // (nodes of the next level that k_build_level has to take)
    }
}

// One workgroup per node of the level [level_start, level_start + gridDim.x).  Reads the node's triangles from
// `src`; a leaf writes their faces to `order` (its final place), an inner node writes them partitioned to `dst`
// and appends its two children to `nodes`.
__global__ void __launch_bounds__(1024)
k_build_level(TriArray src, TriArray dst, GNode* nodes, uint32_t level_start, uint32_t node_cap,
              uint32_t* node_count, uint32_t* order, int max_depth, uint32_t* error, uint32_t small_apart) {
    __shared__ double red[16];
    __shared__ unsigned long long blo[3][BINS][3], bhi[3][BINS][3];
    __shared__ unsigned int bcount[3][BINS];
    __shared__ int s_axis, s_bin;
    __shared__ unsigned int s_left, wl[16], wr[16];
    const uint32_t nwaves = blockDim.x >> 6;
    GNode& nd = nodes[level_start + blockIdx.x];
    const uint32_t first = nd.first, count = nd.count, depth = nd.depth;
    if (small_apart && count <= 64u) return;                      // k_build_level_small's
    const double inf = INFINITY;
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, clo[3] = {inf, inf, inf}, chi[3] = {-inf, -inf, -inf};
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const GTri t = src.load(first + i);
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], t.lo[k]); hi[k] = fmax(hi[k], t.hi[k]);
            clo[k] = fmin(clo[k], t.c[k]); chi[k] = fmax(chi[k], t.c[k]);
        }
    }
    for (int k = 0; k < 3; ++k) {
        lo[k] = block_reduce(lo[k], true, red); hi[k] = block_reduce(hi[k], false, red);
        clo[k] = block_reduce(clo[k], true, red); chi[k] = block_reduce(chi[k], false, red);
    }
    if (threadIdx.x == 0) for (int k = 0; k < 3; ++k) { nd.lo[k] = lo[k]; nd.hi[k] = hi[k]; }
    if (count <= LEAF || (int)depth >= max_depth) {
        if (count > LEAF && count > 7u && threadIdx.x == 0) atomicOr(error, 1u);     // deeper than the walk's stack
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) order[first + i] = src.face[first + i];
        if (threadIdx.x == 0) { nd.left = -1; nd.right = -1; }
        return;
    }
    // bins of the three axes
    for (uint32_t i = threadIdx.x; i < 3 * BINS; i += blockDim.x) {
        bcount[i / BINS][i % BINS] = 0;
        for (int k = 0; k < 3; ++k) { blo[i / BINS][i % BINS][k] = order_key(inf); bhi[i / BINS][i % BINS][k] = order_key(-inf); }
    }
    __syncthreads();
    double scale[3];
    for (int a = 0; a < 3; ++a) scale[a] = chi[a] > clo[a] ? (double)BINS / (chi[a] - clo[a]) : 0.0;
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
        const GTri t = src.load(first + i);
        for (int a = 0; a < 3; ++a) {
            if (!(chi[a] > clo[a])) continue;
            const int b = min(BINS - 1, (int)((t.c[a] - clo[a]) * scale[a]));
            atomicAdd(&bcount[a][b], 1u);
            // (a bin's bounds soon stop moving: a plain read first, the atomic only when it would change something)
            for (int k = 0; k < 3; ++k) {
                const unsigned long long kl = order_key(t.lo[k]), kh = order_key(t.hi[k]);
                if (kl < blo[a][b][k]) atomicMin(&blo[a][b][k], kl);
                if (kh > bhi[a][b][k]) atomicMax(&bhi[a][b][k], kh);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best_axis, best_bin;
        sweep_bins(blo, bhi, bcount, clo, chi, best_axis, best_bin);
        s_axis = best_axis; s_bin = best_bin; s_left = 0;
    }
    __syncthreads();
    const int axis = s_axis, bin = s_bin;
    uint32_t mid;
    if (axis < 0) {                   // all centroids coincide: split in the middle
        mid = first + count / 2;
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) dst.store(first + i, src.load(first + i));
    } else {
        const double a0 = clo[axis], k = scale[axis];
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x)
            if (min(BINS - 1, (int)((src.centre(first + i, axis) - a0) * k)) <= bin) ++mine;
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&s_left, mine);
        __syncthreads();
        const uint32_t n_left = s_left;
        // stable partition: chunk by chunk, ranks from ballots within the wave and wave totals within the chunk
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const unsigned long long lt = (1ull << lane) - 1ull;
        uint32_t run_left = 0, run_right = 0;
        for (uint32_t base = 0; base < count; base += blockDim.x) {
            const uint32_t i = base + threadIdx.x;
            const bool valid = i < count;
            GTri t;
            bool left = false;
            if (valid) { t = src.load(first + i); left = min(BINS - 1, (int)((t.c[axis] - a0) * k)) <= bin; }
            const unsigned long long ml = __ballot(valid && left), mr = __ballot(valid && !left);
            if (lane == 0) { wl[wave] = (unsigned int)__popcll(ml); wr[wave] = (unsigned int)__popcll(mr); }
            __syncthreads();
            uint32_t bl = 0, br = 0, tl = 0, tr = 0;
            for (uint32_t w = 0; w < nwaves; ++w) { if (w < wave) { bl += wl[w]; br += wr[w]; } tl += wl[w]; tr += wr[w]; }
            if (valid) {
                const uint32_t at = left ? run_left + bl + (uint32_t)__popcll(ml & lt) : n_left + run_right + br + (uint32_t)__popcll(mr & lt);
                dst.store(first + at, t);
            }
            run_left += tl; run_right += tr;
            __syncthreads();
        }
        mid = first + n_left;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    if (threadIdx.x == 0) {
        const uint32_t c = atomicAdd(node_count, 2u);
        if (c + 2u > node_cap) { atomicOr(error, 2u); nd.left = -1; nd.right = -1; return; }
        GNode l{}, r{};
        l.first = first; l.count = mid - first; l.depth = depth + 1; l.left = l.right = -1;
        r.first = mid; r.count = first + count - mid; r.depth = depth + 1; r.left = r.right = -1;
        nodes[c] = l; nodes[c + 1] = r;
        nd.left = (int)c; nd.right = (int)(c + 1);
        const uint32_t large = (l.count > 64u ? 1u : 0u) + (r.count > 64u ? 1u : 0u);
        if (large) atomicAdd(node_count + 2, large);             // (nodes of the next level that k_build_level has to take)
    }
}

void device_build(int device, const double* P, uint32_t npos, const uint32_t* F, uint32_t nfaces, int max_depth,
                  std::vector<uint32_t>& face_order, std::vector<rtflat::BuildNode>& bn) {
    const auto t_begin = std::chrono::steady_clock::now();
    rtflat::DeviceRestore restore;
    try {
        ING_HIP(hipSetDevice(device));
        Pool pool;
        hipStream_t st = nullptr;
        double* dP = pool.get<double>(3 * (size_t)npos);
        uint32_t* dF = pool.get<uint32_t>(3 * (size_t)nfaces);
        ING_HIP(hipMemcpyAsync(dP, P, 3 * (size_t)npos * sizeof(double), hipMemcpyHostToDevice, st));
        ING_HIP(hipMemcpyAsync(dF, F, 3 * (size_t)nfaces * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        TriArray tri[2];
        for (TriArray& t : tri) { t.f = pool.get<double>(9 * (size_t)nfaces); t.face = pool.get<uint32_t>(nfaces); t.n = nfaces; }
        const uint32_t node_cap = 2 * nfaces + 2;
        GNode* nodes = pool.get<GNode>(node_cap);
        uint32_t* order = pool.get<uint32_t>(nfaces);
        uint32_t* scalars = pool.get<uint32_t>(3);          // node count, error bits, large nodes of the next level
        hipLaunchKernelGGL(k_tri_boxes, dim3((nfaces + 255) / 256), dim3(256), 0, st, dP, dF, nfaces, tri[0]);
        GNode root{};
        root.first = 0; root.count = nfaces; root.depth = 0; root.left = root.right = -1;
        const uint32_t init[3] = {1u, 0u, 0u};
        uint32_t large_nodes = nfaces > 64u ? 1u : 0u;
        ING_HIP(hipMemcpyAsync(nodes, &root, sizeof root, hipMemcpyHostToDevice, st));
        ING_HIP(hipMemcpyAsync(scalars, init, sizeof init, hipMemcpyHostToDevice, st));
        uint32_t level_start = 0, level_end = 1;
        int from = 0;
        while (level_end > level_start) {
            // (the first levels hold few nodes with many triangles each: large workgroups there, small ones below)
            const uint32_t level_nodes = level_end - level_start;
            const uint32_t block = (uint64_t)level_nodes * 2048u <= nfaces ? 1024u : ((uint64_t)level_nodes * 512u <= nfaces ? 512u : 256u);
            // (the level's nodes of more than 64 triangles, counted when they were made: without any, no large-node launch)
            if (large_nodes)
                hipLaunchKernelGGL(k_build_level, dim3(level_nodes), dim3(block), 0, st, tri[from], tri[1 - from], nodes, level_start, node_cap,
                                   scalars, order, max_depth, scalars + 1, 1u);
            hipLaunchKernelGGL(k_build_level_small, dim3((level_nodes + 3u) / 4u), dim3(256), 0, st, tri[from], tri[1 - from], nodes, level_start, level_end,
                               node_cap, scalars, order, max_depth, scalars + 1);
            ING_HIP(hipGetLastError());
            uint32_t now[3];
            ING_HIP(hipMemcpyAsync(now, scalars, sizeof now, hipMemcpyDeviceToHost, st));
            ING_HIP(hipMemsetAsync(scalars + 2, 0, sizeof(uint32_t), st));
            ING_HIP(hipStreamSynchronize(st));
            large_nodes = now[2];
            if (now[1] & 1u) throw rtflat::Unsupported("bounding hierarchy deeper than the device walk stack");
            if (now[1] & 2u) throw HipFailure{"device hierarchy build ran out of node slots"};
            level_start = level_end; level_end = std::min(now[0], node_cap);
            from = 1 - from;
        }
        const auto t_levels = std::chrono::steady_clock::now();
        std::vector<GNode> g(level_end);
        face_order.resize(nfaces);
        ING_HIP(hipMemcpy(g.data(), nodes, g.size() * sizeof(GNode), hipMemcpyDeviceToHost));
        ING_HIP(hipMemcpy(face_order.data(), order, face_order.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const auto t_copied = std::chrono::steady_clock::now();
        // nodes renumbered in the order the host builder's recursion creates them (children allocated before descending)
        // (the recursion "allocate both children, descend left, descend right" with a stack of its own: the right child waits on it)
        bn.assign(g.size(), rtflat::BuildNode{});
        std::vector<std::pair<int, uint32_t>> todo;              // (index in bn, index in g)
        todo.reserve(64);
        todo.emplace_back(0, 0u);
        int next = 1;
        while (!todo.empty()) {
            const int at = todo.back().first;
            const GNode& n = g[todo.back().second];
            todo.pop_back();
            bn[at].lo = {n.lo[0], n.lo[1], n.lo[2]}; bn[at].hi = {n.hi[0], n.hi[1], n.hi[2]};
            if (n.left < 0) { bn[at].first = n.first; bn[at].count = n.count; continue; }
            const int l = next++, r = next++;
            bn[at].left = l; bn[at].right = r;
            todo.emplace_back(r, (uint32_t)n.right);
            todo.emplace_back(l, (uint32_t)n.left);
        }
        if (std::getenv("RT_DEBUG_INGEST"))
            std::fprintf(stderr, "[rt] device hierarchy build: %u faces, %zu nodes, levels %.2f ms, copies back %.2f ms, renumbering %.2f ms\n", nfaces, g.size(),
                         std::chrono::duration<double, std::milli>(t_levels - t_begin).count(), std::chrono::duration<double, std::milli>(t_copied - t_levels).count(),
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_copied).count());
    } catch (const HipFailure& e) {
        throw rtflat::DeviceFailure("device hierarchy build: " + e.what);
    }
}

// ================================================================== texture rows
__global__ void __launch_bounds__(256) k_flip_rows(uint8_t* texels, uint32_t row_bytes, uint32_t height) {
    const uint32_t y = blockIdx.x;                      // rows y and height-1-y change places
    uint8_t* a = texels + (size_t)y * row_bytes;
    uint8_t* b = texels + (size_t)(height - 1 - y) * row_bytes;
    for (uint32_t i = threadIdx.x; i < row_bytes; i += blockDim.x) { const uint8_t t = a[i]; a[i] = b[i]; b[i] = t; }
}

// ================================================================== OBJ text
enum : uint8_t { L_OTHER = 0, L_V = 1, L_VT = 2, L_VN = 3, L_F = 4, L_OG = 5 };

__device__ inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// number of line starts in each tile of SCAN_TILE bytes / the line starts themselves
__global__ void __launch_bounds__(256) k_line_flags(const char* text, uint32_t n, uint32_t* flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || text[i - 1] == '\n') ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_line_starts(const char* text, uint32_t n, const uint32_t* index, uint32_t* starts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (i == 0 || text[i - 1] == '\n')) starts[index[i]] = i;
}

// Per line: its kind, and for a face line the number of corners.  One thread per line.
__global__ void __launch_bounds__(256)
k_classify(const char* text, uint32_t n, const uint32_t* starts, uint32_t nlines, uint32_t* is_v, uint32_t* is_vt, uint32_t* is_vn,
           uint32_t* is_f, uint32_t* is_og, uint32_t* corners, uint32_t* tris, uint8_t* kind) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines) return;
    uint32_t p = starts[l];
    const uint32_t end = l + 1 < nlines ? starts[l + 1] : n;
    auto at = [&](uint32_t q) { return q < end ? text[q] : '\n'; };
    while (p < end && is_blank(at(p))) ++p;
    uint8_t k = L_OTHER;
    const char c0 = at(p), c1 = at(p + 1), c2 = at(p + 2);
    auto ends_tag = [&](char c) { return is_blank(c) || c == '\n'; };
    if (c0 == 'v' && ends_tag(c1)) k = L_V;
    else if (c0 == 'v' && c1 == 't' && ends_tag(c2)) k = L_VT;
    else if (c0 == 'v' && c1 == 'n' && ends_tag(c2)) k = L_VN;
    else if (c0 == 'f' && ends_tag(c1)) k = L_F;
    else if ((c0 == 'o' || c0 == 'g') && ends_tag(c1)) k = L_OG;
    uint32_t nc = 0;
    if (k == L_F) {
        ++p;
        for (;;) {
            while (p < end && is_blank(at(p))) ++p;
            if (p >= end || at(p) == '\n') break;
            ++nc;
            while (p < end && !is_blank(at(p)) && at(p) != '\n') ++p;
        }
    }
    kind[l] = k;
    is_v[l] = k == L_V; is_vt[l] = k == L_VT; is_vn[l] = k == L_VN; is_f[l] = k == L_F; is_og[l] = k == L_OG;
    corners[l] = nc; tris[l] = nc >= 3 ? nc - 2 : 0u;
}

__constant__ double c_pow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// A decimal literal -> the nearest f32 (what Rust's str::parse::<f32> and C's strtof return).  The value is
// formed in f64 with one rounding (integer below 2^53 times or over an exact power of ten) and rounded again to
// f32; the two roundings together can only differ from a single one when the f64 value sits exactly halfway
// between two f32 values.  That case, and every literal outside the simple range, is not decided here:
// `unsure` is set and the host parses that literal itself.
__device__ float parse_f32(const char* text, uint32_t& p, uint32_t end, bool& unsure) {
    while (p < end && is_blank(text[p])) ++p;
    const uint32_t begin = p;
    bool neg = false;
    if (p < end && (text[p] == '+' || text[p] == '-')) { neg = text[p] == '-'; ++p; }
    unsigned long long mant = 0;
    int digits = 0, e10 = 0;
    bool any = false, dropped = false;
    while (p < end && text[p] >= '0' && text[p] <= '9') {
        any = true;
        if (digits < 18) { mant = mant * 10ull + (unsigned)(text[p] - '0'); if (mant) ++digits; } else { ++e10; if (text[p] != '0') dropped = true; }
        ++p;
    }
    if (p < end && text[p] == '.') {
        ++p;
        while (p < end && text[p] >= '0' && text[p] <= '9') {
            any = true;
            if (digits < 18) { mant = mant * 10ull + (unsigned)(text[p] - '0'); if (mant) ++digits; --e10; } else if (text[p] != '0') dropped = true;
            ++p;
        }
    }
    if (any && p < end && (text[p] == 'e' || text[p] == 'E')) {
        uint32_t q = p + 1;
        bool eneg = false;
        if (q < end && (text[q] == '+' || text[q] == '-')) { eneg = text[q] == '-'; ++q; }
        if (q < end && text[q] >= '0' && text[q] <= '9') {
            int ev = 0;
            while (q < end && text[q] >= '0' && text[q] <= '9') { if (ev < 10000) ev = ev * 10 + (text[q] - '0'); ++q; }
            e10 += eneg ? -ev : ev;
            p = q;
        }
    }
    // (a token that is no number — "nan", "inf", garbage — is left to the host as well)
    if (p < end && !is_blank(text[p]) && text[p] != '\n') {      // more of the token follows: the host's strtof reads what it can of it
        unsure = true;
        while (p < end && !is_blank(text[p]) && text[p] != '\n') ++p;
        return 0.0f;
    }
    if (!any) { unsure = p != begin; return 0.0f; }              // a bare sign
    if (mant == 0) return neg ? -0.0f : 0.0f;
    if (dropped || mant >= (1ull << 53) || e10 < -22 || e10 > 22) { unsure = true; return 0.0f; }
    const double d = e10 < 0 ? (double)mant / c_pow10[-e10] : (double)mant * c_pow10[e10];
    const double ad = fabs(d);
    if (!(ad > 1e-30 && ad < 1e30)) { unsure = true; return 0.0f; }
    if (((unsigned long long)__double_as_longlong(d) & 0x1FFFFFFFull) == 0x10000000ull) { unsure = true; return 0.0f; }
    const float f = (float)d;
    return neg ? -f : f;
}

__device__ inline long parse_long(const char* text, uint32_t& p, uint32_t end) {
    bool neg = false;
    if (p < end && (text[p] == '+' || text[p] == '-')) { neg = text[p] == '-'; ++p; }
    long v = 0;
    while (p < end && text[p] >= '0' && text[p] <= '9') { if (v < (1l << 40)) v = v * 10 + (text[p] - '0'); ++p; }
    return neg ? -v : v;
}

struct ObjArgs {
    const char* text; uint32_t n;
    const uint32_t* starts; uint32_t nlines;
    const uint8_t* kind;
    const uint32_t *v_at, *vt_at, *vn_at, *f_at, *og_at, *corner_at, *tri_at;   // exclusive prefix sums per line
    const uint32_t* og_model;      // per o/g line (in their own numbering): effective boundaries before it, itself included
    uint32_t num_og;
    uint32_t nv, nvt, nvn;
    float *v, *vt, *vn;            // the pools
    int *ck_v, *ck_t, *ck_n; uint32_t* ck_model;   // per corner
    uint32_t* corner_first;        // per face line: its first corner (= corner_at) kept for the triangle pass
    uint32_t* unsure; uint32_t unsure_cap; uint32_t* unsure_count;   // (pool 1/2/3, float index, text offset) triples
    uint32_t* error;
};

// o/g lines: boundary k takes effect when at least one face line lies between the previous o/g line and it
__global__ void __launch_bounds__(256) k_og_effective(const uint32_t* starts_unused, const uint8_t* kind, const uint32_t* og_at, const uint32_t* f_at,
                                                        uint32_t nlines, uint32_t* og_faces_before) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < nlines && kind[l] == L_OG) og_faces_before[og_at[l]] = f_at[l];
}
__global__ void __launch_bounds__(256) k_og_flags(const uint32_t* og_faces_before, uint32_t num_og, uint32_t* effective) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < num_og) effective[k] = og_faces_before[k] > (k ? og_faces_before[k - 1] : 0u) ? 1u : 0u;
}

__device__ inline void note_unsure(const ObjArgs& a, uint32_t pool, uint32_t index, uint32_t offset) {
    const uint32_t at = atomicAdd(a.unsure_count, 1u);
    if (at < a.unsure_cap) { a.unsure[3 * at] = pool; a.unsure[3 * at + 1] = index; a.unsure[3 * at + 2] = offset; }
}

__global__ void __launch_bounds__(256) k_parse_lines(ObjArgs a) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= a.nlines) return;
    const uint8_t k = a.kind[l];
    if (k == L_OG) return;
    uint32_t p = a.starts[l];
    const uint32_t end = l + 1 < a.nlines ? a.starts[l + 1] : a.n;
    while (p < end && is_blank(a.text[p])) ++p;
    if (k == L_OTHER) {   // point and line elements go through the reference parser's face path and add vertices: not read here
        const char c0 = p < end ? a.text[p] : '\n', c1 = p + 1 < end ? a.text[p + 1] : '\n';
        if ((c0 == 'l' || c0 == 'p') && (is_blank(c1) || c1 == '\n')) atomicOr(a.error, 2u);
        return;
    }
    p += (k == L_V || k == L_F) ? 1u : 2u;
    if (k == L_V || k == L_VT || k == L_VN) {
        const uint32_t comps = k == L_VT ? 2u : 3u;
        float* pool = k == L_V ? a.v : k == L_VT ? a.vt : a.vn;
        const uint32_t at = (k == L_V ? a.v_at : k == L_VT ? a.vt_at : a.vn_at)[l];
        for (uint32_t c = 0; c < comps; ++c) {
            while (p < end && is_blank(a.text[p])) ++p;
            const uint32_t tok = p;
            bool unsure = false;
            float f = 0.0f;
            if (p < end && a.text[p] != '\n') f = parse_f32(a.text, p, end, unsure);
            else atomicOr(a.error, 2u);        // fewer components than the reference's parser insists on: it fails the load
            pool[comps * at + c] = f;
            if (unsure) note_unsure(a, k, comps * at + c, tok);
        }
        return;
    }
    // a face: corners "v", "v/vt", "v//vn", "v/vt/vn"; indices count from 1, negative ones back from the pools' ends so far
    const uint32_t nv = a.v_at[l], nvt = a.vt_at[l], nvn = a.vn_at[l];
    const uint32_t og = a.og_at[l];
    const uint32_t model = og ? a.og_model[og - 1] : 0u;
    uint32_t c = a.corner_at[l];
    auto resolve = [](long idx, uint32_t count) -> long { return idx > 0 ? idx - 1 : idx < 0 ? (long)count + idx : -1; };
    for (;;) {
        while (p < end && is_blank(a.text[p])) ++p;
        if (p >= end || a.text[p] == '\n') break;
        long iv = parse_long(a.text, p, end), it = 0, in = 0;
        if (p < end && a.text[p] == '/') {
            ++p;
            if (p < end && a.text[p] != '/' && !is_blank(a.text[p]) && a.text[p] != '\n') it = parse_long(a.text, p, end);
            if (p < end && a.text[p] == '/') { ++p; if (p < end && !is_blank(a.text[p]) && a.text[p] != '\n') in = parse_long(a.text, p, end); }
        }
        while (p < end && !is_blank(a.text[p]) && a.text[p] != '\n') ++p;
        const long rv = resolve(iv, nv), rt = resolve(it, nvt), rn = resolve(in, nvn);
        if (rv < 0 || rv >= (long)a.nv || rt >= (long)a.nvt || rn >= (long)a.nvn || rt < -1 || rn < -1) atomicOr(a.error, 1u);
        a.ck_v[c] = (int)rv; a.ck_t[c] = rt < 0 ? -1 : (int)rt; a.ck_n[c] = rn < 0 ? -1 : (int)rn; a.ck_model[c] = model;
        ++c;
    }
}

// ---- corners with the same (model, v, vt, vn) become one vertex: the table keeps the first corner of every key
struct CornerKeys { const int *v, *t, *n; const uint32_t* model; };
__device__ inline bool same_key(const CornerKeys& k, uint32_t a, uint32_t b) {
    return k.v[a] == k.v[b] && k.t[a] == k.t[b] && k.n[a] == k.n[b] && k.model[a] == k.model[b];
}
__device__ inline uint32_t key_hash(const CornerKeys& k, uint32_t c) {
    unsigned long long h = (unsigned long long)(uint32_t)k.v[c] * 0x9E3779B97F4A7C15ull;
    h ^= ((unsigned long long)(uint32_t)k.t[c] + 0x7F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    h ^= ((unsigned long long)(uint32_t)k.n[c] + 0x1CE4E5B9ull) * 0x94D049BB133111EBull;
    h ^= (unsigned long long)k.model[c] * 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    return (uint32_t)h;
}
__global__ void __launch_bounds__(256) k_table_insert(CornerKeys k, uint32_t ncorners, uint32_t* table, uint32_t mask, uint32_t* slot_of) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncorners) return;
    uint32_t s = key_hash(k, c) & mask;
    for (;;) {
        uint32_t cur = atomicCAS(&table[s], 0xFFFFFFFFu, c);
        if (cur == 0xFFFFFFFFu) break;                          // the slot is this key's now
        if (same_key(k, cur, c)) { atomicMin(&table[s], c); break; }
        s = (s + 1) & mask;
    }
    slot_of[c] = s;
}
__global__ void __launch_bounds__(256) k_mark_first(const uint32_t* table, const uint32_t* slot_of, uint32_t ncorners, const int* ck_t, const int* ck_n,
                                                     uint32_t nvt, uint32_t nvn, uint32_t* first, uint32_t* first_t, uint32_t* first_n) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncorners) return;
    const bool f = table[slot_of[c]] == c;
    first[c] = f;
    // a unified vertex gets a texture coordinate / a normal only when its corner names one (and the file has any)
    first_t[c] = f && nvt && ck_t[c] >= 0;
    first_n[c] = f && nvn && ck_n[c] >= 0;
}
struct GatherArgs {
    const uint32_t *table, *slot_of, *vertex_at, *tex_at, *normal_at;
    const int *ck_v, *ck_t, *ck_n;
    const float *v, *vt, *vn;
    uint32_t ncorners, nvt, nvn;
    double *positions, *texcoords, *normals;
    uint32_t* corner_vertex;
};
__global__ void __launch_bounds__(256) k_gather(GatherArgs a) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.ncorners) return;
    const uint32_t rep = a.table[a.slot_of[c]];
    a.corner_vertex[c] = a.vertex_at[rep];
    if (rep != c) return;
    const uint32_t u = a.vertex_at[c];
    for (int k = 0; k < 3; ++k) a.positions[3 * (size_t)u + k] = (double)a.v[3 * (size_t)a.ck_v[c] + k];
    if (a.nvt && a.ck_t[c] >= 0) for (int k = 0; k < 2; ++k) a.texcoords[2 * (size_t)a.tex_at[c] + k] = (double)a.vt[2 * (size_t)a.ck_t[c] + k];
    if (a.nvn && a.ck_n[c] >= 0) for (int k = 0; k < 3; ++k) a.normals[3 * (size_t)a.normal_at[c] + k] = (double)a.vn[3 * (size_t)a.ck_n[c] + k];
}
// fan triangulation of every face line (harness/raytracer.cpp, Mesh::from_path; tobj does the same)
__global__ void __launch_bounds__(256) k_faces(const uint8_t* kind, const uint32_t* corner_at, const uint32_t* corners, const uint32_t* tri_at, uint32_t nlines,
                                                const uint32_t* corner_vertex, uint32_t* faces) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlines || kind[l] != L_F) return;
    const uint32_t c0 = corner_at[l], nc = corners[l];
    uint32_t t = tri_at[l];
    for (uint32_t i = 1; i + 1 < nc; ++i, ++t) {
        faces[3 * (size_t)t] = corner_vertex[c0];
        faces[3 * (size_t)t + 1] = corner_vertex[c0 + i];
        faces[3 * (size_t)t + 2] = corner_vertex[c0 + i + 1];
    }
}

template <class T> T* host_copy(const T* d, size_t n) {
    T* h = static_cast<T*>(std::malloc(std::max<size_t>(n, 1) * sizeof(T)));
    if (!h) throw HipFailure{"out of host memory"};
    if (n) { hipError_t e = hipMemcpy(h, d, n * sizeof(T), hipMemcpyDeviceToHost); if (e != hipSuccess) { std::free(h); throw HipFailure{hipGetErrorString(e)}; } }
    return h;
}

int obj_parse(int device, const char* text, uint64_t bytes, RtObjMesh* out) {
    if (!text || !out) return ing_fail(RT_ERR_INVALID_ARG, "rt_obj_parse: null argument");
    std::memset(out, 0, sizeof *out);
    if (bytes == 0 || bytes >= (1ull << 31)) return ing_fail(RT_ERR_INVALID_ARG, "rt_obj_parse: text empty or 2 GiB and more");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return ing_fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return ing_fail(RT_ERR_INVALID_ARG, "device index out of range");
    const uint32_t n = (uint32_t)bytes;
    rtflat::DeviceRestore restore;
    try {
        ING_HIP(hipSetDevice(device));
        Pool pool;
        hipStream_t st = nullptr;
        auto grid = [](uint32_t items) { return dim3((items + 255u) / 256u); };
        char* d_text = pool.get<char>(n);
        ING_HIP(hipMemcpyAsync(d_text, text, n, hipMemcpyHostToDevice, st));
        // lines
        uint32_t* line_index = pool.get<uint32_t>(n);
        hipLaunchKernelGGL(k_line_flags, grid(n), dim3(256), 0, st, d_text, n, line_index);
        const uint32_t nlines = exclusive_scan(pool, line_index, n, st);
        uint32_t* starts = pool.get<uint32_t>(nlines);
        hipLaunchKernelGGL(k_line_starts, grid(n), dim3(256), 0, st, d_text, n, line_index, starts);
        // kinds and counts, then every line's place in the pools
        uint32_t *v_at = pool.get<uint32_t>(nlines), *vt_at = pool.get<uint32_t>(nlines), *vn_at = pool.get<uint32_t>(nlines), *f_at = pool.get<uint32_t>(nlines),
                 *og_at = pool.get<uint32_t>(nlines), *corner_at = pool.get<uint32_t>(nlines), *tri_at = pool.get<uint32_t>(nlines), *corners = pool.get<uint32_t>(nlines);
        uint8_t* kind = pool.get<uint8_t>(nlines);
        hipLaunchKernelGGL(k_classify, grid(nlines), dim3(256), 0, st, d_text, n, starts, nlines, v_at, vt_at, vn_at, f_at, og_at, corner_at, tri_at, kind);
        ING_HIP(hipMemcpyAsync(corners, corner_at, nlines * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        const uint32_t nv = exclusive_scan(pool, v_at, nlines, st), nvt = exclusive_scan(pool, vt_at, nlines, st), nvn = exclusive_scan(pool, vn_at, nlines, st);
        const uint32_t nf = exclusive_scan(pool, f_at, nlines, st), nog = exclusive_scan(pool, og_at, nlines, st);
        const uint32_t ncorners = exclusive_scan(pool, corner_at, nlines, st), ntris = exclusive_scan(pool, tri_at, nlines, st);
        (void)nf;
        if (ntris == 0) return ing_fail(RT_ERR_INVALID_ARG, "rt_obj_parse: the text has no faces");
        // model boundaries
        uint32_t* og_faces_before = pool.get<uint32_t>(nog);
        uint32_t* og_model = pool.get<uint32_t>(nog);
        if (nog) {
            hipLaunchKernelGGL(k_og_effective, grid(nlines), dim3(256), 0, st, starts, kind, og_at, f_at, nlines, og_faces_before);
            hipLaunchKernelGGL(k_og_flags, grid(nog), dim3(256), 0, st, og_faces_before, nog, og_model);
            // inclusive count: effective boundaries up to and including this o/g line = exclusive prefix + own flag
            uint32_t* flags = pool.get<uint32_t>(nog);
            ING_HIP(hipMemcpyAsync(flags, og_model, nog * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            (void)exclusive_scan(pool, og_model, nog, st);
            std::vector<uint32_t> ex(nog), fl(nog);
            ING_HIP(hipMemcpy(ex.data(), og_model, nog * 4, hipMemcpyDeviceToHost));
            ING_HIP(hipMemcpy(fl.data(), flags, nog * 4, hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < nog; ++k) ex[k] += fl[k];
            ING_HIP(hipMemcpy(og_model, ex.data(), nog * 4, hipMemcpyHostToDevice));
        }
        // parse
        ObjArgs a{};
        a.text = d_text; a.n = n; a.starts = starts; a.nlines = nlines; a.kind = kind;
        a.v_at = v_at; a.vt_at = vt_at; a.vn_at = vn_at; a.f_at = f_at; a.og_at = og_at; a.corner_at = corner_at; a.tri_at = tri_at;
        a.og_model = og_model; a.num_og = nog; a.nv = nv; a.nvt = nvt; a.nvn = nvn;
        a.v = pool.get<float>(3 * (size_t)nv); a.vt = pool.get<float>(2 * (size_t)nvt); a.vn = pool.get<float>(3 * (size_t)nvn);
        a.ck_v = pool.get<int>(ncorners); a.ck_t = pool.get<int>(ncorners); a.ck_n = pool.get<int>(ncorners); a.ck_model = pool.get<uint32_t>(ncorners);
        a.unsure_cap = 1u << 16;
        a.unsure = pool.get<uint32_t>(3 * (size_t)a.unsure_cap);
        uint32_t* scalars = pool.get<uint32_t>(2);
        ING_HIP(hipMemsetAsync(scalars, 0, 8, st));
        a.unsure_count = scalars; a.error = scalars + 1;
        hipLaunchKernelGGL(k_parse_lines, grid(nlines), dim3(256), 0, st, a);
        ING_HIP(hipGetLastError());
        uint32_t sc[2];
        ING_HIP(hipMemcpyAsync(sc, scalars, 8, hipMemcpyDeviceToHost, st));
        ING_HIP(hipStreamSynchronize(st));
        if (sc[1] & 2u) return ing_fail(RT_ERR_UNSUPPORTED, "rt_obj_parse: a point or line element, or a v / vt / vn line with too few components");
        if (sc[1]) return ing_fail(RT_ERR_INVALID_ARG, "rt_obj_parse: a face names a vertex, texture coordinate or normal the text does not hold");
        if (sc[0] > a.unsure_cap) return ing_fail(RT_ERR_UNSUPPORTED, "rt_obj_parse: too many literals outside the device parser's range");
        if (sc[0]) {   // literals the device would not decide: the host's strtof rounds them once, like the reference's parser
            std::vector<uint32_t> u(3 * (size_t)sc[0]);
            ING_HIP(hipMemcpy(u.data(), a.unsure, u.size() * 4, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < sc[0]; ++i) {
                const uint32_t off = u[3 * i + 2];
                char tok[64];
                uint32_t len = 0;
                while (off + len < n && len < 63 && text[off + len] != ' ' && text[off + len] != '\t' && text[off + len] != '\r' && text[off + len] != '\n') { tok[len] = text[off + len]; ++len; }
                tok[len] = 0;
                char* stop = nullptr;
                const float f = std::strtof(tok, &stop);
                // (Rust's f32 parser takes the whole token or fails the load; strtof would stop early or read hex)
                if (len == 0 || stop != tok + len || std::strpbrk(tok, "xXpP"))
                    return ing_fail(RT_ERR_UNSUPPORTED, std::string("rt_obj_parse: '") + tok + "' is no decimal number");
                float* dst = (u[3 * i] == L_V ? a.v : u[3 * i] == L_VT ? a.vt : a.vn) + u[3 * i + 1];
                ING_HIP(hipMemcpy(dst, &f, 4, hipMemcpyHostToDevice));
            }
        }
        // unified vertices
        uint32_t table_size = 1024;
        while (table_size < 2 * ncorners) table_size <<= 1;
        uint32_t* table = pool.get<uint32_t>(table_size);
        ING_HIP(hipMemsetAsync(table, 0xFF, (size_t)table_size * 4, st));
        uint32_t* slot_of = pool.get<uint32_t>(ncorners);
        CornerKeys keys{a.ck_v, a.ck_t, a.ck_n, a.ck_model};
        hipLaunchKernelGGL(k_table_insert, grid(ncorners), dim3(256), 0, st, keys, ncorners, table, table_size - 1, slot_of);
        uint32_t *vertex_at = pool.get<uint32_t>(ncorners), *tex_at = pool.get<uint32_t>(ncorners), *normal_at = pool.get<uint32_t>(ncorners);
        hipLaunchKernelGGL(k_mark_first, grid(ncorners), dim3(256), 0, st, table, slot_of, ncorners, a.ck_t, a.ck_n, nvt, nvn, vertex_at, tex_at, normal_at);
        const uint32_t nverts = exclusive_scan(pool, vertex_at, ncorners, st), ntex = exclusive_scan(pool, tex_at, ncorners, st), nnorm = exclusive_scan(pool, normal_at, ncorners, st);
        GatherArgs g{};
        g.table = table; g.slot_of = slot_of; g.vertex_at = vertex_at; g.tex_at = tex_at; g.normal_at = normal_at;
        g.ck_v = a.ck_v; g.ck_t = a.ck_t; g.ck_n = a.ck_n; g.v = a.v; g.vt = a.vt; g.vn = a.vn; g.ncorners = ncorners; g.nvt = nvt; g.nvn = nvn;
        g.positions = pool.get<double>(3 * (size_t)nverts); g.texcoords = pool.get<double>(2 * (size_t)ntex); g.normals = pool.get<double>(3 * (size_t)nnorm);
        g.corner_vertex = pool.get<uint32_t>(ncorners);
        hipLaunchKernelGGL(k_gather, grid(ncorners), dim3(256), 0, st, g);
        uint32_t* faces = pool.get<uint32_t>(3 * (size_t)ntris);
        hipLaunchKernelGGL(k_faces, grid(nlines), dim3(256), 0, st, kind, corner_at, corners, tri_at, nlines, g.corner_vertex, faces);
        ING_HIP(hipGetLastError());
        ING_HIP(hipStreamSynchronize(st));
        out->positions = host_copy(g.positions, 3 * (size_t)nverts);
        out->normals = host_copy(g.normals, 3 * (size_t)nnorm);
        out->texcoords = host_copy(g.texcoords, 2 * (size_t)ntex);
        out->faces = host_copy(faces, 3 * (size_t)ntris);
        out->num_positions = nverts; out->num_normals = nnorm; out->num_texcoords = ntex; out->num_faces = ntris;
        return RT_OK;
    } catch (const HipFailure& e) {
        std::free(out->positions); std::free(out->normals); std::free(out->texcoords); std::free(out->faces);
        std::memset(out, 0, sizeof *out);
        return ing_fail(RT_ERR_DEVICE, "rt_obj_parse: " + e.what);
    } catch (const std::exception& e) {                               // (std::bad_alloc of the host-side vectors: nothing may unwind through the C ABI)
        std::free(out->positions); std::free(out->normals); std::free(out->texcoords); std::free(out->faces);
        std::memset(out, 0, sizeof *out);
        return ing_fail(RT_ERR_DEVICE, std::string("rt_obj_parse: ") + e.what());
    }
}

}  // namespace

// ---- used by rt_api.hip
rtflat::MeshBuilder rt_device_mesh_builder(int device) {
    return [device](const double* P, uint32_t npos, const uint32_t* F, uint32_t nfaces, int max_depth,
                    std::vector<uint32_t>& face_order, std::vector<rtflat::BuildNode>& bn) { device_build(device, P, npos, F, nfaces, max_depth, face_order, bn); };
}
// Flips the rows of an uploaded image in place (texels: device memory).
hipError_t rt_device_flip_rows(uint8_t* texels, uint32_t row_bytes, uint32_t height, hipStream_t stream) {
    if (height < 2 || row_bytes == 0) return hipSuccess;
    hipLaunchKernelGGL(k_flip_rows, dim3(height / 2), dim3(256), 0, stream, texels, row_bytes, height);
    return hipGetLastError();
}

extern "C" {

int rt_obj_parse(int device, const char* text, uint64_t bytes, RtObjMesh* out) { return obj_parse(device, text, bytes, out); }

void rt_obj_release(RtObjMesh* mesh) {
    if (!mesh) return;
    std::free(mesh->positions); std::free(mesh->normals); std::free(mesh->texcoords); std::free(mesh->faces);
    std::memset(mesh, 0, sizeof *mesh);
}

}  // extern "C"
