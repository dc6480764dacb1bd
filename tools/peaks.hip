// peaks.hip — measured roofline denominators of this GPU, the numbers the render path's fractions are
// quoted against: FP64 vector FMA rate, FP64 un-fused multiply + add rate (what a build with
// -ffp-contract=off can issue), packed-FP32 FMA rate, and HBM read / write / copy bandwidth.
// Every kernel fills the whole chip (many workgroups per CU); rates are the best of several timed launches.
//   hipcc --offload-arch=gfx950 -O3 -o peaks peaks.hip && ./peaks        (prints one JSON object)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int ITERS = 2048, CHAINS = 8;

// OP 0: f64 FMA; 1: f64 mul then add (two instructions, two flops); 2: packed f32 FMA (two lanes' worth per instruction)
template <int OP>
__global__ void __launch_bounds__(256) k_flops(double* out, double seed) {
    double a[CHAINS];
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p[CHAINS];
    const double b = seed * 1.0000001, c = seed * 0.25;
    const float2v pb = {(float)b, (float)b}, pc = {(float)c, (float)c};
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x * 1e-9; p[i] = float2v{(float)a[i], (float)a[i] + 1.0f}; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
            if (OP == 1) { a[i] = a[i] * b; asm volatile("" : "+v"(a[i])); a[i] = a[i] + c; }
            if (OP == 2) p[i] = __builtin_elementwise_fma(p[i], pb, pc);
        }
    }
    double s = 0.0;
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (double)p[i].x + (double)p[i].y;
    if (s == 0.123456) out[0] = s;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ src, size_t n, unsigned long long* out) {
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(u32x4* __restrict__ dst, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(u32x4{v, v + 1u, v + 2u, v + 3u}, dst + i);
}
__global__ void __launch_bounds__(256) k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

template <class L>
double best_ms(L launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    return best;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double* out = nullptr;
    CHECK(hipMalloc(&out, 64));
    const uint32_t grid = (uint32_t)cus * 16u;
    const double threads = (double)grid * 256.0;
    auto tf = [&](double flops_per_thread, double ms) { return threads * flops_per_thread / (ms * 1e-3) / 1e12; };
    const double ms_fma = best_ms([&] { hipLaunchKernelGGL(k_flops<0>, dim3(grid), dim3(256), 0, nullptr, out, 1.0); }, 5);
    const double ms_ma = best_ms([&] { hipLaunchKernelGGL(k_flops<1>, dim3(grid), dim3(256), 0, nullptr, out, 1.0); }, 5);
    const double ms_pk = best_ms([&] { hipLaunchKernelGGL(k_flops<2>, dim3(grid), dim3(256), 0, nullptr, out, 1.0); }, 5);
    const size_t bytes = (size_t)4 << 30;                          // 4 GiB: sixteen times the 256 MB Infinity Cache
    u32x4 *a = nullptr, *b = nullptr;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    const size_t n = bytes / sizeof(u32x4);
    const uint32_t mgrid = (uint32_t)cus * 32u;
    const double ms_rd = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(mgrid), dim3(256), 0, nullptr, a, n, (unsigned long long*)out); }, 5);
    const double ms_wr = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(mgrid), dim3(256), 0, nullptr, b, n, 7u); }, 5);
    const double ms_cp = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(mgrid), dim3(256), 0, nullptr, a, b, n); }, 5);
    const double gb = (double)bytes / 1e9;
    std::printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, "
                "\"fp64_fma_tflops\": %.2f, \"fp64_mul_add_tflops\": %.2f, \"fp32_packed_fma_tflops\": %.2f, "
                "\"hbm_read_gbs\": %.0f, \"hbm_write_gbs\": %.0f, \"hbm_copy_gbs\": %.0f, "
                "\"note\": \"copy counts bytes read + bytes written; 4 GiB per stream, best of 5 launches\"}\n",
                p.name, p.gcnArchName, cus, p.clockRate / 1000,
                tf(2.0 * ITERS * CHAINS, ms_fma), tf(2.0 * ITERS * CHAINS, ms_ma), tf(4.0 * ITERS * CHAINS, ms_pk),
                gb / (ms_rd * 1e-3), gb / (ms_wr * 1e-3), 2.0 * gb / (ms_cp * 1e-3));
    return 0;
}
