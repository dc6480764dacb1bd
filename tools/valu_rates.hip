// valu_rates.hip — issue cost of the f64 / integer VALU instructions the render kernels are made of,
// measured on the device: cycles per wave instruction with one and with two waves per SIMD, for a
// dependent chain and for four independent chains.  hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096
#define REP 16

template <int OP, int CHAINS>
__global__ void __launch_bounds__(64) k(double* out, double seed, uint32_t iseed) {
    double a[CHAINS], b = seed * 1.0000001, c = seed * 0.25;
    uint32_t u[CHAINS];
    uint64_t w[CHAINS];
    for (int i = 0; i < CHAINS; ++i) { a[i] = seed + i + threadIdx.x * 1e-9; u[i] = iseed + i + threadIdx.x; w[i] = ((uint64_t)u[i] << 20) + 12345u; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) a[i] = __builtin_fma(a[i], b, c);
                if (OP == 1) a[i] = a[i] * b;
                if (OP == 2) a[i] = a[i] + b;
                if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
                if (OP == 4) a[i] = __builtin_amdgcn_rsq(a[i]);
                if (OP == 5) u[i] = u[i] * 0x9E3779B9u + 1u;                       // v_mul_lo_u32 (+ add)
                if (OP == 6) w[i] = w[i] * 0xBF58476D1CE4E5B9ull;                   // 64 x 64 -> 64
                if (OP == 7) a[i] = trunc(a[i]) + b;                                // v_trunc_f64 + add
                if (OP == 8) a[i] = __builtin_amdgcn_ldexp(a[i], 1);
                if (OP == 9) a[i] = fmax(a[i], b);
                if (OP == 10) a[i] = (double)u[i] + a[i], u[i] += 1;               // cvt u32->f64, add, iadd
                if (OP == 11) { float f = (float)a[i]; f = __builtin_fmaf(f, 1.0000001f, 0.25f); a[i] = (double)f; }   // cvt, fma f32, cvt
                if (OP == 12) a[i] = a[i] / b;                                      // full division sequence
                if (OP == 13) a[i] = sqrt(a[i] + 2.0);                              // full sqrt sequence (+ add)
            }
        }
    }
    double s = 0.0;
    for (int i = 0; i < CHAINS; ++i) s += a[i] + (double)u[i] + (double)w[i];
    if (s == 1.2345) out[threadIdx.x] = s;
}

// Shader clock under sustained load: clock64() counts shader cycles, wall_clock64() a constant 100 MHz.
__global__ void __launch_bounds__(64) k_clock(long long* out, double seed, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double a = seed + threadIdx.x * 1e-9, b = seed * 1.0000001, c = seed * 0.25, a2 = a + 1.0, a3 = a + 2.0, a4 = a + 3.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); a4 = __builtin_fma(a4, b, c); }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (a + a2 + a3 + a4 == 1.2345) out[2] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

template <int OP, int CHAINS>
double run(int waves_per_simd, double* d, double clock_ghz, int cus) {
    const int grid = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(grid), dim3(64), 0, 0, d, 1.000001, 7u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, CHAINS>), dim3(grid), dim3(64), 0, 0, d, 1.000001, 7u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // cycles per (source-level) operation per SIMD
    const double ops_per_simd = (double)ITERS * REP * CHAINS * waves_per_simd;
    return ms * 1e-3 * clock_ghz * 1e9 / ops_per_simd;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    const int cus = p.multiProcessorCount;
    double* d;
    hipMalloc(&d, 4096);
    printf("device %s, %d CUs, %.2f GHz nominal; cycles per operation per SIMD (nominal clock)\n", p.name, cus, ghz);
    printf("%-28s %10s %10s %10s %10s\n", "operation", "1w dep", "1w 4ind", "2w dep", "2w 4ind");
    const char* names[] = {"fma f64", "mul f64", "add f64", "rcp f64", "rsq f64", "mul_lo u32 + add", "mul 64x64->64", "trunc f64 + add",
                           "ldexp f64", "max f64", "cvt u32->f64 + add + iadd", "cvt f64->f32, fma f32, cvt", "a / b (f64 sequence)", "sqrt(a + 2) (f64 sequence)"};
#define ROW(OP) printf("%-28s %10.2f %10.2f %10.2f %10.2f\n", names[OP], run<OP, 1>(1, d, ghz, cus), run<OP, 4>(1, d, ghz, cus), run<OP, 1>(2, d, ghz, cus), run<OP, 4>(2, d, ghz, cus));
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13)
    long long* dc;
    hipMalloc(&dc, 64);
    for (int rep = 0; rep < 3; ++rep) {
        for (int iters : {20000, 200000, 1000000}) {
            hipLaunchKernelGGL(k_clock, dim3(cus * 8), dim3(64), 0, 0, dc, 1.000001, iters);
            long long h[2];
            hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
            printf("sustained f64 fma, 2 waves/SIMD, %8.2f ms: clock64/wall_clock64 = %.3f -> shader clock %.2f GHz if wall clock is 100 MHz; %.2f cycles per fma\n",
                   h[1] / 1e5, (double)h[0] / (double)h[1], (double)h[0] / (double)h[1] * 0.1, (double)h[0] / ((double)iters * 64.0 * 2.0));
        }
    }
    hipFree(dc);
    hipFree(d);
    return 0;
}
