// hbm_calib.hip — known byte counts in the access patterns of the render kernels, to calibrate the
// FETCH_SIZE / WRITE_SIZE counters of rocprofv3 on this GPU: 1 GiB streamed in by 8-byte, 4-byte and
// 16-byte loads per lane, 1 GiB streamed out by 8-byte and 4-byte stores per lane, and 64-bit atomic adds
// (one per lane, all to different words).  hipcc --offload-arch=gfx950 -O3 -o hbm_calib hbm_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T>
__global__ void __launch_bounds__(256) k_read(const T* __restrict__ src, size_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const T v = src[i];
        const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
        acc += b[0];
    }
    if (acc == 0x123456789ull) out[0] = acc;
}
template <class T>
__global__ void __launch_bounds__(256) k_write(T* __restrict__ dst, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void __launch_bounds__(256) k_atomic64(unsigned long long* dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(&dst[i], 3ull);
}

int main() {
    const size_t bytes = 1ull << 30;
    void* buf; unsigned long long* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&out, 64) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16;
    hipLaunchKernelGGL(k_read<double>, dim3(grid), dim3(256), 0, 0, (const double*)buf, bytes / 8, out);
    hipLaunchKernelGGL(k_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_read<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_write<double>, dim3(grid), dim3(256), 0, 0, (double*)buf, bytes / 8, 1.5);
    hipLaunchKernelGGL(k_write<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, bytes / 4, 7u);
    hipLaunchKernelGGL(k_atomic64, dim3(grid), dim3(256), 0, 0, (unsigned long long*)buf, bytes / 8);
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes one way (the atomics: read and write)\n", bytes);
    return 0;
}
