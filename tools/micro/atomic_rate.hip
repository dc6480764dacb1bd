// Micro-benchmark: rate of 64-bit integer atomic adds to pseudo-random words of a table,
// the access pattern of per-sample fixed-point colour accumulators.
// Usage: atomic_rate <table-words> <adds-per-thread> <threads>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned long long* t, unsigned int words, int per, int group) {
    unsigned int id = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long h = (id / group) * 0x9E3779B97F4A7C15ull + 12345;
    for (int i = 0; i < per; ++i) {
        h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31;
        unsigned int slot = (unsigned int)(h % (words / 9));
        unsigned int limb = (threadIdx.x % group) % 9;
        atomicAdd(t + (size_t)slot * 9 + limb, h | 1ull);
    }
}
int main(int argc, char** argv) {
    unsigned int words = argc > 1 ? atoi(argv[1]) : 18662400;
    int per = argc > 2 ? atoi(argv[2]) : 64;
    int threads = argc > 3 ? atoi(argv[3]) : (1 << 22);
    int group = argc > 4 ? atoi(argv[4]) : 1;
    unsigned long long* t;
    hipMalloc(&t, (size_t)words * 8);
    hipMemset(t, 0, (size_t)words * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<threads / 256, 256>>>(t, words, 1, group);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<threads / 256, 256>>>(t, words, per, group);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)threads * per;
    printf("words %u adds %.3g group %d: %.2f ms, %.1f G atomics/s\n", words, n, group, ms, n / ms / 1e6);
    return 0;
}
