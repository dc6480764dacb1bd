// rt_output.hip — the stage after the render path: the frame as a PNG file, and progress reports.
//
// Reference: write_to_png (src/lib.rs:41-46: RgbImage::save to "<file_name>.png", a line on stdout
// either way, no error returned) and ProgressTracker (src/progress_tracker.rs:48-77: "START
// RENDERING", a running percentage, "START ANTIALIASING: n / total = p%", "Elapsed Time").
//
// A PNG is the image's scanlines, each behind one filter byte, deflated.  Choosing the filter
// (per row, the one of the five whose output has the smallest sum of absolute values, the usual
// heuristic) and applying it is byte arithmetic over the whole frame and runs on the device,
// where the frame already is: one workgroup per row reads the row and the one above once.  The
// host deflates the filtered scanlines (zlib) and frames the chunks.  Any decoder gives back the
// frame's bytes exactly; the file's own bytes are not the reference's (its encoder picks other
// filters), which no reader can observe.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "rt_flatten.hpp"   // DeviceRestore

#include "rt_amd.h"

int rt_set_error(int code, const std::string& msg);   // rt_api.hip: the message rt_last_error returns on this thread

namespace {

int out_fail(int code, const std::string& msg) { return rt_set_error(code, msg); }

#define OUT_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return out_fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

__device__ inline int paeth(int a, int b, int c) {       // PNG specification, section 9.4
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}
__device__ inline int filtered(int f, int x, int a, int b, int c) {
    switch (f) {
        case 0: return x;
        case 1: return x - a;
        case 2: return x - b;
        case 3: return x - ((a + b) >> 1);
        default: return x - paeth(a, b, c);
    }
}

// Row y of `rgb` (3*width bytes per row, rows top first) -> out[y * (1 + 3*width)]: filter byte + filtered bytes.
__global__ void __launch_bounds__(256)
k_png_filter(const uint8_t* __restrict__ rgb, uint32_t width, uint32_t height, uint8_t* __restrict__ out) {
    const uint32_t y = blockIdx.x;
    if (y >= height) return;
    const uint32_t n = 3u * width;
    const uint8_t* cur = rgb + (size_t)y * n;
    const uint8_t* prev = y ? cur - n : nullptr;
    unsigned int sum[5] = {0, 0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int x = cur[i], a = i >= 3 ? cur[i - 3] : 0, b = prev ? prev[i] : 0, c = (prev && i >= 3) ? prev[i - 3] : 0;
        for (int f = 0; f < 5; ++f) sum[f] += (unsigned int)abs((int)(int8_t)(uint8_t)filtered(f, x, a, b, c));
    }
    __shared__ unsigned int total[5];
    __shared__ int choice;
    if (threadIdx.x < 5) total[threadIdx.x] = 0;
    __syncthreads();
    for (int f = 0; f < 5; ++f) {
        unsigned int v = sum[f];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&total[f], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int best = 0;
        for (int f = 1; f < 5; ++f) if (total[f] < total[best]) best = f;
        choice = best;
    }
    __syncthreads();
    const int f = choice;
    uint8_t* o = out + (size_t)y * (n + 1u);
    if (threadIdx.x == 0) o[0] = (uint8_t)f;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int x = cur[i], a = i >= 3 ? cur[i - 3] : 0, b = prev ? prev[i] : 0, c = (prev && i >= 3) ? prev[i - 3] : 0;
        o[1 + i] = (uint8_t)filtered(f, x, a, b, c);
    }
}

void be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(std::vector<uint8_t>& png, const char type[4], const uint8_t* data, size_t n) {
    be32(png, (uint32_t)n);
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if (n) png.insert(png.end(), data, data + n);
    be32(png, (uint32_t)crc32(0L, png.data() + at, (uInt)(n + 4)));
}

// The zlib stream of `data`, deflated in bands by several host threads: every band is a run of raw deflate blocks
// of its own (no references into the band before it, closed on a byte boundary by a sync flush; the last band
// ends the stream), the bands follow one another behind one zlib header, and the Adler-32 of the whole is put
// together from the bands' own.  Any inflater reads this as one ordinary stream; it is a fraction of a percent
// longer than a single-threaded one.
bool deflate_bands(const std::vector<uint8_t>& data, std::vector<uint8_t>& out) {
    const size_t n = data.size();
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("OMP_NUM_THREADS")) { const long v = std::atol(e); if (v > 0) hw = std::min<unsigned>(hw ? hw : (unsigned)v, (unsigned)v); }
    const size_t bands = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1u, 16u), n / (256u << 10)));   // at least 256 KiB each
    std::vector<std::vector<uint8_t>> part(bands);
    std::vector<uLong> adler(bands), length(bands);
    std::vector<int> ok(bands, 0);
    auto work = [&](size_t b) {
        const size_t from = n * b / bands, to = n * (b + 1) / bands;
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
        part[b].resize(deflateBound(&zs, (uLong)(to - from)) + 16);
        zs.next_in = const_cast<Bytef*>(data.data() + from); zs.avail_in = (uInt)(to - from);
        zs.next_out = part[b].data(); zs.avail_out = (uInt)part[b].size();
        const int rc = deflate(&zs, b + 1 == bands ? Z_FINISH : Z_SYNC_FLUSH);
        const bool done = b + 1 == bands ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0);
        part[b].resize(zs.total_out);
        deflateEnd(&zs);
        adler[b] = adler32(adler32(0L, Z_NULL, 0), data.data() + from, (uInt)(to - from));
        length[b] = (uLong)(to - from);
        ok[b] = done ? 1 : 0;
    };
    std::vector<std::thread> threads;
    for (size_t b = 1; b < bands; ++b) threads.emplace_back(work, b);
    work(0);
    for (auto& t : threads) t.join();
    out.clear();
    out.push_back(0x78); out.push_back(0x9C);                     // zlib header: deflate, 32 KiB window, default level
    uLong sum = adler32(0L, Z_NULL, 0);
    for (size_t b = 0; b < bands; ++b) {
        if (!ok[b]) return false;
        out.insert(out.end(), part[b].begin(), part[b].end());
        sum = adler32_combine(sum, adler[b], (z_off_t)length[b]);
    }
    be32(out, (uint32_t)sum);
    return true;
}

// d_rgb: device memory.  Filters on the device, deflates and writes on the host.
int write_png_device(int device, const uint8_t* d_rgb, uint32_t width, uint32_t height, const char* file_name, hipStream_t stream) {
    if (!d_rgb || !file_name) return out_fail(RT_ERR_INVALID_ARG, "rt_write_png: null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return out_fail(RT_ERR_INVALID_ARG, "rt_write_png: bad image size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return out_fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return out_fail(RT_ERR_INVALID_ARG, "device index out of range");
    OUT_TRY(hipSetDevice(device));
    const size_t line = 3u * (size_t)width + 1u, bytes = line * height;
    uint8_t* d_lines = nullptr;
    OUT_TRY(hipMalloc((void**)&d_lines, bytes));
    hipLaunchKernelGGL(k_png_filter, dim3(height), dim3(256), 0, stream, d_rgb, width, height, d_lines);
    std::vector<uint8_t> lines(bytes);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(lines.data(), d_lines, bytes, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_lines);
    if (e != hipSuccess) return out_fail(RT_ERR_DEVICE, std::string("PNG filter pass: ") + hipGetErrorString(e));
    std::vector<uint8_t> z;
    if (!deflate_bands(lines, z)) return out_fail(RT_ERR_DEVICE, "rt_write_png: deflate failed");
    const size_t zcap = z.size();
    std::vector<uint8_t> png = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, width); be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8 bits, RGB, deflate, adaptive filtering, no interlace
    chunk(png, "IHDR", ihdr.data(), ihdr.size());
    chunk(png, "IDAT", z.data(), zcap);
    chunk(png, "IEND", nullptr, 0);
    const std::string path = std::string(file_name) + ".png";                // lib.rs:42
    FILE* f = std::fopen(path.c_str(), "wb");
    bool ok = f != nullptr;
    if (f) { ok = std::fwrite(png.data(), 1, png.size(), f) == png.size(); ok = (std::fclose(f) == 0) && ok; }
    if (!ok) return out_fail(RT_ERR_INVALID_ARG, "rt_write_png: cannot write " + path);
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_write_png_device(int device, const uint8_t* d_rgb, uint32_t width, uint32_t height, const char* file_name, void* hip_stream) {
    rtflat::DeviceRestore restore;
    try {
        return write_png_device(device, d_rgb, width, height, file_name, static_cast<hipStream_t>(hip_stream));
    } catch (const std::exception& e) {                                   // (std::bad_alloc: nothing may unwind through the C ABI)
        return out_fail(RT_ERR_DEVICE, std::string("rt_write_png_device: ") + e.what());
    }
}

int rt_write_png(int device, const uint8_t* rgb, uint32_t width, uint32_t height, const char* file_name) {
    if (!rgb || !file_name) return out_fail(RT_ERR_INVALID_ARG, "rt_write_png: null argument");
    if (width == 0 || height == 0 || (uint64_t)width * height > (1ull << 28)) return out_fail(RT_ERR_INVALID_ARG, "rt_write_png: bad image size");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return out_fail(RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return out_fail(RT_ERR_INVALID_ARG, "device index out of range");
    rtflat::DeviceRestore restore;
    OUT_TRY(hipSetDevice(device));
    const size_t bytes = 3u * (size_t)width * height;
    uint8_t* d = nullptr;
    OUT_TRY(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemcpy(d, rgb, bytes, hipMemcpyHostToDevice);
    int rc;
    try {
        rc = e == hipSuccess ? write_png_device(device, d, width, height, file_name, nullptr)
                             : out_fail(RT_ERR_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    } catch (const std::exception& ex) {
        rc = out_fail(RT_ERR_DEVICE, std::string("rt_write_png: ") + ex.what());
    }
    (void)hipFree(d);
    return rc;
}

}  // extern "C"
