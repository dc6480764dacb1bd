// rt_math.h — f64 vector helpers and the counter-based RNG for the device path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device_types.h"

#define RT_DEV __device__ __forceinline__

struct d3 {
    double x, y, z;
};
RT_DEV d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV d3 ld3(const double* p) { return mk3(p[0], p[1], p[2]); }
RT_DEV d3 ld3(const float* p) { return mk3((double)p[0], (double)p[1], (double)p[2]); }
RT_DEV void st3(double* p, d3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
RT_DEV d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV d3 operator*(d3 a, double s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV d3 operator*(double s, d3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_DEV d3 operator*(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }   // colours
RT_DEV d3 operator/(d3 a, double s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_DEV double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV d3 cross(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV double length(d3 a) { return sqrt(dot(a, a)); }
RT_DEV d3 normalize(d3 a) { return a * (1.0 / length(a)); }   // cgmath: v * (1/|v|)
RT_DEV d3 neg(d3 a) { return mk3(-a.x, -a.y, -a.z); }

RT_DEV d3 xf_point(const DAffine& m, d3 p) {
    return mk3(((m.r[0] * p.x + m.r[1] * p.y) + m.r[2] * p.z) + m.r[3],
               ((m.r[4] * p.x + m.r[5] * p.y) + m.r[6] * p.z) + m.r[7],
               ((m.r[8] * p.x + m.r[9] * p.y) + m.r[10] * p.z) + m.r[11]);
}
RT_DEV d3 xf_vector(const DAffine& m, d3 v) {
    return mk3((m.r[0] * v.x + m.r[1] * v.y) + m.r[2] * v.z,
               (m.r[4] * v.x + m.r[5] * v.y) + m.r[6] * v.z,
               (m.r[8] * v.x + m.r[9] * v.y) + m.r[10] * v.z);
}
RT_DEV d3 xf_normal(const double* n9, d3 n) {
    return normalize(mk3((n9[0] * n.x + n9[1] * n.y) + n9[2] * n.z,
                         (n9[3] * n.x + n9[4] * n.y) + n9[5] * n.z,
                         (n9[6] * n.x + n9[7] * n.y) + n9[8] * n.z));
}

RT_DEV double fract_(double x) { return x - trunc(x); }   // Rust f64::fract

// SurfaceCoord::new (src/geometry/surface_coord.rs:9-19)
RT_DEV void surface_coord(double u, double v, double& ou, double& ov) {
    u = fract_(u); v = fract_(v);
    ou = u < 0.0 ? 1.0 + u : u;
    ov = v < 0.0 ? 1.0 + v : v;
}

// ---- counter-based RNG: the specification both the device path and the CPU checker follow ----
RT_DEV uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RT_DEV double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
RT_DEV uint64_t rng_stream(uint64_t seed, uint64_t pixel, uint32_t sample) {
    uint64_t h = sm64(seed);
    h = sm64(h ^ pixel);
    return sm64(h ^ (uint64_t)sample);
}
RT_DEV uint32_t hash32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
RT_DEV uint32_t child_path(uint32_t parent, uint32_t ordinal) { return hash32(parent ^ hash32(ordinal + 0x9E3779B9u)); }
RT_DEV double rng_light(uint64_t stream, uint32_t path, uint32_t light, uint32_t counter) {
    return u01(sm64(stream ^ (((uint64_t)path << 32) | ((uint64_t)(light & 0x3FFu) << 22) | (uint64_t)(counter & 0x3FFFFFu))));
}
// One draw that yields two uniforms (its upper and lower 32 bits): the pair of coordinates of a
// light sample that is taken without a shadow test.
RT_DEV void rng_light_pair(uint64_t stream, uint32_t path, uint32_t light, uint32_t counter, double& u, double& v) {
    const uint64_t h = sm64(stream ^ (((uint64_t)path << 32) | ((uint64_t)(light & 0x3FFu) << 22) | (uint64_t)(counter & 0x3FFFFFu)));
    u = (double)(uint32_t)(h >> 32) * (1.0 / 4294967296.0);
    v = (double)(uint32_t)h * (1.0 / 4294967296.0);
}
RT_DEV double rng_aa(uint64_t stream0, uint32_t index) { return u01(sm64(stream0 ^ (0xA5A5A5A500000000ull + index))); }

// ---- IEEE division and square root without the range scaling of the general sequences ----
// x / d is what the compiler's f64 division computes (reciprocal estimate, two Newton steps,
// quotient, one correction with the exact remainder) when neither operand needs the exponent
// scaling of v_div_scale, i.e. for d and x well inside the normal range; the result is then the
// correctly rounded quotient, bit for bit the same as `x / d`.  Splitting it lets several
// quotients by one divisor share the reciprocal.  Callers check the divisor's range
// (nr_in_range) and take the general path otherwise.
RT_DEV double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}
RT_DEV double div_nr(double x, double d, double rcp_d) {
    const double q = x * rcp_d;
    const double rem = __builtin_fma(-d, q, x);
    return __builtin_fma(rem, rcp_d, q);
}
// sqrt(x) as the compiler's f64 sequence computes it for x in the normal range (no scaling of
// tiny arguments, no zero / infinity pass-through).
RT_DEV double sqrt_nr(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
RT_DEV bool nr_in_range(double x) { return (int)(fabs(x) > 0x1p-200) & (int)(fabs(x) < 0x1p200); }
