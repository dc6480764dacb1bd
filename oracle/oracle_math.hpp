// oracle_math.hpp — f64 vector / matrix / colour arithmetic of the CPU oracle.
//
// TEST INFRASTRUCTURE ONLY.  Nothing outside tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may call into oracle/.
//
// The reference does this arithmetic through `euler` 0.4.0 → `cgmath` 0.16.1
// (Cargo.lock), which are not vendored under the reference tree.  The operations are
// restated from cgmath's published definitions, in its evaluation order:
//   dot(a,b)        = a.x*b.x + a.y*b.y + a.z*b.z            (left to right)
//   normalize(v)    = v * (1 / sqrt(dot(v,v)))
//   M * v           = M.col0*v.x + M.col1*v.y + M.col2*v.z + M.col3*v.w
//   inverse(M)      = cofactor matrix * (1/det), transposed
// No reference test pins results at this boundary: agreement with cgmath is to a few
// ulp, not bit-exact (parity unpinned at this boundary; pinned end-to-end by the
// gallery image the reference ships).
#pragma once
#include <cmath>
#include <cstdint>

namespace orc {

struct V3 {
    double x, y, z;
};
inline V3 v3(double x, double y, double z) { return {x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(double s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / length(a)); }
inline bool operator==(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

struct V2 {
    double x, y;
};
inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, double s) { return {a.x * s, a.y * s}; }

// Column-major: c[col][row].
struct M4 {
    double c[4][4];
};
inline M4 m4_identity() {
    M4 m{};
    for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0;
    return m;
}
inline M4 m4_from(const double* p) {
    M4 m;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) m.c[c][r] = p[c * 4 + r];
    return m;
}
inline M4 m4_mul(const M4& a, const M4& b) {
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
            r.c[c][row] = ((a.c[0][row] * b.c[c][0] + a.c[1][row] * b.c[c][1]) + a.c[2][row] * b.c[c][2]) + a.c[3][row] * b.c[c][3];
    return r;
}
inline M4 m4_transpose(const M4& a) {
    M4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) r.c[c][row] = a.c[row][c];
    return r;
}
inline double det3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    return a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
}
// Inverse by cofactors (DMat4::inverse, used at src/geometry.rs:92,100 and
// src/geometry/matrix.rs:56).
inline M4 m4_inverse(const M4& m) {
    // work on a[row][col]
    double a[4][4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) a[r][c] = m.c[c][r];
    double cof[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s[3][3];
            int rr = 0;
            for (int i = 0; i < 4; ++i) {
                if (i == r) continue;
                int cc = 0;
                for (int j = 0; j < 4; ++j) {
                    if (j == c) continue;
                    s[rr][cc++] = a[i][j];
                }
                ++rr;
            }
            double d = det3(s[0][0], s[0][1], s[0][2], s[1][0], s[1][1], s[1][2], s[2][0], s[2][1], s[2][2]);
            cof[r][c] = ((r + c) & 1) ? -d : d;
        }
    double det = a[0][0] * cof[0][0] + a[0][1] * cof[0][1] + a[0][2] * cof[0][2] + a[0][3] * cof[0][3];
    double inv_det = 1.0 / det;
    M4 out;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.c[c][r] = cof[c][r] * inv_det;   // inverse[r][c] = cof[c][r]/det
    return out;
}
struct V4 {
    double x, y, z, w;
};
inline V4 m4_mul_v4(const M4& m, V4 v) {
    V4 r;
    r.x = ((m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z) + m.c[3][0] * v.w;
    r.y = ((m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z) + m.c[3][1] * v.w;
    r.z = ((m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z) + m.c[3][2] * v.w;
    r.w = ((m.c[0][3] * v.x + m.c[1][3] * v.y) + m.c[2][3] * v.z) + m.c[3][3] * v.w;
    return r;
}
// src/geometry/matrix.rs:47-53
inline V3 transform_point(const M4& m, V3 p) {
    V4 r = m4_mul_v4(m, {p.x, p.y, p.z, 1.0});
    return {r.x, r.y, r.z};
}
inline V3 transform_vector(const M4& m, V3 v) {
    V4 r = m4_mul_v4(m, {v.x, v.y, v.z, 0.0});
    return {r.x, r.y, r.z};
}

inline double to_radians(double deg) { return deg * (3.14159265358979323846 / 180.0); }
// src/geometry/matrix.rs:63-97 — the builders the oracle needs for Cube / Tetrahedron faces.
inline M4 rotation_x(double deg) {
    double t = to_radians(deg), s = std::sin(t), c = std::cos(t);
    const double p[16] = {1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1};
    return m4_from(p);
}
inline M4 rotation_y(double deg) {
    double t = to_radians(deg), s = std::sin(t), c = std::cos(t);
    const double p[16] = {c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1};
    return m4_from(p);
}
inline M4 translation(double x, double y, double z) {
    const double p[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1};
    return m4_from(p);
}
inline M4 scaling(double x, double y, double z) {
    const double p[16] = {x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1};
    return m4_from(p);
}

// ---- Color (src/color.rs, src/color/operator.rs) ----
struct Color {
    double r, g, b;
};
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator*(Color a, double s) { return {a.r * s, a.g * s, a.b * s}; }
inline Color operator*(double s, Color a) { return {a.r * s, a.g * s, a.b * s}; }
inline Color operator/(Color a, double s) { return {a.r / s, a.g / s, a.b / s}; }
inline Color operator+(Color a, Color b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Color operator-(Color a, Color b) { return {a.r - b.r, a.g - b.g, a.b - b.b}; }
inline Color& operator+=(Color& a, Color b) { a = a + b; return a; }
const Color BLACK{0, 0, 0}, WHITE{1, 1, 1};
// src/color.rs:42-46
inline double color_diff(Color a, Color b) { return std::fabs(a.r - b.r) + std::fabs(a.b - b.b) + std::fabs(a.g - b.g); }
// src/color.rs:50-56: x.min(1.0).max(0.0)
inline Color color_clamp(Color a) {
    return {std::fmax(std::fmin(a.r, 1.0), 0.0), std::fmax(std::fmin(a.g, 1.0), 0.0), std::fmax(std::fmin(a.b, 1.0), 0.0)};
}
// src/color.rs:10-16, 69-75: powf(1/2.2) * 255 truncated by `as u8` (saturating cast)
inline uint8_t encode_channel(double linear) {
    double v = std::pow(linear, 1.0 / 2.2) * 255.0;
    if (!(v > 0.0)) return 0;     // NaN and negatives → 0
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

// ---- counter-based RNG shared (by specification) with the device path ----
// thread_rng is unseedable (src/render.rs:257, src/light.rs:191,228); both sides
// replace it by the same stateless generator so that the sample pattern matches.
inline uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }
// Stream key of one camera sample: sample 0 is the pass-1 ray, 1..aa_rays the AA rays.
inline uint64_t rng_stream(uint64_t seed, uint64_t pixel, uint32_t sample) {
    uint64_t h = sm64(seed);
    h = sm64(h ^ pixel);
    return sm64(h ^ (uint64_t)sample);
}
inline uint32_t hash32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
// Identity of a secondary ray inside one camera sample's ray tree: independent of
// the order in which an implementation visits the tree.
inline uint32_t child_path(uint32_t parent, uint32_t ordinal) { return hash32(parent ^ hash32(ordinal + 0x9E3779B9u)); }
// draw #counter of light #light at the hit reached by ray `path`
inline double rng_light(uint64_t stream, uint32_t path, uint32_t light, uint32_t counter) {
    return u01(sm64(stream ^ (((uint64_t)path << 32) | ((uint64_t)(light & 0x3FFu) << 22) | (uint64_t)(counter & 0x3FFFFFu))));
}
// AA draws of a pixel: index 0 = rotation, 1+i = radius of ray i
// one draw, two uniforms (upper and lower 32 bits): the coordinates of an unshadowed light sample
inline void rng_light_pair(uint64_t stream, uint32_t path, uint32_t light, uint32_t counter, double& u, double& v) {
    const uint64_t h = sm64(stream ^ (((uint64_t)path << 32) | ((uint64_t)(light & 0x3FFu) << 22) | (uint64_t)(counter & 0x3FFFFFu)));
    u = (double)(uint32_t)(h >> 32) * (1.0 / 4294967296.0);
    v = (double)(uint32_t)h * (1.0 / 4294967296.0);
}
inline double rng_aa(uint64_t stream0, uint32_t index) { return u01(sm64(stream0 ^ (0xA5A5A5A500000000ull + index))); }

}  // namespace orc
