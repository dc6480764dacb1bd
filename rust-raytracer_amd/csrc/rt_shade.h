// rt_shade.h — device-side shading: textures, normal/bump maps, lights (point, square
// area light with adaptive subdivision, directional), Phong, and the interpreter that
// walks a material's shader tree at a hit.
//
// The reference evaluates Shadable::get_color recursively and lets reflective /
// translucent shaders call Scene::cast_ray from inside the recursion
// (src/shader/*.rs).  Here a hit is shaded in one pass that returns
//      colour(hit) = L + Σ_k W_k ∘ colour(ray_k)
// i.e. a local colour L plus a short list of secondary rays with per-channel weights
// W_k and cut-off contributions C_k.  The caller adds weight∘L to the pixel and pushes
// the rays on its stack; no recursion is needed because every group shader is linear
// in the colours of the rays below it (MixShader with two ray-spawning factors is the
// one non-linear case and is refused when the scene is created).
//
// Point-light shadow tests are deferred the same way: Phong emits, per light that faces
// the surface, a SHADOW entry carrying that light's finished Phong term; the caller runs
// the any-hit query later, in the same traversal loop as every other ray, and adds the
// term if the light is visible.  Lights whose sampling depends on earlier visibility
// results (the square light's adaptive subdivision) are evaluated in place instead.
#pragma once
#include "rt_trace.h"

struct Spawn {            // a ray requested by the shader tree
    d3 o, d;
    d3 w;                 // weight of the new ray's colour in the sample (RAY) / colour added to the sample if the light is visible (SHADOW)
    d3 c;                 // RAY: Ray::contribution of the new ray (ray.rs:22-26); SHADOW: c.x = distance to the light
    uint32_t ordinal;     // RAY: identifies the spawning shader + branch for the RNG path
    uint32_t shadow;      // 1: visibility test whose colour `w` counts only if nothing blocks the light
};

// Where the shader tree puts the rays it asks for.  A sink numbers rays and shadow tests
// separately; mark() packs both counts into one word so that the interpreter can remember a
// position and later scale the weights of everything added between two marks.
//   ray(o, d, w, c, ordinal)     a reflected / refracted ray: weight in this hit's colour, Ray::contribution, RNG branch
//   shadow(o, d, dist, w)        a point-light visibility test carrying that light's finished Phong term
//   room()                       can another entry be added?
//   scale(from, to, k)           w *= k for the entries between two marks
// Both sinks multiply a new weight by the weight of the hit's own ray first and apply group-shader
// factors (scale) afterwards, so that the two schedules of the trace round identically.
struct ArraySink {        // a per-thread array, read back by the caller
    Spawn* a;
    uint32_t n, nrays, nshadows;
    d3 parent_w;
    uint32_t child_depth;     // Ray::depth of the rays spawned here
    bool drop_dead;           // scene without a sky box: a ray that will fail the gate of Scene::cast_ray adds exactly nothing
    RT_DEV void init(Spawn* p, d3 pw, uint32_t depth, bool drop) { a = p; n = 0; nrays = 0; nshadows = 0; parent_w = pw; child_depth = depth; drop_dead = drop; }
    RT_DEV bool room() const { return n < RT_SHADE_MAX_SPAWN; }
    RT_DEV uint32_t mark() const { return n; }
    RT_DEV void ray(d3 o, d3 d, d3 w, d3 c, uint32_t ordinal) {
        if (drop_dead && !(child_depth > 0 && (c.x + c.y + c.z) > RT_MIN_CONTRIBUTION)) return;   // scene.rs:67-73 with a BLACK background
        Spawn& s = a[n++]; ++nrays;
        s.o = o; s.d = d; s.w = parent_w * w; s.c = c; s.ordinal = ordinal; s.shadow = 0;
    }
    RT_DEV void shadow(d3 o, d3 d, double dist, d3 w) {
        Spawn& s = a[n++]; ++nshadows;
        s.o = o; s.d = d; s.w = parent_w * w; s.c = mk3(dist, 0.0, 0.0); s.ordinal = 0; s.shadow = 1;
    }
    RT_DEV void scale(uint32_t from, uint32_t to, d3 k) { for (uint32_t i = from; i < to; ++i) a[i].w = a[i].w * k; }
    RT_DEV void scale(uint32_t from, uint32_t to, double k) { for (uint32_t i = from; i < to; ++i) a[i].w = a[i].w * k; }
};

// ------------------------------------------------------------------ images
RT_DEV d3 texel_rgb(const DScene& sc, const DImage& im, uint32_t x, uint32_t y) {   // Color::from_rgb, color.rs:78-84
    const uint8_t* p = sc.texels + im.offset + ((size_t)y * im.width + x) * 3;
    return mk3(sc.gamma_lut[p[0]], sc.gamma_lut[p[1]], sc.gamma_lut[p[2]]);
}
// ImageTexture::get_color (src/texture.rs:52-69), including the (w-1)x(h-1) grid and the
// swapped vertical weights.
RT_DEV d3 texture_color(const DScene& sc, uint32_t image, double u, double v) {
    const DImage& im = sc.images[image];
    double gu = fract_(u) * (double)(im.width - 1), gv = fract_(v) * (double)(im.height - 1);
    uint32_t x = (uint32_t)floor(gu), y = (uint32_t)floor(gv);
    double fu = fract_(gu), fv = fract_(gv);
    d3 up_left = texel_rgb(sc, im, x, y), up_right = texel_rgb(sc, im, x + 1, y);
    d3 bot_left = texel_rgb(sc, im, x, y + 1), bot_right = texel_rgb(sc, im, x + 1, y + 1);
    return ((up_left * (1.0 - fu) * fv + up_right * fu * fv) + bot_left * (1.0 - fu) * (1.0 - fv)) + bot_right * fu * (1.0 - fv);
}
// Scene::get_background_color / SkyBox::get_color (src/scene.rs:99-106, 31-43)
template <uint32_t F>
RT_DEV d3 background_color(const DScene& sc, d3 dir) {
    if (!(F & RTF_IMAGES) || !sc.has_background) return mk3(0.0, 0.0, 0.0);
    d3 d = normalize(xf_vector(sc.background_w2l, dir));
    double azimuth = atan2(d.z, d.x);
    double elevation = asin(d.y);
    double u, v;
    surface_coord((azimuth / 2.0) / RT_PI + 0.5, elevation / RT_PI + 0.5, u, v);
    return texture_color(sc, sc.background_image, u, v);
}

// ------------------------------------------------------------------ normal maps (src/normal_map.rs)
RT_DEV double bump_height(const DScene& sc, const DImage& im, double depth, uint32_t x, uint32_t y) {   // :57-60
    return ((double)sc.texels[im.offset + (size_t)y * im.width + x] / 255.0) * depth;
}
RT_DEV d3 map_normal(const DScene& sc, const DNormalMap& nm, double u, double v) {
    const DImage& im = sc.images[nm.image];
    if (nm.kind == 0) {                                           // NormalMap::get_normal :122-129
        uint32_t x = (uint32_t)floor(fract_(u) * (double)im.width), y = (uint32_t)floor(fract_(v) * (double)im.height);
        const uint8_t* p = sc.texels + im.offset + ((size_t)y * im.width + x) * 3;
        return normalize(mk3((double)(128 - (int)p[0]), (double)(128 - (int)p[1]), (double)p[2]));
    }
    double gu = fract_(u) * (double)(im.width - 1), gv = fract_(v) * (double)(im.height - 1);   // BumpMap::get_normal :64-96
    uint32_t x = (uint32_t)floor(gu), y = (uint32_t)floor(gv);
    double fu = fract_(gu), fv = fract_(gv);
    double up_left = bump_height(sc, im, nm.depth, x, y), up_right = bump_height(sc, im, nm.depth, x + 1, y);
    double bot_left = bump_height(sc, im, nm.depth, x, y + 1), bot_right = bump_height(sc, im, nm.depth, x + 1, y + 1);
    double up = fu * up_right + (1.0 - fu) * up_left;
    double bot = fu * bot_right + (1.0 - fu) * bot_left;
    double h = up - bot;
    double h_denom = sqrt(h * h + 1.0);
    d3 y_vec = mk3(0.0, h / h_denom, 1.0 / h_denom);
    double left = fv * up_left + (1.0 - fv) * bot_left;
    double right = fv * up_right + (1.0 - fv) * bot_right;
    h = right - left;
    h_denom = sqrt(h * h + 1.0);
    d3 x_vec = mk3(h / h_denom, 0.0, 1.0 / h_denom);
    return normalize(y_vec + x_vec);
}
// NormalMappable::calculate_normal = basis(old_normal, tangent) * n  (normal_map.rs:10-12, matrix.rs:112-128)
RT_DEV d3 perturbed_normal(const DScene& sc, const DNormalMap& nm, double u, double v, d3 old_normal, d3 up) {
    d3 n = map_normal(sc, nm, u, v);
    if (up.x == old_normal.x && up.y == old_normal.y && up.z == old_normal.z) up = mk3(0.0, 0.0, -1.0);
    d3 side = normalize(cross(old_normal, up));
    d3 up2 = normalize(cross(side, old_normal));
    return mk3((side.x * n.x + up2.x * n.y) + old_normal.x * n.z,
               (side.y * n.x + up2.y * n.y) + old_normal.y * n.z,
               (side.z * n.x + up2.z * n.y) + old_normal.z * n.z);
}

// ------------------------------------------------------------------ Phong + lights
struct PhongCtx {
    d3 p, n, view;        // hit point, (possibly perturbed) normal, -ray.direction
    d3 diffuse, specular;
    double shininess;
    d3 total;
};
// x^shininess of the specular term (src/shader/phong.rs:40, f64::powf), x >= 0.  Materials almost
// always carry a small whole-number exponent: those are raised by repeated squaring (at most
// 12 multiplications, within 2 ulp of the correctly rounded power) instead of the general pow,
// whose ~200 instructions were a third of the cost of a light sample in area-lit scenes.
RT_DEV double shininess_pow(double x, double shininess) {
    if (shininess >= 1.0 && shininess <= 64.0 && shininess == floor(shininess)) {
        uint32_t e = (uint32_t)shininess;
        double r = 1.0;
        while (e) { if (e & 1u) r *= x; x *= x; e >>= 1; }
        return r;
    }
    return pow(x, shininess);
}
// one Illum::Lit term of PhongShader::get_color (src/shader/phong.rs:37-42)
RT_DEV void phong_lit(PhongCtx& c, double surface_dot, d3 light_direction, d3 intensity) {
    d3 reflection_direction = normalize(2.0 * surface_dot * c.n - light_direction);
    double specular_factor = shininess_pow(fmax(dot(reflection_direction, c.view), 0.0), c.shininess) / surface_dot;
    d3 phong_color = c.specular * specular_factor + c.diffuse;
    c.total = c.total + phong_color * surface_dot * intensity;
}

// Shadow test of Lightable::get_illums_at (src/light.rs:31-36): does anything lie between the
// hit point and the light?  Used by the lights that are evaluated inside the shader.
template <uint32_t F, bool COUNT>
RT_DEV bool occluded(const DScene& sc, d3 o, d3 d, double max_distance, Work<COUNT>& wk) {
    SceneHit h;
    RT_COUNT(wk, rays_shadow);
    return trace_scene<F, COUNT, false>(sc, o, d, true, max_distance, h, wk);
}

// SquareLight::get_illums_at → subdivide_illumination (src/light.rs:248-298, 370-374).
//
// The reference walks a quadtree over the light, depth first and one hit at a time: four jittered,
// shadow-tested probes per quadrant, then either their Phong terms (deepest level, or all four
// dark), 4^depth untested samples in their place (all four lit: guess_illums, :222-245), or the four
// child quadrants.  A hit in a penumbra visits up to 341 quadrants, its neighbours one, so a wave
// that gives each lane its own hit idles most of its lanes.  Here the lanes that reach this
// function together share ALL the work of their hits, level by level:
//   - every lane keeps the quadrants of its own hit at the current level as a bit set (position
//     code within the level: two bits per ancestor);
//   - round by round, as many quadrants as the wave has lanes / 4 are published (owner lane and
//     code, in LDS), and each lane runs ONE probe — with its shadow ray — of one of them, for
//     whichever hit it belongs to;
//   - the owners then read the outcomes of their quadrants and add the terms, mark the quadrant
//     for filling, or set the four children in the next level's bit set;
//   - the quadrants to fill are sampled by all lanes together, one quadrant (of 64 samples or more)
//     or four (of 16) at a time.
// This needs the random draws to be addressable: a draw is keyed by (hit, light, quadrant, number
// within the quadrant), not by how many draws came before it (rt_math.h; the CPU checker follows
// the same rule).  Each owner adds its terms in a fixed order (level, then code), so the result is
// the same whichever lanes helped; the filled samples of a quadrant are summed in fixed point
// (integer limbs) for the same reason.  Against the reference's depth-first f64 sum the order
// differs, i.e. the result by rounding only.
RT_DEV d3 phong_term(d3 n, d3 view, d3 diffuse, d3 specular, double shininess, double surface_dot, d3 light_direction, d3 intensity) {
    d3 reflection_direction = normalize(2.0 * surface_dot * n - light_direction);
    double specular_factor = shininess_pow(fmax(dot(reflection_direction, view), 0.0), shininess) / surface_dot;
    d3 phong_color = specular * specular_factor + diffuse;
    return phong_color * surface_dot * intensity;
}
RT_DEV d3 bcast3(d3 v, int src) { return mk3(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src)); }
RT_DEV uint64_t bcast_u64(uint64_t v, int src) { return ((uint64_t)__shfl((uint32_t)(v >> 32), src) << 32) | (uint64_t)__shfl((uint32_t)v, src); }
// Sum of `v` over the lanes of `here` below this one, and over all of them (uniform control flow).
RT_DEV void wave_exscan(unsigned long long here, uint32_t lane, uint32_t v, uint32_t& below, uint32_t& total) {
    below = 0; total = 0;
    for (unsigned long long m = here; m; m &= m - 1ull) {
        const int l = __ffsll((long long)m) - 1;
        const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v, l);
        if ((uint32_t)l < lane) below += x;
        total += x;
    }
}
struct QuadBox { double minx, miny, maxx, maxy; };
// The quadrant with position code `code` at `level` (0: the whole light), by the reference's own
// arithmetic from the root down (light.rs:284-297; children in visiting order).
RT_DEV QuadBox quad_box(double half, uint32_t level, uint32_t code) {
    QuadBox b{-half, -half, half, half};
    for (int k = (int)level - 1; k >= 0; --k) {
        const uint32_t child = (code >> (2 * k)) & 3u;
        const double hx = (b.maxx - b.minx) * 0.5, hy = (b.maxy - b.miny) * 0.5;
        const double midx = b.minx + hx, midy = b.miny + hy;
        if (child == 0u) { b.miny = b.miny + hy; b.maxx = b.maxx - hx; }          // (left, top)
        else if (child == 1u) { b.minx = midx; b.miny = midy; }                    // (mid, max)
        else if (child == 2u) { b.maxx = midx; b.maxy = midy; }                    // (min, mid)
        else { b.minx = midx; b.maxy = b.maxy - hy; }                              // (bot, right)
    }
    return b;
}
RT_DEV uint32_t quad_node(uint32_t level, uint32_t code) { return ((1u << (2u * level)) - 1u) / 3u + code; }

#define RT_AREA_WAVES(F) (((F) & RTF_LDS_BVH) ? 8 : 1)        // waves per workgroup of the kernels that shade area lights
#define RT_AREA_LDS_BYTES(waves) ((waves) * 2048u)            // upper bound of the static LDS below, per workgroup
template <uint32_t F, bool COUNT>
RT_DEV void square_light(const DScene& sc, const DLight& L, PhongCtx& c, uint64_t stream, uint32_t path, Work<COUNT>& wk) {
    constexpr int WAVES = RT_AREA_WAVES(F);
    __shared__ uint16_t q_pub[WAVES][64];                          // this round's quadrants: owner lane << 8 | code
    __shared__ __align__(4) uint8_t q_lit[WAVES][64];              // ... their probes' outcomes
    __shared__ double q_term[WAVES][64][3];                        // ... and Phong terms (deepest level)
    __shared__ uint16_t g_pub[WAVES][4];                           // the quadrants being filled
    __shared__ unsigned long long g_acc[WAVES][4][9];              // ... 3 channels x 3 limbs of weight 2^0, 2^-40, 2^-80
    const uint32_t lane = threadIdx.x & 63u, wave = (threadIdx.x >> 6) % WAVES;
    const unsigned long long here = __ballot(1);
    const uint32_t H = (uint32_t)__popcll(here), rank = (uint32_t)__popcll(here & ((1ull << lane) - 1ull));
    const uint32_t Q = H >= 4u ? H / 4u : 1u;                      // quadrants per round
    const double half = L.size * 0.5;
    const d3 lpos = ld3(L.position), lcol = ld3(L.color);
    const d3 lcp = lcol * L.power;
    uint32_t cur[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, nxt[8];
    for (uint32_t level = 0; level < 5u; ++level) {                // SUB_DEPTH = 5, light.rs:187
        const uint32_t depth = 5u - level;
        uint32_t cnt = 0;
        for (int w = 0; w < 8; ++w) cnt += (uint32_t)__popc(cur[w]);
        uint32_t lo, total;
        wave_exscan(here, lane, cnt, lo, total);
        if (total == 0u) break;
        for (int w = 0; w < 8; ++w) nxt[w] = 0u;
        unsigned long long fill = 0ull;                            // quadrants of this level to fill with samples (levels 0..3: codes < 64)
        uint32_t taken = 0, cw = 0, cm = cur[0];
        for (uint32_t base = 0; base < total; base += Q) {
            // ---- owners publish their quadrants number base .. base+Q-1 of the wave
            uint32_t my_first = 0, my_n = 0;
            while (taken < cnt && lo + taken < base + Q) {
                while (cm == 0u) cm = cur[++cw];
                const uint32_t code = cw * 32u + (uint32_t)(__ffs((int)cm) - 1);
                cm &= cm - 1u;
                const uint32_t slot = lo + taken - base;
                if (my_n == 0u) my_first = slot;
                q_pub[wave][slot] = (uint16_t)((lane << 8) | code);
                ++my_n; ++taken;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- one probe per lane (several in turn when fewer than four lanes are here)
            const uint32_t nq = min(Q, total - base);
            for (uint32_t t = 0; t * H < 4u * nq; ++t) {
                const uint32_t item = t * H + rank;
                const bool valid = item < 4u * nq;
                const uint32_t s = item & 3u;
                const uint32_t e = q_pub[wave][valid ? item >> 2 : 0u];
                const int owner = valid ? (int)(e >> 8) : (int)lane;
                const uint32_t code = e & 255u;
                const d3 P = bcast3(c.p, owner), N = bcast3(c.n, owner);
                const uint64_t ostream = bcast_u64(stream, owner);
                const uint32_t opath = __shfl(path, owner);
                d3 V = mk3(0.0, 0.0, 0.0), KD = V, KS = V;
                double SH = 1.0;
                if (depth == 1u) { V = bcast3(c.view, owner); KD = bcast3(c.diffuse, owner); KS = bcast3(c.specular, owner); SH = __shfl(c.shininess, owner); }
                bool lit = false;
                if (valid) {
                    // subdivide_points :206-218 — this probe's jittered point
                    const QuadBox q = quad_box(half, level, code);
                    const double chx = q.maxx - q.minx, chy = q.maxy - q.miny;
                    const double lowx = q.minx + chx * 0.25, lowy = q.miny + chy * 0.25;
                    const double highx = q.minx + chx * 0.75, highy = q.miny + chy * 0.75;
                    const double cx = s == 1u ? highx : (s == 3u ? lowx + chx * 0.5 : lowx);
                    const double cz = s == 0u ? lowy + chy * 0.5 : (s == 1u ? highy : lowy);
                    const double rx = chx * 0.25, ry = chy * 0.25;
                    const double area_porp = (q.maxx - q.minx) * (q.maxy - q.miny) * L.inv_area * 0.25;   // :251
                    const uint32_t draw = quad_node(level, code) * 2048u + 2u * s;
                    const double dx = -rx + (rx - (-rx)) * rng_light(ostream, opath, L.index, draw);
                    const double dz = -ry + (ry - (-ry)) * rng_light(ostream, opath, L.index, draw + 1u);
                    const d3 hit_to_light = (lpos + mk3(cx + dx, 0.0, cz + dz)) - P;
                    const d3 light_direction = normalize(hit_to_light);
                    const double surface_dot = dot(light_direction, N);
                    if (surface_dot > 0.0) {
                        const double light_distance = length(hit_to_light);
                        if (!occluded<F, COUNT>(sc, P, light_direction, light_distance, wk)) {
                            lit = true;
                            if (depth == 1u) {                     // :277-279: the deepest level's probes are the samples
                                const d3 intensity = lcp / (4.0 * RT_PI * light_distance * light_distance) * area_porp;   // :365-368, :270
                                st3(q_term[wave][item], phong_term(N, V, KD, KS, SH, surface_dot, light_direction, intensity));
                            }
                        }
                    }
                    q_lit[wave][item] = lit ? 1 : 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- owners read the outcomes of their quadrants, in order
            for (uint32_t j = 0; j < my_n; ++j) {
                const uint32_t slot = my_first + j;
                const uint32_t code = q_pub[wave][slot] & 255u;
                const uint32_t lit4 = *reinterpret_cast<const uint32_t*>(&q_lit[wave][4u * slot]) & 0x01010101u;
                if (depth == 1u) {
                    for (uint32_t s = 0; s < 4u; ++s)
                        if ((lit4 >> (8u * s)) & 1u) { RT_COUNT(wk, shade_evals); c.total = c.total + ld3(q_term[wave][4u * slot + s]); }
                } else if (lit4 == 0x01010101u) {
                    fill |= 1ull << code;
                } else if (lit4 != 0u) {
                    nxt[(4u * code) >> 5] |= 0xFu << ((4u * code) & 31u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- fill the quadrants whose four probes were lit: 4^depth samples each, no shadow tests
        uint32_t fcnt = (uint32_t)__popcll(fill), flo, ftotal;
        wave_exscan(here, lane, fcnt, flo, ftotal);
        if (ftotal != 0u) {
            const uint32_t tn = 1u << (2u * depth);
            const uint32_t G = (tn == 16u && H == 64u) ? 4u : 1u;   // small quadrants: four at a time, sixteen lanes each
            const uint32_t per = H / G;
            // Whole-number exponents (the usual case) are raised by squaring under scalar control flow when
            // one quadrant is filled at a time: every lane then has the same exponent.
            uint32_t ftaken = 0;
            for (uint32_t fbase = 0; fbase < ftotal; fbase += G) {
                    uint32_t my_first = 0, my_n = 0;
                while (ftaken < fcnt && flo + ftaken < fbase + G) {
                    const uint32_t code = (uint32_t)(__ffsll((long long)fill) - 1);
                    fill &= fill - 1ull;
                    const uint32_t slot = flo + ftaken - fbase;
                    if (my_n == 0u) my_first = slot;
                    g_pub[wave][slot] = (uint16_t)((lane << 8) | code);
                    for (int k = 0; k < 9; ++k) g_acc[wave][slot][k] = 0ull;
                    ++my_n; ++ftaken;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t g = rank / per, r = rank % per;
                const bool valid = fbase + g < ftotal;
                const uint32_t e = g_pub[wave][valid ? g : 0u];
                const int owner = valid ? (int)(e >> 8) : (int)lane;
                const uint32_t code = e & 255u;
                const d3 P = bcast3(c.p, owner), N = bcast3(c.n, owner), V = bcast3(c.view, owner);
                const d3 KD = bcast3(c.diffuse, owner), KS = bcast3(c.specular, owner);
                const double SH = __shfl(c.shininess, owner);
                const uint64_t ostream = bcast_u64(stream, owner);
                const uint32_t opath = __shfl(path, owner);
                const QuadBox q = quad_box(half, level, code);
                const double bap = 4.0 * ((q.maxx - q.minx) * (q.maxy - q.miny) * L.inv_area * 0.25) / (double)tn;   // :223-224
                const uint32_t draw0 = quad_node(level, code) * 2048u + 8u;
                const double wx = q.maxx - q.minx, wz = q.maxy - q.miny;
                const bool whole = SH >= 1.0 && SH <= 64.0 && SH == floor(SH);
                const bool uniform_expo = G == 1u && __builtin_amdgcn_readfirstlane((int)whole) != 0;
                const uint32_t expo = uniform_expo ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)SH) : 0u;
                double l[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                bool odd = false;                                  // a term that the fast arithmetic or fixed point cannot hold
                for (uint32_t i = r; i < (valid ? tn : 0u); i += per) {
                    double u, v;
                    rng_light_pair(ostream, opath, L.index, draw0 + i, u, v);
                    const double sx = q.minx + wx * u, sz = q.miny + wz * v;
                    const d3 hit_to_light = (lpos + mk3(sx, 0.0, sz)) - P;
                    // The divisions and square roots below are the IEEE ones, computed without the range
                    // scaling of the general sequences and with one reciprocal per divisor; `odd` sends a
                    // quadrant with a divisor outside their range to the term-by-term evaluation.
                    const double d2 = dot(hit_to_light, hit_to_light);
                    const double light_distance = sqrt_nr(d2);
                    const double rcp_distance = rcp_nr(light_distance);
                    const d3 light_direction = mk3(div_nr(hit_to_light.x, light_distance, rcp_distance),
                                                   div_nr(hit_to_light.y, light_distance, rcp_distance),
                                                   div_nr(hit_to_light.z, light_distance, rcp_distance));
                    const double surface_dot = dot(light_direction, N);
                    RT_COUNT(wk, shade_evals);
                    const double spread = 4.0 * RT_PI * light_distance * light_distance;       // get_intensity :365-368
                    const double rcp_spread = rcp_nr(spread);
                    const d3 intensity = mk3(div_nr(lcp.x, spread, rcp_spread), div_nr(lcp.y, spread, rcp_spread), div_nr(lcp.z, spread, rcp_spread)) * bap;
                    // phong_term(N, V, KD, KS, SH, surface_dot, light_direction, intensity)
                    const d3 rv = 2.0 * surface_dot * N - light_direction;
                    const double r2 = dot(rv, rv);
                    const double rlen = sqrt_nr(r2);
                    const d3 reflection_direction = rv * div_nr(1.0, rlen, rcp_nr(rlen));
                    double x = fmax(dot(reflection_direction, V), 0.0), powered;
                    if (uniform_expo) {
                        powered = 1.0;
                        for (uint32_t ex = expo; ex; ex >>= 1) { if (ex & 1u) powered *= x; x *= x; }
                    } else {
                        powered = shininess_pow(x, SH);
                    }
                    const double specular_factor = div_nr(powered, surface_dot, rcp_nr(surface_dot));
                    const d3 phong_color = KS * specular_factor + KD;
                    const d3 term = phong_color * surface_dot * intensity;
                    odd |= !((int)nr_in_range(d2) & (int)nr_in_range(r2) & (int)nr_in_range(surface_dot));
                    odd |= !(fabs(term.x) < 0x1p40) | !(fabs(term.y) < 0x1p40) | !(fabs(term.z) < 0x1p40);
                    const double tv[3] = {term.x, term.y, term.z};
                    for (int ch = 0; ch < 3; ++ch) {               // limbs of an odd term are never used
                        const double l0 = trunc(tv[ch]);
                        const double r1 = (tv[ch] - l0) * 0x1p40;
                        const double l1 = trunc(r1);
                        const double l2 = trunc((r1 - l1) * 0x1p40);
                        l[3 * ch] += l0; l[3 * ch + 1] += l1; l[3 * ch + 2] += l2;
                    }
                }
                const unsigned long long odd_lanes = __ballot(odd);
                const unsigned long long group_lanes = G == 1u ? ~0ull : 0xFFFFull << (16u * g);   // (G == 4 only when all 64 lanes are here: rank == lane)
                if (valid && (odd_lanes & group_lanes) == 0ull)
                    for (int k = 0; k < 9; ++k) if (l[k] != 0.0) atomicAdd(&g_acc[wave][g][k], (unsigned long long)(long long)l[k]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t j = 0; j < my_n; ++j) {
                    const uint32_t slot = my_first + j;
                    const bool slot_odd = (odd_lanes & (G == 1u ? ~0ull : 0xFFFFull << (16u * slot))) != 0ull;
                    if (slot_odd) {                                // rare: this quadrant the reference's way, term by term
                        const uint32_t mcode = g_pub[wave][slot] & 255u;
                        const QuadBox mq = quad_box(half, level, mcode);
                        const double tap = 4.0 * ((mq.maxx - mq.minx) * (mq.maxy - mq.miny) * L.inv_area * 0.25) / (double)tn;
                        const uint32_t mdraw = quad_node(level, mcode) * 2048u + 8u;
                        for (uint32_t i = 0; i < tn; ++i) {
                            double u, v;
                            rng_light_pair(stream, path, L.index, mdraw + i, u, v);
                            const double sx = mq.minx + (mq.maxx - mq.minx) * u;
                            const double sz = mq.miny + (mq.maxy - mq.miny) * v;
                            const d3 hit_to_light = (lpos + mk3(sx, 0.0, sz)) - c.p;
                            const double light_distance = length(hit_to_light);
                            const d3 light_direction = hit_to_light / light_distance;
                            phong_lit(c, dot(light_direction, c.n), light_direction, lcp / (4.0 * RT_PI * light_distance * light_distance) * tap);
                        }
                    } else {
                        const unsigned long long* a = g_acc[wave][slot];
                        c.total = c.total + mk3(((double)(long long)a[2] * 0x1p-80 + (double)(long long)a[1] * 0x1p-40) + (double)(long long)a[0],
                                                ((double)(long long)a[5] * 0x1p-80 + (double)(long long)a[4] * 0x1p-40) + (double)(long long)a[3],
                                                ((double)(long long)a[8] * 0x1p-80 + (double)(long long)a[7] * 0x1p-40) + (double)(long long)a[6]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        for (int w = 0; w < 8; ++w) cur[w] = nxt[w];
    }
}

// PhongShader::get_color (src/shader/phong.rs:29-47).  Returns the part that is known now;
// deferred point lights are appended to `spawn` as SHADOW entries.
template <uint32_t F, bool COUNT, class SINK>
RT_DEV d3 phong_color(const DScene& sc, const DShader& sh, d3 p, d3 n, d3 ray_dir, d3 win, uint64_t stream, uint32_t path,
                      SINK& sink, Work<COUNT>& wk) {
    PhongCtx c;
    c.p = p; c.n = n; c.view = -1.0 * ray_dir;
    c.diffuse = ld3(sh.p); c.specular = ld3(sh.p + 3); c.shininess = sh.p[9];
    c.total = ld3(sh.p + 6) * ld3(sc.ambient);                    // phong.rs:32
    for (uint32_t li = 0; li < sc.num_lights; ++li) {
        const DLight& L = sc.lights[li];
        if (L.kind == 0) {                                        // PointLight through Lightable::get_illums_at, light.rs:17-44,140-148
            d3 hit_to_light = ld3(L.position) - p;
            d3 light_direction = normalize(hit_to_light);
            double surface_dot = dot(light_direction, n);
            if (surface_dot <= 0.0) continue;
            double light_distance = length(hit_to_light);
            d3 intensity = (L.power * ld3(L.color)) / (L.falloff[0] + L.falloff[1] * light_distance + L.falloff[2] * light_distance * light_distance) / 1.0;
            bool inline_test = (F & RTF_INLINE_LIGHTS) && sc.inline_point_lights;
            if (inline_test) {
                if constexpr (F & RTF_INLINE_LIGHTS) {
                    if (occluded<F, COUNT>(sc, p, light_direction, light_distance, wk)) continue;
                    RT_COUNT(wk, shade_evals);
                    phong_lit(c, surface_dot, light_direction, intensity);
                }
            } else if (sink.room()) {
                // (the material's constants are read again for every light, through a pointer the compiler cannot see
                // through: kept in registers across the loop they cost the shade kernels their third wave per SIMD)
                const double* mat = sh.p;
                asm volatile("" : "+v"(mat));
                PhongCtx one;
                one.p = p; one.n = n; one.view = c.view;
                one.diffuse = ld3(mat); one.specular = ld3(mat + 3); one.shininess = mat[9];
                one.total = mk3(0.0, 0.0, 0.0);
                phong_lit(one, surface_dot, light_direction, intensity);
                sink.shadow(p, light_direction, light_distance, win * one.total);
            }
        } else if constexpr (F & RTF_INLINE_LIGHTS) {
            if (L.kind == 1) {
                square_light<F, COUNT>(sc, L, c, stream, path, wk);
            } else {                                              // DirectionLight, light.rs:104-119
                d3 dir = ld3(L.position);
                double surface_dot = dot(dir, n);
                if (surface_dot <= 0.0) continue;
                if (occluded<F, COUNT>(sc, p, -1.0 * dir, INFINITY, wk)) continue;
                RT_COUNT(wk, shade_evals);
                phong_lit(c, surface_dot, -1.0 * dir, ld3(L.color) * L.power);
            }
        }
    }
    return c.total;
}

// ------------------------------------------------------------------ secondary-ray construction (src/geometry/ray.rs)
RT_DEV d3 reflect_dir(d3 d, d3 n) { return normalize(d - 2.0 * dot(d, n) * n); }   // :38-53
RT_DEV d3 transmit_dir(d3 d, d3 n, double refractive_index) {                    // :56-84
    double incident_cos = dot(d, n);
    if (incident_cos < 0.0) { incident_cos = -incident_cos; refractive_index = 1.0 / refractive_index; }
    else n = n * -1.0;
    double refraction_factor = 1.0 - refractive_index * refractive_index * (1.0 - incident_cos * incident_cos);
    if (refraction_factor < 0.0) return reflect_dir(d, n);
    return refractive_index * d + (refractive_index * incident_cos - sqrt(refraction_factor)) * n;
}
RT_DEV void fresnel(d3 d, d3 n, double ior, double& kt, double& kr) {             // translucent.rs:19-46
    double cosi = dot(d, n);
    double etai = 1.0, etat = ior;
    if (cosi > 0.0) { double t = etai; etai = etat; etat = t; }
    double sint = etai / etat * sqrt(fmax(1.0 - cosi * cosi, 0.0));
    if (sint >= 1.0) { kt = 0.0; kr = 1.0; return; }
    double cost = sqrt(fmax(1.0 - sint * sint, 0.0));
    cosi = fabs(cosi);
    double rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    double rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    double reflectance = (rs * rs + rp * rp) * 0.5;
    kt = 1.0 - reflectance; kr = reflectance;
}

// ------------------------------------------------------------------ shader-tree interpreter
// Walks the material of a hit depth-first, left to right — the order in which the
// reference's nested get_color calls run — and returns L; spawned rays go to `spawn`.
// KINDS / DEPTH narrow the interpreter to what the materials it will meet can contain (bit k of KINDS:
// shader kind k may occur; DEPTH: deepest nesting of group shaders): a kernel that only shades glass
// (Composite of Reflection and Translucent) carries neither the Phong code nor its registers, one
// that shades at most one level of grouping no frame array in scratch memory.  The caller guarantees
// the bounds (the host derives them from the scene's materials); a shader outside them yields black.
#define RT_SHADER_KINDS_ALL 0xFFu
template <uint32_t F, bool COUNT, class SINK, uint32_t KINDS = RT_SHADER_KINDS_ALL, int DEPTH = RT_SHADE_MAX_FRAMES>
RT_DEV d3 shade_hit(const DScene& sc, uint32_t root_shader, const WorldHit& hit, d3 ray_dir, d3 ray_contrib,
                    uint64_t stream, uint32_t path, SINK& sink, Work<COUNT>& wk) {
    constexpr auto has = [](uint32_t kind) { return (KINDS >> kind) & 1u; };
    struct Frame {
        uint32_t shader, cursor, spawn_begin, child_spawn_begin;
        d3 acc;          // MIX: running product; COMPOSITE: running sum (local parts)
        d3 cin;          // contribution handed to this group
        d3 win;          // weight of the group's value in the hit's colour (the product of the factors above it)
        d3 saved_n;      // CHAIN: normal to restore on exit
    };
    // The innermost open group is kept in `top` (registers); the groups around it wait in `below`.
    // Materials with one level of grouping never touch `below`, whose dynamic indexing puts it into
    // scratch memory: a scratch read has to wait for the queue stores issued before it.
    Frame top;
    Frame below[DEPTH > 1 ? DEPTH - 1 : 1];
    int depth = 0;
    d3 n = hit.n;
    uint32_t cur = root_shader;
    d3 cin = ray_contrib;
    d3 win = mk3(1.0, 1.0, 1.0);                                  // new rays and shadow tests are given their final weight at once
    d3 value = mk3(0.0, 0.0, 0.0);
    for (;;) {
        // ---- enter `cur` with contribution `cin`
        const DShader& sh = sc.shaders[cur];
        bool returned = true;
        switch (sh.kind) {
            case DS_PHONG:
                if constexpr (has(DS_PHONG)) value = phong_color<F, COUNT>(sc, sh, hit.p, n, ray_dir, win, stream, path, sink, wk); else value = mk3(0.0, 0.0, 0.0);
                break;
            case DS_TEXTURE:                                                                  // shader/texture.rs:16-19
                if constexpr ((F & RTF_IMAGES) && has(DS_TEXTURE)) value = texture_color(sc, sh.ref, hit.u, hit.v); else value = mk3(0.0, 0.0, 0.0);
                break;
            case DS_NORMAL_MAP:                                                               // shader.rs:170-173
                if constexpr ((F & RTF_IMAGES) && has(DS_NORMAL_MAP)) value = perturbed_normal(sc, sc.normal_maps[sh.ref], hit.u, hit.v, n, hit.tangent); else value = mk3(0.0, 0.0, 0.0);
                break;
            case DS_REFLECTION: {                                                             // shader/reflection.rs:14-18
                value = mk3(0.0, 0.0, 0.0);
                if constexpr (has(DS_REFLECTION)) {
                    d3 k = ld3(sh.p);
                    if (sink.room()) sink.ray(hit.p, reflect_dir(ray_dir, n), win * k, cin * k, cur * 4 + 0);
                }
                break;
            }
            case DS_TRANSLUCENT: if constexpr (!has(DS_TRANSLUCENT)) { value = mk3(0.0, 0.0, 0.0); break; } else {   // shader/translucent.rs:50-62
                if (sh.p[4] != 0.0) {
                    // the shader's earlier revision (kept as a comment at translucent.rs:59): the colour of the
                    // transmitted ray alone, unscaled, its contribution scaled by `translucency`
                    if (sink.room()) sink.ray(hit.p, transmit_dir(ray_dir, n, sh.p[3]), win, cin * ld3(sh.p), cur * 4 + 1);
                    value = mk3(0.0, 0.0, 0.0);
                    break;
                }
                double kt, kr;
                fresnel(ray_dir, n, sh.p[3], kt, kr);
                if (kt > 0.0 && sink.room()) sink.ray(hit.p, transmit_dir(ray_dir, n, sh.p[3]), win * kt, cin * kt, cur * 4 + 1);
                if (kr > 0.0 && sink.room()) sink.ray(hit.p, reflect_dir(ray_dir, n), win * kr, cin * kr, cur * 4 + 2);
                value = mk3(0.0, 0.0, 0.0);
                break;
            }
            default: if constexpr (DEPTH == 0) { value = mk3(0.0, 0.0, 0.0); break; } else {   // group shaders open a frame and descend into their first child
                if (depth >= DEPTH || sh.num_children == 0) { value = mk3(0.0, 0.0, 0.0); break; }
                if constexpr (DEPTH > 1) { if (depth > 0) below[depth - 1] = top; }
                depth++;
                Frame& f = top;
                f.shader = cur; f.cursor = 0; f.spawn_begin = sink.mark(); f.child_spawn_begin = f.spawn_begin; f.cin = cin; f.win = win; f.saved_n = n;
                if (has(DS_MIX) && sh.kind == DS_MIX) {                                       // shader.rs:66-69
                    f.acc = mk3(1.0, 1.0, 1.0);
                    cur = sc.shader_child_idx[sh.first_child];
                    cin = f.cin * f.acc; win = f.win * f.acc;
                } else if (!has(DS_CHAIN) || sh.kind == DS_COMPOSITE) {                       // shader.rs:95-102
                    f.acc = mk3(0.0, 0.0, 0.0);
                    cur = sc.shader_child_idx[sh.first_child];
                    cin = f.cin * sc.shader_child_weight[sh.first_child]; win = f.win * sc.shader_child_weight[sh.first_child];
                } else {                                                                      // CHAIN shader.rs:134-152
                    if constexpr (F & RTF_IMAGES) {
                        for (uint32_t k = 0; k < sh.nm_count; ++k)
                            n = perturbed_normal(sc, sc.normal_maps[sc.chain_nmaps[sh.nm_begin + k]], hit.u, hit.v, n, hit.tangent);
                    }
                    f.acc = mk3(0.0, 0.0, 0.0);
                    f.cursor = sh.num_children - 1;
                    cur = sc.shader_child_idx[sh.first_child + sh.num_children - 1];
                    cin = f.cin; win = f.win;
                }
                returned = false;
                break;
            }
        }
        if (!returned) continue;
        // ---- `value` (+ spawns since child_spawn_begin) returns to the enclosing frames
        for (;;) {
            if (DEPTH == 0 || depth == 0) return value;
            Frame& f = top;
            const DShader& g = sc.shaders[f.shader];
            if (has(DS_MIX) && g.kind == DS_MIX) {
                // acc' = acc * (value + rays_child): earlier rays scale by value, the child's by acc (they were, through win)
                sink.scale(f.spawn_begin, f.child_spawn_begin, value);
                f.acc = f.acc * value;
            } else if (!has(DS_CHAIN) || g.kind == DS_COMPOSITE) {
                double w = sc.shader_child_weight[g.first_child + f.cursor];   // (the child's rays carry w already, through win)
                f.acc = f.acc + w * value;
            } else {
                f.acc = value;
            }
            f.cursor++;
            if ((!has(DS_CHAIN) || g.kind != DS_CHAIN) && f.cursor < g.num_children) {   // next sibling
                f.child_spawn_begin = sink.mark();
                cur = sc.shader_child_idx[g.first_child + f.cursor];
                const bool mix = has(DS_MIX) && g.kind == DS_MIX;
                cin = mix ? f.cin * f.acc : f.cin * sc.shader_child_weight[g.first_child + f.cursor];
                win = mix ? f.win * f.acc : f.win * sc.shader_child_weight[g.first_child + f.cursor];
                break;
            }
            value = f.acc;                                           // group finished
            if constexpr (has(DS_CHAIN)) n = f.saved_n;
            depth--;
            if constexpr (DEPTH > 1) { if (depth > 0) top = below[depth - 1]; }
        }
    }
}
