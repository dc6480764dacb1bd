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

// SquareLight::get_illums_at → subdivide_illumination (src/light.rs:248-298, 370-374) with
// an explicit stack of quadrants.  Quadrants are visited in the reference's depth-first
// order so that the draw counter of the RNG advances identically.
//
// A quadrant whose four probes are all lit is filled with 4^depth unshadowed samples
// (guess_illums, :222-245), up to 1024 Phong terms for one hit: by far the largest piece of work
// of an area-lit scene, and one lane's alone.  Here the lanes that entered this function
// together stay in its loop until all of them are done, and every such batch of samples is
// spread over all of them: the random draws are counter-based, so sample i can be evaluated by
// any lane, and the terms are summed in fixed point (integer limbs, like the frame's
// accumulators), so the sum does not depend on which lanes helped.  It differs from the
// reference's left-to-right f64 sum by rounding only (relative 1e-16 per term).
RT_DEV d3 phong_term(d3 n, d3 view, d3 diffuse, d3 specular, double shininess, double surface_dot, d3 light_direction, d3 intensity) {
    d3 reflection_direction = normalize(2.0 * surface_dot * n - light_direction);
    double specular_factor = shininess_pow(fmax(dot(reflection_direction, view), 0.0), shininess) / surface_dot;
    d3 phong_color = specular * specular_factor + diffuse;
    return phong_color * surface_dot * intensity;
}
RT_DEV d3 bcast3(d3 v, int src) { return mk3(__shfl(v.x, src), __shfl(v.y, src), __shfl(v.z, src)); }
template <uint32_t F, bool COUNT>
RT_DEV void square_light(const DScene& sc, const DLight& L, PhongCtx& c, uint64_t stream, uint32_t path, Work<COUNT>& wk) {
    __shared__ unsigned long long guess_acc[16][9];               // per wave: 3 channels x 3 limbs of weight 2^0, 2^-40, 2^-80
    struct Quad { double minx, miny, maxx, maxy; uint32_t depth; };
    Quad st[16];
    int sp = 0;
    double half = L.size * 0.5;
    st[sp++] = Quad{-half, -half, half, half, 5u};                // SUB_DEPTH = 5, light.rs:187
    uint32_t counter = 0;
    const d3 lpos = ld3(L.position), lcol = ld3(L.color);
    const uint32_t lane = threadIdx.x & 63u, wave = (threadIdx.x >> 6) & 15u;
    for (;;) {
        const bool busy = sp > 0;
        if (__ballot(busy) == 0ull) break;                         // idle lanes stay on as helpers
        bool task = false;                                         // a batch of unshadowed samples to evaluate
        double tminx = 0.0, tminy = 0.0, tmaxx = 0.0, tmaxy = 0.0, tap = 0.0;
        uint32_t tn = 0, tcounter = 0;
        if (busy) {
            Quad q = st[--sp];
            // subdivide_points :206-218 — four jittered samples, eight draws
            double chx = q.maxx - q.minx, chy = q.maxy - q.miny;
            double lowx = q.minx + chx * 0.25, lowy = q.miny + chy * 0.25;
            double highx = q.minx + chx * 0.75, highy = q.miny + chy * 0.75;
            double cx[4] = {lowx, highx, lowx, lowx + chx * 0.5};
            double cz[4] = {lowy + chy * 0.5, highy, lowy, lowy};
            double rx = chx * 0.25, ry = chy * 0.25;
            double area_porp = (q.maxx - q.minx) * (q.maxy - q.miny) * L.inv_area * 0.25;   // :251
            double s_dot[4]; d3 s_dir[4]; d3 s_int[4]; bool s_lit[4];
            uint32_t lit_count = 0;
            for (int s = 0; s < 4; ++s) {
                double dx = -rx + (rx - (-rx)) * rng_light(stream, path, L.index, counter++);
                double dz = -ry + (ry - (-ry)) * rng_light(stream, path, L.index, counter++);
                d3 light_pos = lpos + mk3(cx[s] + dx, 0.0, cz[s] + dz);
                d3 hit_to_light = light_pos - c.p;
                d3 light_direction = normalize(hit_to_light);
                double surface_dot = dot(light_direction, c.n);
                s_lit[s] = false;
                if (surface_dot <= 0.0) continue;
                double light_distance = length(hit_to_light);
                if (occluded<F, COUNT>(sc, c.p, light_direction, light_distance, wk)) continue;
                s_lit[s] = true; s_dot[s] = surface_dot; s_dir[s] = light_direction;
                s_int[s] = lcol * L.power / (4.0 * RT_PI * light_distance * light_distance) * area_porp;   // :365-368, :270
                lit_count++;
            }
            if (q.depth == 1 || lit_count == 0) {                 // :277-279
                for (int s = 0; s < 4; ++s) if (s_lit[s]) { RT_COUNT(wk, shade_evals); phong_lit(c, s_dot[s], s_dir[s], s_int[s]); }
            } else if (lit_count == 4) {                          // guess_illums :222-245
                task = true;
                tminx = q.minx; tminy = q.miny; tmaxx = q.maxx; tmaxy = q.maxy;
                tn = 1u << (2 * q.depth);
                tap = 4.0 * area_porp / (double)tn;
                tcounter = counter;
                counter += 2u * tn;
            } else {                                              // :284-297, children pushed in reverse visiting order
                double hx = (q.maxx - q.minx) * 0.5, hy = (q.maxy - q.miny) * 0.5;
                double midx = q.minx + hx, midy = q.miny + hy;
                if (sp + 4 <= 16) {
                    st[sp++] = Quad{q.minx + hx, q.miny, q.maxx, q.maxy - hy, q.depth - 1};   // (bot, right)
                    st[sp++] = Quad{q.minx, q.miny, midx, midy, q.depth - 1};                 // (min, mid)
                    st[sp++] = Quad{midx, midy, q.maxx, q.maxy, q.depth - 1};                 // (mid, max)
                    st[sp++] = Quad{q.minx, q.miny + hy, q.maxx - hx, q.maxy, q.depth - 1};   // (left, top)
                }
            }
        }
        unsigned long long pend = __ballot(task);
        if (pend == 0ull) continue;
        const unsigned long long here = __ballot(1);
        const uint32_t helpers = (uint32_t)__popcll(here), rank = (uint32_t)__popcll(here & ((1ull << lane) - 1ull));
        while (pend) {
            const int src = __ffsll((long long)pend) - 1;
            pend &= pend - 1ull;
            // the owner's hit and batch
            const d3 P = bcast3(c.p, src), N = bcast3(c.n, src), V = bcast3(c.view, src);
            const d3 KD = bcast3(c.diffuse, src), KS = bcast3(c.specular, src);
            const double SH = __shfl(c.shininess, src);
            const double bminx = __shfl(tminx, src), bminy = __shfl(tminy, src), bmaxx = __shfl(tmaxx, src), bmaxy = __shfl(tmaxy, src);
            const double bap = __shfl(tap, src);
            const uint32_t bn = __shfl(tn, src), bcounter = __shfl(tcounter, src), bpath = __shfl(path, src);
            const uint64_t bstream = ((uint64_t)__shfl((uint32_t)(stream >> 32), src) << 32) | (uint64_t)__shfl((uint32_t)stream, src);
            if ((int)lane == src) for (int k = 0; k < 9; ++k) guess_acc[wave][k] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            double l[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            bool odd = false;                                      // a term that fixed point cannot hold (non-finite, huge)
            for (uint32_t i = rank; i < bn; i += helpers) {
                double sx = bminx + (bmaxx - bminx) * rng_light(bstream, bpath, L.index, bcounter + 2u * i);
                double sz = bminy + (bmaxy - bminy) * rng_light(bstream, bpath, L.index, bcounter + 2u * i + 1u);
                d3 hit_to_light = (lpos + mk3(sx, 0.0, sz)) - P;
                double light_distance = length(hit_to_light);
                d3 light_direction = hit_to_light / light_distance;
                double surface_dot = dot(light_direction, N);
                RT_COUNT(wk, shade_evals);
                d3 term = phong_term(N, V, KD, KS, SH, surface_dot, light_direction, lcol * L.power / (4.0 * RT_PI * light_distance * light_distance) * bap);
                const double tv[3] = {term.x, term.y, term.z};
                for (int ch = 0; ch < 3; ++ch) {
                    const double v = tv[ch];
                    if (!(fabs(v) < 0x1p40)) { odd = true; continue; }
                    double l0 = trunc(v);
                    double r1 = (v - l0) * 0x1p40;
                    double l1 = trunc(r1);
                    double l2 = trunc((r1 - l1) * 0x1p40);
                    l[3 * ch] += l0; l[3 * ch + 1] += l1; l[3 * ch + 2] += l2;
                }
            }
            for (int k = 0; k < 9; ++k) if (l[k] != 0.0) atomicAdd(&guess_acc[wave][k], (unsigned long long)(long long)l[k]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const bool any_odd = __ballot(odd) != 0ull;
            if ((int)lane == src) {
                if (any_odd) {                                    // rare: evaluate this batch the reference's way, term by term
                    for (uint32_t i = 0; i < tn; ++i) {
                        double sx = tminx + (tmaxx - tminx) * rng_light(stream, path, L.index, tcounter + 2u * i);
                        double sz = tminy + (tmaxy - tminy) * rng_light(stream, path, L.index, tcounter + 2u * i + 1u);
                        d3 hit_to_light = (lpos + mk3(sx, 0.0, sz)) - c.p;
                        double light_distance = length(hit_to_light);
                        d3 light_direction = hit_to_light / light_distance;
                        phong_lit(c, dot(light_direction, c.n), light_direction, lcol * L.power / (4.0 * RT_PI * light_distance * light_distance) * tap);
                    }
                } else {
                    const unsigned long long* g = guess_acc[wave];
                    c.total = c.total + mk3(((double)(long long)g[2] * 0x1p-80 + (double)(long long)g[1] * 0x1p-40) + (double)(long long)g[0],
                                            ((double)(long long)g[5] * 0x1p-80 + (double)(long long)g[4] * 0x1p-40) + (double)(long long)g[3],
                                            ((double)(long long)g[8] * 0x1p-80 + (double)(long long)g[7] * 0x1p-40) + (double)(long long)g[6]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
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
                PhongCtx one = c;
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
template <uint32_t F, bool COUNT, class SINK>
RT_DEV d3 shade_hit(const DScene& sc, uint32_t root_shader, const WorldHit& hit, d3 ray_dir, d3 ray_contrib,
                    uint64_t stream, uint32_t path, SINK& sink, Work<COUNT>& wk) {
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
    Frame below[RT_SHADE_MAX_FRAMES - 1];
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
            case DS_PHONG: value = phong_color<F, COUNT>(sc, sh, hit.p, n, ray_dir, win, stream, path, sink, wk); break;
            case DS_TEXTURE:                                                                  // shader/texture.rs:16-19
                if constexpr (F & RTF_IMAGES) value = texture_color(sc, sh.ref, hit.u, hit.v); else value = mk3(0.0, 0.0, 0.0);
                break;
            case DS_NORMAL_MAP:                                                               // shader.rs:170-173
                if constexpr (F & RTF_IMAGES) value = perturbed_normal(sc, sc.normal_maps[sh.ref], hit.u, hit.v, n, hit.tangent); else value = mk3(0.0, 0.0, 0.0);
                break;
            case DS_REFLECTION: {                                                             // shader/reflection.rs:14-18
                d3 k = ld3(sh.p);
                if (sink.room()) sink.ray(hit.p, reflect_dir(ray_dir, n), win * k, cin * k, cur * 4 + 0);
                value = mk3(0.0, 0.0, 0.0);
                break;
            }
            case DS_TRANSLUCENT: {                                                            // shader/translucent.rs:50-62
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
            default: {   // group shaders open a frame and descend into their first child
                if (depth >= RT_SHADE_MAX_FRAMES || sh.num_children == 0) { value = mk3(0.0, 0.0, 0.0); break; }
                if (depth > 0) below[depth - 1] = top;
                depth++;
                Frame& f = top;
                f.shader = cur; f.cursor = 0; f.spawn_begin = sink.mark(); f.child_spawn_begin = f.spawn_begin; f.cin = cin; f.win = win; f.saved_n = n;
                if (sh.kind == DS_MIX) {                                                      // shader.rs:66-69
                    f.acc = mk3(1.0, 1.0, 1.0);
                    cur = sc.shader_child_idx[sh.first_child];
                    cin = f.cin * f.acc; win = f.win * f.acc;
                } else if (sh.kind == DS_COMPOSITE) {                                         // shader.rs:95-102
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
            if (depth == 0) return value;
            Frame& f = top;
            const DShader& g = sc.shaders[f.shader];
            if (g.kind == DS_MIX) {
                // acc' = acc * (value + rays_child): earlier rays scale by value, the child's by acc (they were, through win)
                sink.scale(f.spawn_begin, f.child_spawn_begin, value);
                f.acc = f.acc * value;
            } else if (g.kind == DS_COMPOSITE) {
                double w = sc.shader_child_weight[g.first_child + f.cursor];   // (the child's rays carry w already, through win)
                f.acc = f.acc + w * value;
            } else {
                f.acc = value;
            }
            f.cursor++;
            if (g.kind != DS_CHAIN && f.cursor < g.num_children) {   // next sibling
                f.child_spawn_begin = sink.mark();
                cur = sc.shader_child_idx[g.first_child + f.cursor];
                cin = g.kind == DS_MIX ? f.cin * f.acc : f.cin * sc.shader_child_weight[g.first_child + f.cursor];
                win = g.kind == DS_MIX ? f.win * f.acc : f.win * sc.shader_child_weight[g.first_child + f.cursor];
                break;
            }
            value = f.acc;                                           // group finished
            n = f.saved_n;
            depth--;
            if (depth > 0) top = below[depth - 1];
        }
    }
}
