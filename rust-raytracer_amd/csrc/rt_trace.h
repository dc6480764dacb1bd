// rt_trace.h — device-side geometry: ray/primitive intersection, the mesh BVH walk,
// the CSG interval engine and the two scene queries (closest hit, any hit).
//
// Everything is f64 like the reference.  Each routine cites the reference code whose
// results it must reproduce; the control structure is different on purpose:
//   * one flat loop over instances with composed transforms instead of the recursive
//     SceneNode::trace (src/scene.rs:196-225);
//   * hit records are kept minimal (distance + two ids) while searching, and the full
//     Intersect (point, normal, tangent, uv) is rebuilt once for the winner;
//   * meshes are walked with an explicit stack over a 2-wide BVH;
//   * composite shapes run a postfix program over fixed-capacity hit lists.
#pragma once
#include "rt_math.h"

#define RT_MIN_DISTANCE 0.001        // Ray::MIN_DISTANCE, src/geometry/ray.rs:12
#define RT_MIN_CONTRIBUTION 0.003    // Ray::MIN_CONTRIBUTION, src/geometry/ray.rs:13
#define RT_PI 3.14159265358979323846

// Dynamic LDS of the trace kernel:
//   [BVH walk stacks: RT_LDS_STACK rows of blockDim.x words][per wave: list of 64 x RT_BATCH walk jobs][BVH blob]
extern __shared__ __align__(16) uint8_t rt_lds_dynamic[];
#define RT_LDS_FIXED_BYTES(threads) ((threads) * (RT_LDS_STACK + RT_BATCH) * 4u)
RT_DEV uint32_t* rt_lds_stack() { return reinterpret_cast<uint32_t*>(rt_lds_dynamic); }
RT_DEV uint32_t* rt_lds_joblist() { return reinterpret_cast<uint32_t*>(rt_lds_dynamic) + (size_t)blockDim.x * RT_LDS_STACK + (threadIdx.x >> 6) * 64u * RT_BATCH; }
RT_DEV const uint8_t* rt_lds_base() { return rt_lds_dynamic + (size_t)RT_LDS_FIXED_BYTES(blockDim.x); }

#define RT_SHADOW_ENTRY 0xFFFFFFFFu

// Copies the scene's small tables into LDS (all threads of the workgroup take part) and returns
// a scene description whose pointers to those tables lead into the copy.  The device code reads
// them many times per ray, each read depending on the one before (instance -> primitive ->
// shader -> child shader ...): out of LDS such a chain costs a fraction of what it costs out of L2.
// HOT is a compile-time choice (false: the tables stay where they are, e.g. too large for LDS) so
// that the compiler knows the address space of every table pointer: reads become LDS
// instructions, which, unlike reads through pointers that may lead anywhere, do not have to wait
// for the queue stores issued before them.
template <bool HOT>
RT_DEV DScene localise_scene(const DScene& sc, uint8_t* lds) {
    DScene l = sc;
    if (!HOT) return l;
    const uint4* src = reinterpret_cast<const uint4*>(sc.hot_blob);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (uint32_t i = threadIdx.x; i < sc.hot_bytes / 16u; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
#define RT_REBASE(field) if (sc.field) l.field = reinterpret_cast<decltype(l.field)>(lds + (reinterpret_cast<const uint8_t*>(sc.field) - sc.hot_blob))
    RT_REBASE(instances); RT_REBASE(inst_box); RT_REBASE(mesh_instances); RT_REBASE(prims); RT_REBASE(tris); RT_REBASE(polys); RT_REBASE(points);
    RT_REBASE(faced); RT_REBASE(meshes); RT_REBASE(csg_roots); RT_REBASE(csg_ops); RT_REBASE(csg_leaves);
    RT_REBASE(shaders); RT_REBASE(shader_bounds); RT_REBASE(shader_child_idx); RT_REBASE(shader_child_weight);
    RT_REBASE(chain_nmaps); RT_REBASE(normal_maps); RT_REBASE(images); RT_REBASE(gamma_lut); RT_REBASE(lights);
#undef RT_REBASE
    return l;
}

struct LRay {            // a ray in some local space, direction normalised (ray.rs:90-97)
    d3 o, d;
};

struct Cand {            // candidate hit in the primitive's own space
    double t;
    uint32_t a;          // primitive-specific id: sphere root, face, plane, triangle, csg leaf
    uint32_t b;          // bit 0: normal inverted; bits 8..: csg hit index
};

struct Details {         // Intersect fields in the primitive's own space (geometry/intersect.rs:4-11)
    d3 p, n, tangent;
    double u, v;
};

template <bool COUNT>
struct Work {            // per-thread work counters, compiled out when COUNT is false
    unsigned long long instance_tests, box_tests, tri_tests, shade_evals, rays_shadow, rays_secondary, stack_overflows, box_tests_f32, csg_events;
    unsigned long long prof[16];
};
// Occupancy probe: one lane of the wave counts the trip, every lane with work counts itself.
#define RT_PROF(w, iters, lanes) do { if (COUNT) { unsigned long long m_ = __ballot(1); \
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) { (w).prof[iters]++; (w).prof[lanes] += (unsigned long long)__popcll(m_); } } } while (0)
#define RT_PROF_CYC(w, which, t0) do { if (COUNT) { unsigned long long m_ = __ballot(1); long long t1_ = clock64(); \
    if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)m_) - 1)) (w).prof[which] += (unsigned long long)(t1_ - (t0)); (t0) = t1_; } } while (0)

#define RT_COUNT(w, field) do { if (COUNT) (w).field++; } while (0)

// Does a hit at world distance `t` on instance `i` replace the best one so far (`best` on `best_i`)?
// The reference compares with strict '<' level by level, own primitive first, then the children in
// insertion order (src/scene.rs:200-217): of hits at equal distance the earlier node in DFS
// pre-order wins.  It measures those distances after a chain of per-level transforms, the flat
// instance loop after one composed transform, so surfaces that tie exactly there can differ by a
// few units in the last place here (and the other way round).  Hits that close are decided by
// exact_tie_order below, the reference's own way; where that cannot be done (meshes, composite shapes,
// wrapped primitives) distances within 8 ulp count as equal and go to the earlier instance.
RT_DEV bool near_tie(double t, double best, double ulps) {
    return fabs(t - best) <= ulps * 2.220446049250313e-16 * fmax(fabs(t), fabs(best));
}
RT_DEV bool replaces_hit(double t, uint32_t i, double best, uint32_t best_i, bool found) {
    if (!found) return t < best;
    if (near_tie(t, best, 8.0)) return i < best_i;
    return t < best;
}

// world/parent ray → local ray; returns s = |M_lin d| (local length of a unit parent step).  Under a uniform scale
// the length is the matrix's own, known beforehand: it stands in for the square root and the division of the
// reference's renormalisation (geometry/ray.rs:90-97) at a cost of one unit in the last place of the direction.
// EXACT takes the reference's arithmetic instead: the pass that decides ties needs every bit.
template <bool EXACT = false>
RT_DEV LRay to_local(const DXform& x, d3 o, d3 d, double& s, double& inv_s) {
    LRay r;
    r.o = xf_point(x.w2l, o);
    d3 v = xf_vector(x.w2l, d);
    if (!EXACT && x.uniform_inv_s != 0.0) { inv_s = x.uniform_inv_s; s = x.uniform_s; }
    else { s = length(v); inv_s = 1.0 / s; }
    r.d = v * inv_s;
    return r;
}

// ------------------------------------------------------------------ sphere (src/primitive/sphere.rs)
// two_intersects :31-44; n = number of valid roots (t >= MIN_DISTANCE, :47), ascending
RT_DEV int sphere_roots(double radius, const LRay& r, double t[2], uint32_t which[2]) {
    d3 l = neg(r.o);
    double adj = dot(l, r.d);
    double d2 = dot(l, l) - (adj * adj);
    double radius2 = radius * radius;
    if (d2 > radius2) return 0;
    double thc = sqrt(radius2 - d2);
    double t0 = adj - thc, t1 = adj + thc;
    int n = 0;
    if (!(t0 < RT_MIN_DISTANCE)) { t[n] = t0; which[n] = 0; ++n; }
    if (!(t1 < RT_MIN_DISTANCE)) { t[n] = t1; which[n] = 1; ++n; }
    return n;
}
RT_DEV void sphere_details(const LRay& r, double t, Details& o) {   // :18-29, :51-56
    o.p = r.o + (t * r.d);
    d3 hn = normalize(o.p);
    double azimuth = atan2(hn.z, hn.x);
    double elevation = asin(hn.y);
    surface_coord((azimuth / 2.0) / RT_PI + 0.5, elevation / RT_PI + 0.5, o.u, o.v);
    o.n = hn;
    o.tangent = cross(hn, cross(mk3(0.0, 1.0, 0.0), hn));
}

// ------------------------------------------------------------------ rectangle / plane / triangle / polygon (src/primitive/plane.rs)
RT_DEV bool rect_hit(double w, double h, const LRay& r, double& t) {   // :77-90
    double hit_distance = r.o.z / r.d.z * -1.0;
    d3 p = r.o + (hit_distance * r.d);
    double hb = w / 2.0, vb = h / 2.0;
    if (hit_distance < RT_MIN_DISTANCE || p.x < -hb || p.x > hb || p.y < -vb || p.y > vb) return false;
    if (!(hit_distance == hit_distance)) return false;   // 0/0: origin in the plane and direction parallel to it
    t = hit_distance;
    return true;
}
RT_DEV void rect_details(double w, double h, const LRay& r, double t, Details& o) {   // :91-96
    o.p = r.o + (t * r.d);
    double hb = w / 2.0, vb = h / 2.0;
    surface_coord((o.p.x / hb + 1.0) / 2.0, (o.p.y / vb + 1.0) / 2.0, o.u, o.v);
    o.n = mk3(0.0, 0.0, 1.0);
    o.tangent = mk3(0.0, 1.0, 0.0);
}
RT_DEV bool plane_hit(const double* p, const LRay& r, double& t) {     // :46-60
    d3 origin = ld3(p), normal = ld3(p + 3);
    double origin_dot = dot(normal, r.o - origin);
    double direction_dot = dot(normal, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance >= RT_MIN_DISTANCE) { t = hit_distance; return true; }
    return false;
}
RT_DEV void plane_details(const double* p, const LRay& r, double t, Details& o) {   // :37-42, :55
    d3 origin = ld3(p), tangent = ld3(p + 6);
    o.p = r.o + (t * r.d);
    d3 a = o.p - origin;
    d3 v = (dot(a, tangent) / dot(tangent, tangent)) * tangent;
    d3 u = a - v;
    surface_coord(fract_(length(u) / p[9]), fract_(length(v) / p[10]), o.u, o.v);
    o.n = ld3(p + 3);
    o.tangent = tangent;
}
RT_DEV bool tri_hit(const DTri& tr, const LRay& r, double& t) {        // :141-164
    d3 n = ld3(tr.n);
    double origin_dot = dot(n, r.o - ld3(tr.v));
    double direction_dot = dot(n, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance > RT_MIN_DISTANCE) {
        d3 p = r.o + (hit_distance * r.d);
        if (dot(cross(ld3(tr.e), p - ld3(tr.v)), n) >= 0.0 && dot(cross(ld3(tr.e + 3), p - ld3(tr.v + 3)), n) >= 0.0 &&
            dot(cross(ld3(tr.e + 6), p - ld3(tr.v + 6)), n) >= 0.0) {
            t = hit_distance;
            return true;
        }
    }
    return false;
}
// Cand::b of a primitive tested with DOUBTING: one of its inside tests came out within rounding of zero (the ray
// runs along an edge), so the last place of the ray's coordinates in the primitive's space decided hit or miss.
#define RT_CAND_DOUBT 0x40000000u
RT_DEV bool doubtful(double value, d3 a, d3 b) { return value * value <= 1e-20 * (dot(a, a) * dot(b, b)); }
// tri_hit, also saying whether the verdict hung on rounding
RT_DEV bool tri_hit_doubt(const DTri& tr, const LRay& r, double& t, bool& doubt) {
    d3 n = ld3(tr.n);
    double origin_dot = dot(n, r.o - ld3(tr.v));
    double direction_dot = dot(n, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance > RT_MIN_DISTANCE) {
        d3 p = r.o + (hit_distance * r.d);
        bool inside = true;
        for (int k = 0; k < 3; ++k) {
            const d3 e = ld3(tr.e + 3 * k), pv = p - ld3(tr.v + 3 * k);
            const double a = dot(cross(e, pv), n);
            doubt = doubt || doubtful(a, e, pv);
            inside = inside && a >= 0.0;
        }
        if (inside) { t = hit_distance; return true; }
    }
    return false;
}
RT_DEV void tri_details(const DTri& tr, const LRay& r, double t, Details& o) {   // :129-136, :158-159
    o.p = r.o + (t * r.d);
    double side_a = length(o.p - ld3(tr.v)), side_b = length(o.p - ld3(tr.v + 3));
    double side_c = tr.side_c;
    double cos_theta = (side_c * side_c + side_a * side_a - side_b * side_b) / (2.0 * side_c * side_a);
    double x = side_a * cos_theta;
    double y = sqrt(side_a * side_a - x * x);
    surface_coord(x / tr.sc_width, y / tr.sc_height, o.u, o.v);
    o.n = ld3(tr.n);
    o.tangent = ld3(tr.tangent);
}
RT_DEV bool poly_hit(const DScene& sc, const DPoly& pl, const LRay& r, double& t) {   // :207-234
    d3 n = ld3(pl.n);
    const double* V = sc.points + 3 * (size_t)pl.first_vertex;
    const double* E = sc.points + 3 * (size_t)pl.first_edge;
    double origin_dot = dot(n, r.o - ld3(V));
    double direction_dot = dot(n, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance > RT_MIN_DISTANCE) {
        d3 p = r.o + (hit_distance * r.d);
        for (uint32_t i = 0; i < pl.count; ++i)
            if (!(dot(cross(ld3(E + 3 * i), p - ld3(V + 3 * i)), n) >= 0.0)) return false;
        t = hit_distance;
        return true;
    }
    return false;
}

RT_DEV bool poly_hit_doubt(const DScene& sc, const DPoly& pl, const LRay& r, double& t, bool& doubt) {
    d3 n = ld3(pl.n);
    const double* V = sc.points + 3 * (size_t)pl.first_vertex;
    const double* E = sc.points + 3 * (size_t)pl.first_edge;
    double origin_dot = dot(n, r.o - ld3(V));
    double direction_dot = dot(n, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance > RT_MIN_DISTANCE) {
        d3 p = r.o + (hit_distance * r.d);
        bool inside = true;
        for (uint32_t i = 0; i < pl.count; ++i) {
            const d3 e = ld3(E + 3 * i), pv = p - ld3(V + 3 * i);
            const double a = dot(cross(e, pv), n);
            doubt = doubt || doubtful(a, e, pv);
            inside = inside && a >= 0.0;
        }
        if (inside) { t = hit_distance; return true; }
    }
    return false;
}

// ------------------------------------------------------------------ Cube / Tetrahedron (src/primitive/cube.rs:40-69, 131-163)
// Faces are tested in the reference's order; the first two valid faces are kept, ordered
// by distance in the solid's space, and the loop stops at the second.
template <bool EXACT = false>
RT_DEV int faced_hits(const DScene& sc, const DFaced& f, const LRay& r, double t[2], uint32_t face[2]) {
    int n = 0;
    for (uint32_t i = 0; i < f.nfaces; ++i) {
        double s, inv_s;
        LRay fr = to_local<EXACT>(f.face[i], r.o, r.d, s, inv_s);
        double tf;
        bool hit = f.base_is_triangle ? tri_hit(sc.tris[f.base_tri], fr, tf) : rect_hit(f.base_w, f.base_h, fr, tf);
        if (!hit) continue;
        d3 p = xf_point(f.face[i].l2w, fr.o + (tf * fr.d));
        // (the reference measures from the ray's origin as it comes back through the face's matrix, not from the origin
        // it started with: Intersect::transform, intersect.rs:31-34, on the face-space intersect of cube.rs:44-47)
        double dist = length(p - xf_point(f.face[i].l2w, fr.o));
        if (!(dist >= RT_MIN_DISTANCE)) continue;
        if (n == 0) { t[0] = dist; face[0] = i; n = 1; }
        else {
            if (dist < t[0]) { t[1] = t[0]; face[1] = face[0]; t[0] = dist; face[0] = i; }
            else { t[1] = dist; face[1] = i; }
            n = 2;
            break;
        }
    }
    return n;
}
template <bool EXACT = false>
RT_DEV void faced_details(const DScene& sc, const DFaced& f, const LRay& r, uint32_t i, Details& o) {
    double s, inv_s, tf = 0.0;
    LRay fr = to_local<EXACT>(f.face[i], r.o, r.d, s, inv_s);
    Details b;
    if (f.base_is_triangle) { tri_hit(sc.tris[f.base_tri], fr, tf); tri_details(sc.tris[f.base_tri], fr, tf, b); }
    else { rect_hit(f.base_w, f.base_h, fr, tf); rect_details(f.base_w, f.base_h, fr, tf, b); }
    o.p = xf_point(f.face[i].l2w, b.p);
    o.n = xf_normal(f.face[i].nmat, b.n);
    o.tangent = xf_vector(f.face[i].l2w, b.tangent);
    if (f.is_cube) {                                             // cube.rs:12, 33-38
        const double off_u[6] = {2.0, 0.0, 1.0, 3.0, 1.0, 1.0}, off_v[6] = {1.0, 1.0, 1.0, 1.0, 2.0, 0.0};
        surface_coord((b.u + off_u[i]) / 4.0, (b.v + off_v[i]) / 3.0, o.u, o.v);
    } else {                                                     // cube.rs:124-129
        surface_coord((b.u + (double)i / 2.0) / 2.5, 1.0 - b.v, o.u, o.v);
    }
}

// ------------------------------------------------------------------ Polyhedron (src/primitive/polyhedron.rs:99-136)
RT_DEV bool polyhedron_hit(const DScene& sc, const DPrim& ph, const LRay& r, double& t, uint32_t& plane) {
    bool found = false;
    for (uint32_t i = 0; i < ph.count; ++i) {
        const double* pp = sc.prims[ph.ref + i].p;
        double ti;
        if (!plane_hit(pp, r, ti)) continue;
        d3 p = r.o + (ti * r.d);
        bool inside = true;
        for (uint32_t k = 1; k < ph.count; ++k) {
            const double* q = sc.prims[ph.ref + (k + i) % ph.count].p;
            if (dot(ld3(q) - p, ld3(q + 3)) < 0.0) { inside = false; break; }
        }
        if (!inside) continue;
        if (!found || ti < t) { t = ti; plane = i; found = true; }
    }
    return found;
}

RT_DEV bool polyhedron_hit_doubt(const DScene& sc, const DPrim& ph, const LRay& r, double& t, uint32_t& plane, bool& doubt) {
    bool found = false;
    for (uint32_t i = 0; i < ph.count; ++i) {
        const double* pp = sc.prims[ph.ref + i].p;
        double ti;
        if (!plane_hit(pp, r, ti)) continue;
        d3 p = r.o + (ti * r.d);
        bool inside = true;
        for (uint32_t k = 1; k < ph.count; ++k) {
            const double* q = sc.prims[ph.ref + (k + i) % ph.count].p;
            const d3 qp = ld3(q) - p, qn = ld3(q + 3);
            const double a = dot(qp, qn);
            doubt = doubt || doubtful(a, qp, qn);
            inside = inside && !(a < 0.0);
        }
        if (!inside) continue;
        if (!found || ti < t) { t = ti; plane = i; found = true; }
    }
    return found;
}

// ------------------------------------------------------------------ mesh (src/mesh.rs)
RT_DEV void mesh_details(const DMeshTri& tr, const DMeshAttr& at, const LRay& r, double t, Details& o) {   // :429-462
    d3 v1 = ld3(tr.v1), v2 = ld3(tr.v2), v3 = ld3(tr.v3);
    o.p = r.o + (t * r.d);
    d3 vv = cross(v2 - v1, o.p - v1);
    d3 uu = cross(v1 - v3, o.p - v3);
    double u = length(uu) * tr.inv_area;
    double v = length(vv) * tr.inv_area;
    double w = 1.0 - u - v;
    if (at.flags & 1u) o.n = w * ld3(at.n1) + u * ld3(at.n2) + v * ld3(at.n3);
    else o.n = ld3(tr.n);
    if (at.flags & 2u) {
        double tx = (at.uv[0] * w + at.uv[2] * u) + at.uv[4] * v;
        double ty = (at.uv[1] * w + at.uv[3] * u) + at.uv[5] * v;
        surface_coord(tx, ty, o.u, o.v);
    } else {
        surface_coord(0.0, 0.0, o.u, o.v);
    }
    o.tangent = mk3(0.0, 1.0, 0.0);
}

RT_DEV bool slab(const double* lo, const double* hi, d3 o, d3 inv, double tmax, double& tnear) {
    double t0 = (lo[0] - o.x) * inv.x, t1 = (hi[0] - o.x) * inv.x;
    double tn = fmin(t0, t1), tf = fmax(t0, t1);
    t0 = (lo[1] - o.y) * inv.y; t1 = (hi[1] - o.y) * inv.y;
    tn = fmax(tn, fmin(t0, t1)); tf = fmin(tf, fmax(t0, t1));
    t0 = (lo[2] - o.z) * inv.z; t1 = (hi[2] - o.z) * inv.z;
    tn = fmax(tn, fmin(t0, t1)); tf = fmin(tf, fmax(t0, t1));
    tnear = tn;
    return tn <= tf && tf >= 0.0 && tn <= tmax;
}

// ---- the box test of the f32 node records (LBvhNode), in f32
// A box test only culls: it may accept a box the ray misses, never reject one it hits.  That leaves
// room for cheaper arithmetic than the f64 slab test above, as long as every rounding error is
// covered: per axis the entry and exit parameters are computed as
//     t_near = P_near * inv + a_near,   t_far = P_far * inv + a_far        (one fused f32 operation each)
// with P_near / P_far the box's lo or hi plane by the sign of the direction, inv = 1/d rounded to
// f32, a_near = -(o/d) - c and a_far = -(o/d) + c rounded to f32, and c = (|o| + M) |1/d| 2^-21 where
// M bounds |P| over the whole tree.  The three roundings (inv, a, the fused operation) each
// contribute at most (|P| + |o|) |1/d| 2^-24 (+ c 2^-24 for a), together less than c/2, so
// [t_near, t_far] contains the exact interval.  In space the slack is (|o| + M) 2^-21, far below the
// size of any box.  Directions with a zero component give infinities / NaNs; fminf / fmaxf drop
// NaNs, which leaves that axis unconstrained (accepting, never rejecting).
typedef float rt_f2 __attribute__((ext_vector_type(2)));
struct Slab32 {
    rt_f2 inv[3];          // (1/d, 1/d) per axis
    rt_f2 a[3];            // (a_near, a_far) per axis
    bool neg[3];           // 1/d < 0: the hi plane is the near one
};
RT_DEV Slab32 slab32_setup(d3 o, d3 inv, double max_abs) {
    Slab32 r;
    const double ov[3] = {o.x, o.y, o.z}, iv[3] = {inv.x, inv.y, inv.z};
    for (int k = 0; k < 3; ++k) {
        const double oi = ov[k] * iv[k];
        const double c = (fabs(ov[k]) + max_abs) * fabs(iv[k]) * 0x1p-21;
        const float i32 = (float)iv[k];
        r.inv[k] = rt_f2{i32, i32};
        r.a[k] = rt_f2{(float)(-oi - c), (float)(-oi + c)};
        r.neg[k] = iv[k] < 0.0;
    }
    return r;
}
RT_DEV float slab32_limit(double tmax) {            // tmax as f32, rounded up
    const float t = (float)tmax;
    return t + fabsf(t) * 0x1p-22f;
}
RT_DEV bool slab32_box(const float* lo, const float* hi, const Slab32& r, float tmax, float& tnear) {
    const rt_f2 tx = __builtin_elementwise_fma(r.neg[0] ? rt_f2{hi[0], lo[0]} : rt_f2{lo[0], hi[0]}, r.inv[0], r.a[0]);
    const rt_f2 ty = __builtin_elementwise_fma(r.neg[1] ? rt_f2{hi[1], lo[1]} : rt_f2{lo[1], hi[1]}, r.inv[1], r.a[1]);
    const rt_f2 tz = __builtin_elementwise_fma(r.neg[2] ? rt_f2{hi[2], lo[2]} : rt_f2{lo[2], hi[2]}, r.inv[2], r.a[2]);
    const float tn = fmaxf(fmaxf(tx.x, ty.x), tz.x), tf = fminf(fminf(tx.y, ty.y), tz.y);
    tnear = tn;
    return tn <= tf && tf >= 0.0f && tn <= tmax;
}

// LDS copies of a node / a triangle, read 16 bytes at a time (the blob and the records' offsets in
// it are multiples of 16 bytes; left to itself the compiler reads them word by word).
RT_DEV LBvhNode lds_node(const LBvhNode* p) {
    const uint4* w = reinterpret_cast<const uint4*>(p);
    union { uint4 w[4]; LBvhNode n; } u;
    u.w[0] = w[0]; u.w[1] = w[1]; u.w[2] = w[2]; u.w[3] = w[3];
    return u.n;
}
RT_DEV LMeshTri lds_tri(const LMeshTri* p) {
    const uint4* w = reinterpret_cast<const uint4*>(p);
    union { uint4 w[4]; LMeshTri t; } u;
    u.w[0] = w[0]; u.w[1] = w[1]; u.w[2] = w[2]; u.w[3] = w[3];
    return u.t;
}
// Mesh::check_triangle (mesh.rs:416-443) on either triangle record.  `limit`: a hit at or beyond it
// (closest-hit query: the best distance so far; any-hit query: the distance of the light, itself
// included when `inclusive`) changes nothing for the caller, so the three edge tests are skipped
// for it; the result is what the full test followed by the caller's comparison gives.
template <class TRI>
RT_DEV bool mesh_tri_test(const TRI& tr, const LRay& r, double limit, bool inclusive, double& t) {
    d3 v1 = ld3(tr.v1), v2 = ld3(tr.v2), v3 = ld3(tr.v3), normal = ld3(tr.n);
    double origin_dot = dot(normal, r.o - v1);
    double direction_dot = dot(normal, r.d);
    double hit_distance = -origin_dot / direction_dot;
    if (hit_distance > RT_MIN_DISTANCE && (inclusive ? hit_distance <= limit : hit_distance < limit)) {
        d3 p = r.o + (hit_distance * r.d);
        if (dot(cross(v2 - v1, p - v1), normal) < 0.0) return false;
        if (dot(cross(v3 - v2, p - v2), normal) < 0.0) return false;
        if (dot(cross(v1 - v3, p - v3), normal) < 0.0) return false;
        t = hit_distance;
        return true;
    }
    return false;
}

// Closest triangle with t in (MIN_DISTANCE, tmax), or — with `any` — whether some triangle
// has t <= tmax (inclusive) / t < tmax.  One routine serves both queries so that lanes
// tracing shadow rays and lanes tracing camera/secondary rays share the same loop.
// Replaces BoundingNode::check_intersect (mesh.rs:167-279); both structures are exact, so
// the closest hit is the same.
//
// Shape of the walk ("while-while"): an inner loop descends through interior nodes until
// the lane holds a leaf, then the leaf's triangles are tested.  Lanes of a wave therefore
// sit together either in box tests or in triangle tests instead of interleaving the two,
// which is what keeps the 64-wide vector unit filled on incoherent rays.
// With LDS the node / triangle records come from the workgroup's LDS copy (LBvhNode /
// LMeshTri), otherwise from the global-memory arrays.
#define RT_WALK_DONE 0xFFFFFFFFu
template <bool LDS, bool COUNT>
RT_DEV bool mesh_walk(const DScene& sc, const DMesh& m, const LRay& r, bool any, double tmax, bool inclusive, double& t_out, uint32_t& tri_out, Work<COUNT>& wk) {
    d3 inv = mk3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    double tn;
    RT_COUNT(wk, box_tests);
    if (!slab(m.lo, m.hi, r.o, inv, tmax, tn)) return false;
    const DBvhNode* gnodes = sc.bvh_nodes + m.node_off;
    const DMeshTri* gtris = sc.mesh_tris + m.tri_off;
    const LBvhNode* lnodes = reinterpret_cast<const LBvhNode*>(rt_lds_base() + m.lds_node_off);
    const LMeshTri* ltris = reinterpret_cast<const LMeshTri*>(rt_lds_base() + m.lds_tri_off);
    uint32_t stack[RT_BVH_STACK];
    int sp = 0;
    uint32_t cur = 0;
    bool found = false;
    double best = tmax;
    Slab32 s32 = slab32_setup(r.o, inv, m.max_abs);          // (LDS walk: the f32 box test)
    float best32 = slab32_limit(best);
    while (cur != RT_WALK_DONE) {
        while (!(cur & 0x80000000u)) {                         // descend to a leaf
            bool h0, h1, swap;
            uint32_t c0, c1;
            if (COUNT) { wk.box_tests += 2; if (LDS) wk.box_tests_f32 += 2; }
            if (LDS) {
                float tn0, tn1;
                const LBvhNode n = lds_node(lnodes + cur);
                h0 = slab32_box(n.lo[0], n.hi[0], s32, best32, tn0);
                h1 = slab32_box(n.lo[1], n.hi[1], s32, best32, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
            } else {
                double tn0, tn1;
                const DBvhNode& n = gnodes[cur];
                h0 = slab(n.lo[0], n.hi[0], r.o, inv, best, tn0);
                h1 = slab(n.lo[1], n.hi[1], r.o, inv, best, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
            }
            if (h0 && h1) {
                cur = swap ? c1 : c0;
                if (sp < RT_BVH_STACK) stack[sp++] = swap ? c0 : c1;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else cur = sp > 0 ? stack[--sp] : RT_WALK_DONE;
        }
        if (cur == RT_WALK_DONE) break;
        uint32_t base = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
        for (uint32_t i = 0; i < cnt; ++i) {                     // the leaf's triangles
            double t;
            RT_COUNT(wk, tri_tests);
            const double lim = any ? tmax : best; const bool incl = any && inclusive;
            bool hit = LDS ? mesh_tri_test(lds_tri(ltris + base + i), r, lim, incl, t) : mesh_tri_test(gtris[base + i], r, lim, incl, t);
            if (hit) {
                if (any) {
                    if (inclusive ? t <= tmax : t < tmax) { found = true; }
                } else if (t < best) {
                    best = t; tri_out = base + i; found = true;
                    if (LDS) best32 = slab32_limit(best);
                }
            }
        }
        if (any && found) break;
        cur = sp > 0 ? stack[--sp] : RT_WALK_DONE;
    }
    t_out = best;
    return found;
}

// The mesh part of all queries of a WAVE's trip in one loop.  Each lane brings up to RT_BATCH
// queries (jobs[], bit j of live_mask set for those that still need the meshes); how long a query
// walks differs by two orders of magnitude and the queries of one lane are alike (they belong to
// the same few pixels), so the wave pools them: the live queries of all 64 lanes go into a list
// in LDS and every lane takes the next one off the list whenever it is free.  A query is walked
// through every mesh instance by the lane that took it; the result is merged into the owner's
// jobs[] entry: the closest hit over the instances already tested (ties go to the earlier
// instance, src/scene.rs:200-217), or the any-hit verdict.
// Setting up (taking a query, ray -> instance space, root box) is only done when refill_min
// lanes are waiting for it or nobody is walking any more.  The walk itself is mesh_walk's; its
// stack lives in LDS (one column per thread), the overflow in the lane's global memory.
template <bool LDS, bool COUNT>
RT_DEV void mesh_walk_batch(const DScene& sc, uint8_t* lane_mem, uint32_t live_mask, uint32_t refill_min, Work<COUNT>& wk) {
    const uint32_t nm = sc.num_mesh_instances;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // the wave's job list: (owner lane << 8) | query index
    uint32_t* list = rt_lds_joblist();
    uint32_t total = 0;
    if (nm > 0) {
        for (uint32_t j = 0; j < RT_BATCH; ++j) {
            const bool live = (live_mask >> j) & 1u;
            const unsigned long long m = __ballot(live);
            if (live) list[total + (uint32_t)__popcll(m & lt)] = (lane << 8) | j;
            total += (uint32_t)__popcll(m);
        }
    }
    if (total == 0) return;
    // queries written by phase A and the rays behind them are read by other lanes of the wave from here on
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint8_t* wave_mem = lane_mem - (size_t)lane * RT_LANE_BYTES;      // lane 0's block; the wave's blocks are consecutive
    uint32_t next = 0;                                               // wave-uniform: first query not yet taken
    bool have = false, walking = false;
    TraceJob* q = nullptr;
    const double* od = nullptr;
    uint32_t m = 0;
    // the query being walked: running result over the mesh instances
    double qt = 0.0, qct = 0.0;
    uint32_t qinst = 0, qfound = 0, qca = 0;
    LRay lr; lr.o = mk3(0.0, 0.0, 0.0); lr.d = lr.o;
    d3 inv = lr.o;
    Slab32 s32 = slab32_setup(lr.o, lr.o, 0.0);                      // (LDS walk: the f32 box test)
    float best32 = 0.0f;
    double best = 0.0, tmax = 0.0, inv_s = 1.0;
    bool any = false, inclusive = false, found = false;
    uint32_t tri = 0, inst = 0, cur = RT_WALK_DONE;
    const DBvhNode* gnodes = sc.bvh_nodes;
    const DMeshTri* gtris = sc.mesh_tris;
    const LBvhNode* lnodes = reinterpret_cast<const LBvhNode*>(rt_lds_base());
    const LMeshTri* ltris = reinterpret_cast<const LMeshTri*>(rt_lds_base());
    uint32_t* lstack = rt_lds_stack() + threadIdx.x;
    const uint32_t lstride = blockDim.x;
    uint32_t* gstack = reinterpret_cast<uint32_t*>(lane_mem + RT_LANE_OFF_BVH);
    int sp = 0;
    auto push = [&](uint32_t v) {
        if (sp < RT_LDS_STACK) lstack[sp * lstride] = v;
        else if (sp < RT_BVH_STACK) gstack[sp - RT_LDS_STACK] = v;
        ++sp;
    };
    auto pop = [&]() -> uint32_t {
        if (sp == 0) return RT_WALK_DONE;
        --sp;
        return sp < RT_LDS_STACK ? lstack[sp * lstride] : gstack[sp - RT_LDS_STACK];
    };
    auto finish_query = [&]() {                                       // write the merged result back to its owner
        q->t = qt; q->inst = qinst; q->found = qfound; q->ct = qct; q->ca = qca; q->cb = 0u;
        have = false;
    };
    for (;;) {
        const bool need = !walking && (have || next < total);
        const unsigned long long nmask = __ballot(need), wmask = __ballot(walking);
        if (nmask == 0ull && wmask == 0ull) break;
        if (nmask != 0ull && (wmask == 0ull || __popcll(nmask) >= refill_min)) {
            // free lanes take the next queries off the list
            const bool want = need && !have;
            const unsigned long long jm = __ballot(want);
            const uint32_t idx = next + (uint32_t)__popcll(jm & lt);
            next += (uint32_t)__popcll(jm);
            if (want && idx < total) {
                const uint32_t e = list[idx];
                uint8_t* owner = wave_mem + (size_t)(e >> 8) * RT_LANE_BYTES;
                q = reinterpret_cast<TraceJob*>(owner + RT_LANE_OFF_JOBS) + (e & 255u);
                od = reinterpret_cast<const double*>(owner + q->src);
                any = (q->flags & 1u) != 0;
                qt = q->t; qinst = q->inst; qfound = q->found; qct = q->ct; qca = q->ca;
                have = true; m = 0;
            }
            if (need && have) {
                RT_PROF(wk, PF_T_REFILLS, PF_T_REFILL_LANES);
                do {   // next mesh instance whose bounding box the ray reaches
                    inst = sc.mesh_instances[m];
                    const DInstance& in = sc.instances[inst];
                    const DMesh& mesh = sc.meshes[sc.prims[in.prim].ref];
                    double s;
                    lr = to_local(in.x, ld3(od), ld3(od + 3), s, inv_s);
                    RT_COUNT(wk, instance_tests);
                    inv = mk3(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z);
                    tmax = qt * s;
                    inclusive = (in.is_root & RT_INST_ROOT) != 0;
                    RT_COUNT(wk, box_tests);
                    double tn;
                    if (slab(mesh.lo, mesh.hi, lr.o, inv, tmax, tn) && !(sc.has_clear_to_light && any && (in.is_root & RT_INST_CLEAR_TO_LIGHT))) {
                        walking = true; cur = 0; sp = 0; found = false; best = tmax;
                        if (LDS) { s32 = slab32_setup(lr.o, inv, mesh.max_abs); best32 = slab32_limit(best); }
                        gnodes = sc.bvh_nodes + mesh.node_off; gtris = sc.mesh_tris + mesh.tri_off;
                        lnodes = reinterpret_cast<const LBvhNode*>(rt_lds_base() + mesh.lds_node_off);
                        ltris = reinterpret_cast<const LMeshTri*>(rt_lds_base() + mesh.lds_tri_off);
                    } else if (++m == nm) {
                        finish_query();
                    }
                } while (!walking && have);
            }
        }
        while (walking && !(cur & 0x80000000u)) {                   // descend to a leaf
            bool h0, h1, swap;
            uint32_t c0, c1;
            if (COUNT) { wk.box_tests += 2; if (LDS) wk.box_tests_f32 += 2; }
            RT_PROF(wk, PF_T_ITERS, PF_T_LANES);
            if (LDS) {
                float tn0, tn1;
                const LBvhNode n = lds_node(lnodes + cur);
                h0 = slab32_box(n.lo[0], n.hi[0], s32, best32, tn0);
                h1 = slab32_box(n.lo[1], n.hi[1], s32, best32, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
            } else {
                double tn0, tn1;
                const DBvhNode& n = gnodes[cur];
                h0 = slab(n.lo[0], n.hi[0], lr.o, inv, best, tn0);
                h1 = slab(n.lo[1], n.hi[1], lr.o, inv, best, tn1);
                c0 = n.child[0]; c1 = n.child[1];
                swap = tn1 < tn0;
            }
            if (h0 && h1) {
                cur = swap ? c1 : c0;
                push(swap ? c0 : c1);
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else cur = pop();
        }
        if (walking) {
            if (cur != RT_WALK_DONE) {                                // the leaf's triangles
                RT_PROF(wk, PF_T_LEAF_ITERS, PF_T_LEAF_LANES);
                uint32_t base = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
                for (uint32_t i = 0; i < cnt; ++i) {
                    double t;
                    RT_COUNT(wk, tri_tests);
                    const double lim = any ? tmax : best; const bool incl = any && inclusive;
                    bool hit = LDS ? mesh_tri_test(lds_tri(ltris + base + i), lr, lim, incl, t) : mesh_tri_test(gtris[base + i], lr, lim, incl, t);
                    if (hit) {
                        if (any) {
                            if (inclusive ? t <= tmax : t < tmax) found = true;
                        } else if (t < best) {
                            best = t; tri = base + i; found = true;
                            if (LDS) best32 = slab32_limit(best);
                        }
                    }
                }
                cur = (any && found) ? RT_WALK_DONE : pop();
            }
            if (cur == RT_WALK_DONE) {                                // this instance is done: merge, move on
                bool settled = false;
                if (found) {
                    if (any) { qfound = 1u; settled = true; }
                    else {
                        double tw = best * inv_s;
                        if (replaces_hit(tw, inst, qt, qinst, qfound != 0u)) {
                            qt = tw; qinst = inst; qfound = 1u;
                            qct = best; qca = tri;
                        }
                    }
                }
                ++m;
                if (settled || m == nm) finish_query();
                walking = false;
            }
        }
    }
    // results were written by other lanes of the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ------------------------------------------------------------------ basic (non-composite) primitives
// get_all_intersects of a primitive without wrappers: up to two hits, ascending.
// DOUBTING: a triangle, polygon or polyhedron also reports in c[0].b (RT_CAND_DOUBT, hit or miss) whether its verdict
// hung on rounding.
template <uint32_t F, bool COUNT, bool DOUBTING = false>
RT_DEV int basic_hits(const DScene& sc, const DPrim& p, const LRay& r, double tmax_closest, Cand c[2], Work<COUNT>& wk) {
    if constexpr (DOUBTING && (F & RTF_MISC) != 0) {
        if (p.kind == DP_TRIANGLE || p.kind == DP_POLYGON || p.kind == DP_POLYHEDRON) {
            bool doubt = false, hit;
            c[0].a = 0;
            if (p.kind == DP_TRIANGLE) hit = tri_hit_doubt(sc.tris[p.ref], r, c[0].t, doubt);
            else if (p.kind == DP_POLYGON) hit = poly_hit_doubt(sc, sc.polys[p.ref], r, c[0].t, doubt);
            else hit = polyhedron_hit_doubt(sc, p, r, c[0].t, c[0].a, doubt);
            c[0].b = doubt ? RT_CAND_DOUBT : 0u;
            return hit ? 1 : 0;
        }
    }
    switch (p.kind) {
        case DP_SPHERE: {
            double t[2]; uint32_t w[2];
            int n = sphere_roots(p.p[0], r, t, w);
            for (int i = 0; i < n; ++i) { c[i].t = t[i]; c[i].a = w[i]; c[i].b = 0; }
            return n;
        }
        case DP_RECTANGLE: { c[0].a = 0; c[0].b = 0; return rect_hit(p.p[0], p.p[1], r, c[0].t) ? 1 : 0; }
        case DP_PLANE: if constexpr (F & RTF_MISC) { c[0].a = 0; c[0].b = 0; return plane_hit(p.p, r, c[0].t) ? 1 : 0; } return 0;
        case DP_TRIANGLE: if constexpr (F & RTF_MISC) { c[0].a = 0; c[0].b = 0; return tri_hit(sc.tris[p.ref], r, c[0].t) ? 1 : 0; } return 0;
        case DP_POLYGON: if constexpr (F & RTF_MISC) { c[0].a = 0; c[0].b = 0; return poly_hit(sc, sc.polys[p.ref], r, c[0].t) ? 1 : 0; } return 0;
        case DP_FACED:
            if constexpr (F & RTF_FACED) {
                double t[2]; uint32_t f[2];
                int n = faced_hits(sc, sc.faced[p.ref], r, t, f);
                for (int i = 0; i < n; ++i) { c[i].t = t[i]; c[i].a = f[i]; c[i].b = 0; }
                return n;
            }
            return 0;
        case DP_POLYHEDRON: if constexpr (F & RTF_MISC) { c[0].b = 0; return polyhedron_hit(sc, p, r, c[0].t, c[0].a) ? 1 : 0; } return 0;
        case DP_MESH:
            // un-wrapped meshes that are node primitives are walked by trace_scene itself; this
            // site only serves meshes behind OneWay / Inverted or inside composite shapes
            if constexpr ((F & RTF_MESH) && (F & (RTF_WRAP | RTF_CSG))) {
                c[0].b = 0;
                return mesh_walk<(F & RTF_LDS_BVH) != 0, COUNT>(sc, sc.meshes[p.ref], r, false, tmax_closest, false, c[0].t, c[0].a, wk) ? 1 : 0;
            }
            return 0;
        default: return 0;
    }
}
template <uint32_t F>
RT_DEV void basic_details(const DScene& sc, const DPrim& p, const LRay& r, const Cand& c, Details& o) {
    switch (p.kind) {
        case DP_SPHERE: sphere_details(r, c.t, o); break;
        case DP_RECTANGLE: rect_details(p.p[0], p.p[1], r, c.t, o); break;
        case DP_PLANE: if constexpr (F & RTF_MISC) plane_details(p.p, r, c.t, o); break;
        case DP_TRIANGLE: if constexpr (F & RTF_MISC) tri_details(sc.tris[p.ref], r, c.t, o); break;
        case DP_POLYGON:
            if constexpr (F & RTF_MISC) {
                o.p = r.o + (c.t * r.d); o.n = ld3(sc.polys[p.ref].n); o.tangent = mk3(0.0, 1.0, 0.0);
                surface_coord(0.0, 0.0, o.u, o.v);
            }
            break;
        case DP_FACED: if constexpr (F & RTF_FACED) faced_details(sc, sc.faced[p.ref], r, c.a, o); break;
        case DP_POLYHEDRON:
            if constexpr (F & RTF_MISC) {
                plane_details(sc.prims[p.ref + c.a].p, r, c.t, o);
                surface_coord(((double)c.a + o.u) / (double)p.count, o.v, o.u, o.v);   // polyhedron.rs:109-113
            }
            break;
        case DP_MESH:
            if constexpr (F & RTF_MESH) {
                const DMesh& m = sc.meshes[p.ref];
                mesh_details(sc.mesh_tris[m.tri_off + c.a], sc.mesh_attrs[m.tri_off + c.a], r, c.t, o);
            }
            break;
        default: break;
    }
    if (c.b & 1u) o.n = neg(o.n);
}
// n·d sign of a candidate, the quantity OneWay / Inverted / hit_direction test
template <uint32_t F>
RT_DEV double facing(const DScene& sc, const DPrim& p, const LRay& r, const Cand& c) {
    if (p.kind == DP_SPHERE && !(c.b & 1u)) return dot(r.d, r.o + (c.t * r.d));   // same sign as d·normalize(p)
    Details d;
    basic_details<F>(sc, p, r, c, d);
    return dot(r.d, d.n);
}
// Apply the OneWay / Inverted chain to a *closest* hit (src/primitive.rs:33-40, 64-72).
template <uint32_t F>
RT_DEV bool wrap_closest(const DScene& sc, const DPrim& p, const LRay& r, Cand& c) {
    for (uint32_t w = p.wrap; w; w >>= 2) {
        double f = facing<F>(sc, p, r, c);
        if ((w & 3u) == DWRAP_ONEWAY) { if (!(f < 0.0)) return false; }
        else { if (!(f >= 0.0)) return false; c.b ^= 1u; }
    }
    return true;
}
// Apply the chain to a get_all_intersects list (src/primitive.rs:42-45, 74-80): OneWay
// filters, Inverted flips every normal without filtering.
template <uint32_t F>
RT_DEV int wrap_all(const DScene& sc, const DPrim& p, const LRay& r, Cand c[2], int n) {
    for (uint32_t w = p.wrap; w; w >>= 2) {
        if ((w & 3u) == DWRAP_ONEWAY) {
            int m = 0;
            for (int i = 0; i < n; ++i) if (facing<F>(sc, p, r, c[i]) < 0.0) c[m++] = c[i];
            n = m;
        } else {
            for (int i = 0; i < n; ++i) c[i].b ^= 1u;
        }
    }
    return n;
}

// ------------------------------------------------------------------ CSG (src/composite_shape.rs:75-123 + shape files)
struct CsgHit {
    double t;            // distance along the ray in the composite's parent space
    uint32_t leaf;       // csg_leaves index
    uint32_t meta;       // bit0 enter, bit1 inverted, bits 8..15 hit index within the leaf, bit 2 leaf cand.b bit0
};
RT_DEV int csg_predicate(uint32_t kind, uint32_t cur, uint32_t states, uint32_t n) {   // 0 skip, 1 emit, 2 emit inverted
    bool s0 = states & 1u, s1 = states & 2u;
    switch (kind) {
        case DC_SUBTRACT:                                          // subtract_shape.rs:46-53
            if (!s1 && cur == 0) return 1;
            if (s0 && cur == 1) return 2;
            return 0;
        case DC_AND: return (((states >> (1 - cur)) & 1u) || (s0 && s1)) ? 1 : 0;   // and_shape.rs:25
        case DC_MULTI_AND: {                                       // and_shape.rs:93
            uint32_t all = (1u << n) - 1u;
            return ((states | (1u << cur)) & all) == all ? 1 : 0;
        }
        case DC_OR: return (!(s0 && s1) && ((!s0 && !s1) || ((states >> cur) & 1u))) ? 1 : 0;   // or_shape.rs:25-27
        case DC_XOR:                                               // xor_shape.rs:25-33
            if (!s0 && !s1) return 1;
            if ((s0 && s1) || ((states >> (1 - cur)) & 1u)) return 2;
            return 1;
    }
    return 0;
}
template <uint32_t F, bool COUNT>
RT_DEV bool csg_closest(const DScene& sc, const DCsgRoot& root, const LRay& r, Cand& out, Work<COUNT>& wk) {
    CsgHit arena[RT_CSG_ARENA];
    uint32_t lbegin[RT_CSG_MAX_LISTS], lcount[RT_CSG_MAX_LISTS];
    uint32_t nl = 0, top = 0;
    for (uint32_t oi = 0; oi < root.op_count; ++oi) {
        const DCsgOp& op = sc.csg_ops[root.op_begin + oi];
        if (op.kind == DC_BASE) {                                  // BaseShape::get_all_intersects, base_shape.rs:49-53
            const DCsgLeaf& leaf = sc.csg_leaves[op.leaf];
            const DPrim& p = sc.prims[leaf.prim];
            double s, inv_s;
            LRay lr = to_local(leaf.x, r.o, r.d, s, inv_s);
            Cand c[2];
            RT_COUNT(wk, instance_tests);                              // (a leaf: ray into its space + primitive test, like a node's)
            int n = basic_hits<F, COUNT>(sc, p, lr, INFINITY, c, wk);
            if constexpr (F & RTF_WRAP) { if (p.wrap) n = wrap_all<F>(sc, p, lr, c, n); }
            lbegin[nl] = top; lcount[nl] = 0;
            for (int i = 0; i < n && top < RT_CSG_ARENA; ++i) {
                CsgHit h;
                h.t = c[i].t * inv_s;
                h.leaf = op.leaf;
                bool enter = facing<F>(sc, p, lr, c[i]) < 0.0;        // hit_direction, composite_shape.rs:27-30
                h.meta = (enter ? 1u : 0u) | ((c[i].b & 1u) << 2) | ((uint32_t)i << 8);
                arena[top++] = h; lcount[nl]++;
            }
            ++nl;
        } else {
            uint32_t k = op.num_children, base = nl - k;
            CsgHit outl[RT_CSG_MAX_HITS];
            uint32_t no = 0, states = 0, cursor[8];
            for (uint32_t i = 0; i < k; ++i) {                     // initial inside-state from the first hit, :99-109
                cursor[i] = 0;
                if (lcount[base + i] > 0 && !(arena[lbegin[base + i]].meta & 1u)) states |= 1u << i;
            }
            for (;;) {                                             // fold of pairwise merges, ties → later list, :33-72, :111
                int pick = -1; double bt = 0.0;
                for (uint32_t i = 0; i < k; ++i) {
                    if (cursor[i] >= lcount[base + i]) continue;
                    double t = arena[lbegin[base + i] + cursor[i]].t;
                    if (pick < 0 || !(bt < t)) { pick = (int)i; bt = t; }
                }
                if (pick < 0) break;
                RT_COUNT(wk, csg_events);
                CsgHit h = arena[lbegin[base + pick] + cursor[pick]++];
                int pr = csg_predicate(op.kind, (uint32_t)pick, states, k);
                if (pr && no < RT_CSG_MAX_HITS) {
                    if (pr == 2) h.meta ^= 3u;                     // invert_normal flips the facing too
                    outl[no++] = h;
                }
                states ^= 1u << pick;
            }
            top = lbegin[base];
            for (uint32_t i = 0; i < no; ++i) arena[top + i] = outl[i];
            lcount[base] = no; top += no;
            nl = base + 1;
        }
    }
    if (nl == 0 || lcount[0] == 0) return false;
    const CsgHit& h = arena[lbegin[0]];
    out.t = h.t;
    out.a = h.leaf;
    out.b = ((h.meta >> 1) & 1u) | (((h.meta >> 2) & 1u) << 1) | ((h.meta >> 8) << 8);
    return true;
}
// ---- the same answer without lists, for the common case
// With every event (leaf hit) at a distance of its own, each predicate of csg_predicate says exactly
// "the value of this node's set operation changes here" (Subtract: P and not N, And, Or, Xor,
// MultiAnd), a child's initial state "its first event is an exit" says "the ray starts inside it",
// and an emitted event enters the composite iff that value becomes true.  So the closest hit of the
// whole tree is the first event, in order of distance, at which the root's value - a boolean
// expression of the leaves' inside bits - changes; its normal is inverted iff the leaf hit and the
// root disagree about entering.  That needs no hit lists (the generic machine keeps them in
// scratch memory, where lanes with different list lengths hit different cache lines with every
// access): leaf events in registers, one bit of state per leaf.  Returns 0: no hit, 1: hit,
// 2: not applicable - equal distances (their order of processing matters), a leaf whose two hits do
// not alternate between entering and leaving, or more than RT_CSG_FAST_LEAVES leaves - in which
// case the caller runs the generic machine.
#define RT_CSG_FAST_LEAVES 4
RT_DEV bool csg_value(const DScene& sc, const DCsgRoot& root, uint32_t inside_bits) {
    uint32_t stk = 0;                                              // a stack of booleans, top in bit 0
    for (uint32_t oi = 0; oi < root.op_count; ++oi) {
        const DCsgOp& op = sc.csg_ops[root.op_begin + oi];
        if (op.kind == DC_BASE) { stk = (stk << 1) | ((inside_bits >> (op.leaf - root.leaf_begin)) & 1u); continue; }
        if (op.kind == DC_MULTI_AND) {
            const uint32_t k = op.num_children, all = (1u << k) - 1u;
            stk = ((stk >> k) << 1) | ((stk & all) == all ? 1u : 0u);
            continue;
        }
        const uint32_t b = stk & 1u, a = (stk >> 1) & 1u;          // children in order: a below b
        uint32_t v;
        switch (op.kind) {
            case DC_SUBTRACT: v = a & (b ^ 1u); break;
            case DC_AND: v = a & b; break;
            case DC_OR: v = a | b; break;
            default: v = a ^ b; break;                             // DC_XOR
        }
        stk = ((stk >> 2) << 1) | v;
    }
    return (stk & 1u) != 0u;
}
template <uint32_t F, bool COUNT>
RT_DEV int csg_closest_fast(const DScene& sc, const DCsgRoot& root, const LRay& r, Cand& out, Work<COUNT>& wk) {
    constexpr int L = RT_CSG_FAST_LEAVES;
    if (root.num_leaves > (uint32_t)L) return 2;
    double et[2 * L];                                              // event distances in the composite's parent space (+inf: none)
    uint32_t em[2 * L];                                            // bit 0: enters its leaf, bit 2: the leaf candidate's b bit 0
    uint32_t inside_bits = 0;
    bool odd = false;
#pragma unroll
    for (int k = 0; k < 2 * L; ++k) { et[k] = INFINITY; em[k] = 0; }
    for (uint32_t j = 0; j < root.num_leaves; ++j) {               // BaseShape::get_all_intersects, base_shape.rs:49-53
        const DCsgLeaf& leaf = sc.csg_leaves[root.leaf_begin + j];
        const DPrim& p = sc.prims[leaf.prim];
        double s, inv_s;
        LRay lr = to_local(leaf.x, r.o, r.d, s, inv_s);
        Cand c[2];
        RT_COUNT(wk, instance_tests);                                  // (a leaf: ray into its space + primitive test, like a node's)
        int n = basic_hits<F, COUNT>(sc, p, lr, INFINITY, c, wk);
        if constexpr (F & RTF_WRAP) { if (p.wrap) n = wrap_all<F>(sc, p, lr, c, n); }
        double t0 = INFINITY, t1 = INFINITY;
        uint32_t m0 = 0, m1 = 0;
        bool enter0 = false;
        if (n > 0) {
            enter0 = facing<F>(sc, p, lr, c[0]) < 0.0;                // hit_direction, composite_shape.rs:27-30
            t0 = c[0].t * inv_s; m0 = (enter0 ? 1u : 0u) | ((c[0].b & 1u) << 2);
            if (!enter0) inside_bits |= 1u << j;                   // initial inside-state from the first hit, :99-109
        }
        if (n > 1) {
            const bool enter1 = facing<F>(sc, p, lr, c[1]) < 0.0;
            t1 = c[1].t * inv_s; m1 = (enter1 ? 1u : 0u) | ((c[1].b & 1u) << 2);
            if (enter1 == enter0) odd = true;
        }
#pragma unroll
        for (int k = 0; k < L; ++k)                                 // (register arrays: written by selection, never by a run-time index)
            if ((uint32_t)k == j) { et[2 * k] = t0; et[2 * k + 1] = t1; em[2 * k] = m0; em[2 * k + 1] = m1; }
    }
    if (odd) return 2;
    bool value = csg_value(sc, root, inside_bits);
    double last = -INFINITY;
    for (int step = 0; step < 2 * L; ++step) {
        double mt = INFINITY;
        uint32_t mk = 0, mm = 0, same = 0;
#pragma unroll
        for (int k = 0; k < 2 * L; ++k) {
            const bool take = et[k] > last && et[k] < mt;
            same = take ? 0u : (et[k] == mt && et[k] < INFINITY ? same + 1u : same);
            mk = take ? (uint32_t)k : mk; mm = take ? em[k] : mm; mt = take ? et[k] : mt;
        }
        if (!(mt < INFINITY)) return 0;
        if (same) return 2;                                        // two events at one distance
        RT_COUNT(wk, csg_events);
        inside_bits ^= 1u << (mk >> 1);
        const bool now = csg_value(sc, root, inside_bits);
        if (now != value) {
            const uint32_t inverted = ((mm & 1u) != 0u) != now ? 1u : 0u;
            out.t = mt;
            out.a = root.leaf_begin + (mk >> 1);
            out.b = inverted | (((mm >> 2) & 1u) << 1) | ((mk & 1u) << 8);
            return 1;
        }
        last = mt;
    }
    return 0;
}

template <uint32_t F, bool COUNT>
RT_DEV void csg_details(const DScene& sc, const LRay& r, const Cand& c, Details& o, Work<COUNT>& wk) {
    const DCsgLeaf& leaf = sc.csg_leaves[c.a];
    const DPrim& p = sc.prims[leaf.prim];
    double s, inv_s;
    LRay lr = to_local(leaf.x, r.o, r.d, s, inv_s);
    Cand lc[2];
    Work<false> none;
    int n = basic_hits<F, false>(sc, p, lr, INFINITY, lc, none);
    if constexpr (F & RTF_WRAP) { if (p.wrap) n = wrap_all<F>(sc, p, lr, lc, n); }
    uint32_t idx = c.b >> 8;
    Details d;
    basic_details<F>(sc, p, lr, lc[idx < (uint32_t)n ? idx : 0], d);
    o.p = xf_point(leaf.x.l2w, d.p);
    o.n = xf_normal(leaf.x.nmat, d.n);
    if (c.b & 1u) o.n = neg(o.n);
    o.tangent = xf_vector(leaf.x.l2w, d.tangent);
    o.u = d.u; o.v = d.v;
}

// ------------------------------------------------------------------ exact ties (src/scene.rs:196-225)
// The distance SceneNode::trace compares for the closest hit of instance `i`, at depth `level` of the instance's
// path of nodes: the ray goes down through every node's inverse transform and is renormalised at each
// (geometry/ray.rs:90-97), the primitive is intersected in its own space, and the hit comes back up with its
// distance measured again at every level between the transformed hit point and the transformed origin
// (geometry/intersect.rs:31-45).  `level` is the node at which the comparison takes place: the instance's own
// node (the primitive's own distance) or an ancestor of it.  False when the primitive is of a kind this does
// not cover (meshes, composite shapes, wrapped primitives), or is not hit by the ray that went down the chain.
// level -1: the distance the scene's caller sees, above the root node's own transform.  `cand` (optional): the hit
// in the primitive's space, as the loop over instances would record it.
template <uint32_t F>
RT_DEV bool chain_distance(const DScene& sc, d3 o, d3 d, uint32_t i, int level, double& dist, Cand* cand = nullptr) {
    const DChain ch = sc.inst_chain[i];
    const DPrim& p = sc.prims[sc.instances[i].prim];
    if (p.kind == DP_MESH || p.kind == DP_CSG || ((F & RTF_WRAP) && p.wrap)) return false;
    LRay r;
    r.o = o; r.d = d;
    for (uint32_t k = 0; k < ch.len; ++k) {
        const DNodeXf& x = sc.node_xf[sc.chain_nodes[ch.first + k]];
        r.o = xf_point(x.inv, r.o);
        r.d = normalize(xf_vector(x.inv, r.d));
    }
    Cand c[2];
    Work<false> none;
    bool faced = false;
    if constexpr (F & RTF_FACED) faced = p.kind == DP_FACED;
    d3 hit, org;
    if (faced) {
        if constexpr (F & RTF_FACED) {                             // a cube's faces, with the ray renormalised in each face's frame as the reference does
            double t[2]; uint32_t fc[2];
            const DFaced& f = sc.faced[p.ref];
            if (faced_hits<true>(sc, f, r, t, fc) == 0) return false;
            dist = t[0];
            if (cand) { cand->t = t[0]; cand->a = fc[0]; cand->b = 0; }
            Details dt;                                            // the hit point comes back through the face's matrix (cube.rs:44-47)
            faced_details<true>(sc, f, r, fc[0], dt);
            hit = dt.p;
            double fs, finv_s;                                     // ... and so does the ray's origin, which the levels above measure from
            const DXform& fx = f.face[fc[0]];
            org = xf_point(fx.l2w, to_local<true>(fx, r.o, r.d, fs, finv_s).o);
        }
    } else {
        if (basic_hits<F, false>(sc, p, r, INFINITY, c, none) == 0) return false;
        dist = c[0].t;
        if (cand) *cand = c[0];
        hit = r.o + (c[0].t * r.d); org = r.o;
    }
    for (int k = (int)ch.len - 1; k > level; --k) {
        const DNodeXf& x = sc.node_xf[sc.chain_nodes[ch.first + k]];
        hit = xf_point(x.trans, hit);
        org = xf_point(x.trans, org);
        dist = length(hit - org);
    }
    return true;
}
// Two hits whose distances differ by rounding only, on instances `i` (the one being tested) and `j` (the best so
// far): 1 when the reference would keep i's, 0 when j's, -1 when this cannot tell.  The reference compares them at
// their lowest common ancestor, strictly: the later one (DFS pre-order = instance order) wins only when nearer.
template <uint32_t F>
RT_DEV int exact_tie_order(const DScene& sc, d3 o, d3 d, uint32_t i, uint32_t j) {
    const DChain a = sc.inst_chain[i], b = sc.inst_chain[j];
    uint32_t common = 0;
    while (common < a.len && common < b.len && sc.chain_nodes[a.first + common] == sc.chain_nodes[b.first + common]) ++common;
    if (common == 0) return -1;
    double di = 0.0, dj = 0.0;
#pragma unroll 1
    for (int k = 0; k < 2; ++k) {                                  // (one copy of the code for both: this path is rare, its registers are not)
        double dk;
        if (!chain_distance<F>(sc, o, d, k ? j : i, (int)common - 1, dk)) return -1;
        if (k) dj = dk; else di = dk;
    }
    return i > j ? (di < dj ? 1 : 0) : (dj < di ? 0 : 1);
}

// A sub-queue's list of such decisions, in global memory: (query, instance tested, best instance at that moment).
// A full list drops further notes (their decisions stand as the 8-ulp rule made them, and which notes were dropped
// depends on the order the lanes arrived in): sized for the seams of a 5000 x 5000 frame under AA, 256 sub-queues x
// 1024 notes per generation of rays.
#define RT_TIE_LIST 1024u
struct TieLog {
    uint32_t* count;           // entries noted so far (null: nothing is noted)
    uint32_t* entries;         // RT_TIE_LIST x 3 words
    uint32_t query;            // the query the calling lane is working on
    RT_DEV void note(uint32_t i, uint32_t j) {
        const uint32_t at = atomicAdd(count, 1u);
        if (at < RT_TIE_LIST) { entries[3 * at] = query; entries[3 * at + 1] = i; entries[3 * at + 2] = j; }
    }
};

// ------------------------------------------------------------------ scene queries
struct SceneHit {
    double t;            // world distance
    uint32_t inst;
    Cand c;              // in the instance's primitive space
};

// WITH_CSG = false: composite shapes are not evaluated; a composite whose bounding box the ray reaches is
// reported in *deferred (bit min(root index, 31)) for the caller to answer elsewhere, and counts as a miss here.
// DOUBTING: see basic_hits; the caller clears c.b beforehand and finds RT_CAND_DOUBT there afterwards, hit or miss.
template <uint32_t F, bool COUNT, bool WITH_CSG = true, bool DOUBTING = false>
RT_DEV bool prim_closest(const DScene& sc, const DPrim& p, const LRay& lr, double tmax_local, Cand& c, Work<COUNT>& wk, uint32_t* deferred = nullptr) {
    if constexpr (F & RTF_CSG) {
        if (p.kind == DP_CSG) {
            const DCsgRoot& root = sc.csg_roots[p.ref];
            if (root.bounded) {                                       // most rays pass the composite by: no leaf to hit, no lists to merge
                double tn;
                RT_COUNT(wk, box_tests);
                if (!slab(root.lo, root.hi, lr.o, mk3(1.0 / lr.d.x, 1.0 / lr.d.y, 1.0 / lr.d.z), INFINITY, tn)) return false;
            }
            if constexpr (WITH_CSG) {
                const int fast = csg_closest_fast<F, COUNT>(sc, root, lr, c, wk);
                if (fast != 2) return fast == 1;
                return csg_closest<F, COUNT>(sc, root, lr, c, wk);
            } else {
                *deferred |= 1u << min(p.ref, 31u);
                return false;
            }
        }
    }
    Cand cc[2];
    // an un-wrapped mesh may prune by the best distance so far; wrapped primitives must see their true closest hit
    bool wrapped = (F & RTF_WRAP) && p.wrap;
    if constexpr (DOUBTING) cc[0].b = 0;
    int n = basic_hits<F, COUNT, DOUBTING>(sc, p, lr, wrapped ? INFINITY : tmax_local, cc, wk);
    if constexpr (DOUBTING) { if (wrapped) cc[0].b &= ~RT_CAND_DOUBT; }
    if (n == 0) { if constexpr (DOUBTING) c.b = cc[0].b & RT_CAND_DOUBT; return false; }
    c = cc[0];
    if constexpr (F & RTF_WRAP) { if (p.wrap) return wrap_closest<F>(sc, p, lr, c); }
    return true;
}

// One scene query for both kinds of ray:
//   any == false: Scene::cast_ray's root.trace(ray) (src/scene.rs:68, 196-225) — closest hit over
//                 all instances, strict '<', earlier instances (DFS pre-order) win ties;
//   any == true:  SceneNode::partial_trace_until_distance (src/scene.rs:227-248) — does anything
//                 lie closer than max_distance ('<=' for the root's own primitive, '<' below it)?
// With SKIP_MESH the un-wrapped mesh instances are left out: the caller walks them for a
// whole batch of rays at once (mesh_walk_batch / k_wf_walk) and merges the results.
// Scenes with many instances carry a BVH over the instances' world-space boxes (RTF_TLAS); the
// reference tests every child of a node in turn (src/scene.rs:205-217), which only fixes the
// winner among equal distances: the earlier instance, kept here by an explicit tie rule.
// `ties` (optional): closest-hit decisions between two instances whose distances differ by rounding only are noted
// there (TieLog), for the caller to decide again the reference's way once the query is through.
// DOUBTING: an instance whose verdict hung on rounding (basic_hits) is noted as (i, i).  EXCLUDING: the `nexclude`
// instances listed in `exclude` are left out (the query taken again without the instances the second look judges).
template <uint32_t F, bool COUNT, bool SKIP_MESH, bool WITH_CSG = true, class TIES = TieLog, bool DOUBTING = false, bool EXCLUDING = false>
RT_DEV bool trace_scene(const DScene& sc, d3 o, d3 d, bool any, double max_distance, SceneHit& hit, Work<COUNT>& wk, uint32_t* deferred = nullptr, TIES* ties = nullptr,
                        const uint32_t* exclude = nullptr, uint32_t nexclude = 0u) {
    bool found = false;
    double best = any ? max_distance : INFINITY;
    // one instance; returns true when an any-hit query is settled
    auto visit = [&](uint32_t i) -> bool {
        if constexpr (EXCLUDING) { for (uint32_t k = 0; k < nexclude; ++k) if (i == exclude[k]) return false; }
        const DInstance& in = sc.instances[i];
        const DPrim& p = sc.prims[in.prim];
        if constexpr (SKIP_MESH) { if (p.kind == DP_MESH && !((F & RTF_WRAP) && p.wrap)) return false; }
        if (sc.has_clear_to_light && any && (in.is_root & RT_INST_CLEAR_TO_LIGHT)) return false;   // (an earlier revision's shadow test, see DInstance; the scene-wide flag keeps the table read out of every other scene's loop)
        RT_COUNT(wk, instance_tests);
        if (p.kind == DP_RECTANGLE && !((F & RTF_WRAP) && p.wrap)) {
            // A rectangle lies in its frame's z = 0 plane, and the ray meets that plane at -o.z / d.z (rect_hit): the
            // two z components alone (the same expressions to_local evaluates, a seventh of its work) settle most
            // rectangles of a room.  Same sign: the plane lies behind the ray, rect_hit would return false.  Beyond
            // the best distance so far (any-hit: beyond the light) by more than any rounding of the quotient: the
            // hit would lose the comparison that follows.
            const double* mt = in.x.w2l.r;
            const double oz = ((mt[8] * o.x + mt[9] * o.y) + mt[10] * o.z) + mt[11];
            const double vz = (mt[8] * d.x + mt[9] * d.y) + mt[10] * d.z;
            if (oz * vz > 0.0 || fabs(oz) > (best * fabs(vz)) * (1.0 + 1e-9)) return false;
        }
        double s, inv_s;
        LRay lr = to_local(in.x, o, d, s, inv_s);
        Cand c;
        if constexpr (DOUBTING) c.b = 0;
        bool got;
        bool direct_mesh = false;
        if constexpr ((F & RTF_MESH) && !SKIP_MESH) direct_mesh = p.kind == DP_MESH && !((F & RTF_WRAP) && p.wrap);
        if (direct_mesh) {
            if constexpr ((F & RTF_MESH) && !SKIP_MESH) {
                // one call site for both query kinds: shadow lanes and camera/secondary lanes walk together
                c.b = 0;
                got = mesh_walk<(F & RTF_LDS_BVH) != 0, COUNT>(sc, sc.meshes[p.ref], lr, any, best * s, (in.is_root & RT_INST_ROOT) != 0, c.t, c.a, wk);
                if (any) return got;
            }
        } else {
            got = prim_closest<F, COUNT, WITH_CSG, DOUBTING>(sc, p, lr, any ? INFINITY : best * s, c, wk, deferred);
            if constexpr (DOUBTING) {
                if ((c.b & RT_CAND_DOUBT) && ties && ties->count) ties->note(i, i);   // (looked at again by resolve_doubt, or as a shadow test's verdict)
                c.b &= ~RT_CAND_DOUBT;
            }
            if (any) return got && ((in.is_root & RT_INST_ROOT) ? c.t <= max_distance * s : c.t < max_distance * s);
        }
        if (!got) return false;
        double tw = c.t * inv_s;
        if (ties && ties->count && found && near_tie(tw, best, 64.0)) ties->note(i, hit.inst);   // (rare: two surfaces at one distance; looked at again by resolve_tie)
        if (replaces_hit(tw, i, best, hit.inst, found)) { best = tw; hit.inst = i; hit.c = c; found = true; }
        return false;
    };
    bool walked = false;
    if constexpr (F & RTF_TLAS) {
        if (sc.tlas_nodes) {
            walked = true;
            for (uint32_t k = 0; k < sc.num_unbounded; ++k) if (visit(sc.tlas_unbounded[k])) return true;
            const d3 inv = mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
            uint32_t stack[RT_TLAS_STACK];
            int sp = 0;
            uint32_t cur = 0;
            while (cur != RT_WALK_DONE) {
                if (cur & 0x80000000u) {                           // leaf: a few instances
                    uint32_t base = cur & 0x0FFFFFFFu, cnt = (cur >> 28) & 7u;
                    for (uint32_t k = 0; k < cnt; ++k) if (visit(sc.tlas_index[base + k])) return true;
                    cur = sp > 0 ? stack[--sp] : RT_WALK_DONE;
                    continue;
                }
                const DBvhNode& n = sc.tlas_nodes[cur];
                double tn0, tn1;
                RT_COUNT(wk, box_tests); RT_COUNT(wk, box_tests);
                bool h0 = slab(n.lo[0], n.hi[0], o, inv, best, tn0);
                bool h1 = slab(n.lo[1], n.hi[1], o, inv, best, tn1);
                if (h0 && h1) {
                    bool swap = tn1 < tn0;
                    cur = swap ? n.child[1] : n.child[0];
                    if (sp < RT_TLAS_STACK) stack[sp++] = swap ? n.child[0] : n.child[1];
                } else if (h0) cur = n.child[0];
                else if (h1) cur = n.child[1];
                else cur = sp > 0 ? stack[--sp] : RT_WALK_DONE;
            }
        }
    }
    if (!walked) {
        // Few instances, no TLAS: where the host found it worth while (inst_box_mask), an instance's world-space
        // box is tested first.  Changing into a sphere's or a cube's space and testing it costs several times
        // the box test, and most rays miss most boxes (a shadow test ends before it reaches far objects; after
        // the first hit only nearer boxes matter).
        if (sc.inst_box_mask) {
            const d3 inv = mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
            for (uint32_t i = 0; i < sc.num_instances; ++i) {
                if (i < 32u && ((sc.inst_box_mask >> i) & 1u)) {
                    double tn;
                    RT_COUNT(wk, box_tests);
                    if (!slab(sc.inst_box + 6u * i, sc.inst_box + 6u * i + 3u, o, inv, best, tn)) continue;
                }
                if (visit(i)) return true;
            }
        } else {
            for (uint32_t i = 0; i < sc.num_instances; ++i) if (visit(i)) return true;
        }
    }
    hit.t = best;
    return found;
}

// One noted decision, looked at again: instances i (tested later in the loop) and j met at distances that differ by
// rounding only.  `hit` is the query's answer as the loop left it.  When the reference's own comparison
// (exact_tie_order) ranks the two the other way than the loop did, and one of them is the answer, the answer
// changes hands; returns whether it did.
template <uint32_t F>
RT_DEV bool resolve_tie(const DScene& sc, d3 o, d3 d, uint32_t i, uint32_t j, SceneHit& hit) {
    if (hit.inst != i && hit.inst != j) j = hit.inst;              // j has lost to another surface since (a third one at the same distance, or a nearer one): i meets that one
    const int exact = exact_tie_order<F>(sc, o, d, i, j);
    if (exact < 0) return false;
    const uint32_t want = exact ? i : j;
    if (hit.inst == want) return false;
    const DInstance& in = sc.instances[want];                      // its hit as the loop saw it
    double s, inv_s;
    const LRay lr = to_local(in.x, o, d, s, inv_s);
    Work<false> none;
    uint32_t unused = 0;
    Cand c;
    if (!prim_closest<F, false, false>(sc, sc.prims[in.prim], lr, INFINITY, c, none, &unused)) return false;
    hit.inst = want; hit.c = c; hit.t = c.t * inv_s;
    return true;
}

// A query's noted doubts, looked at again: the loop's verdict on each of the instances in `doubted` (triangles,
// polygons or polyhedra met along an edge; ascending, at most RT_DOUBT_SET of them) hung on the last place of the
// ray in the primitive's space, which the loop reaches through one composed matrix and the reference through its
// chain of nodes.  The chain's verdicts stand: the query is taken again without those instances, and each of them
// that the chain's ray hits then enters against the answer the reference's way (exact_tie_order).  The result does
// not depend on the order in which the notes were made.  `found` / `hit`: the query's answer, replaced.
#define RT_DOUBT_SET 4u
template <uint32_t F>
RT_DEV void resolve_doubts(const DScene& sc, d3 o, d3 d, const uint32_t* doubted, uint32_t n, SceneHit& hit, bool& found) {
    const SceneHit before = hit;
    const bool had = found;
    SceneHit again;
    again.t = 0.0; again.inst = 0; again.c.t = 0.0; again.c.a = 0; again.c.b = 0;
    Work<false> none;
    uint32_t unused = 0;
    found = trace_scene<F, false, (F & RTF_MESH) != 0, true, TieLog, false, true>(sc, o, d, false, 0.0, again, none, &unused, nullptr, doubted, n);
    if (found) hit = again;
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t i = doubted[k];
        double world_t = 0.0;
        Cand ce;
        ce.t = 0.0; ce.a = 0; ce.b = 0;
        if (!chain_distance<F>(sc, o, d, i, -1, world_t, &ce)) continue;
        bool wins = !found;
        if (found) {
            const int order = exact_tie_order<F>(sc, o, d, i, hit.inst);
            wins = order < 0 ? replaces_hit(world_t, i, hit.t, hit.inst, true) : order == 1;
        }
        if (wins) { hit.inst = i; hit.c = ce; hit.t = world_t; found = true; }
    }
    // (the same instance and face as the loop's answer: its own numbers stay, which the shading that follows was made for)
    if (found && had && hit.inst == before.inst && hit.c.a == before.c.a) hit = before;
}
// ... and a shadow test's: does anything but the doubted instances lie before the light, or one of them by its chain?
template <uint32_t F>
RT_DEV bool resolve_shadow_doubts(const DScene& sc, d3 o, d3 d, double limit, const uint32_t* doubted, uint32_t n) {
    SceneHit other;
    other.t = 0.0; other.inst = 0; other.c.t = 0.0; other.c.a = 0; other.c.b = 0;
    Work<false> none;
    uint32_t unused = 0;
    if (trace_scene<F, false, (F & RTF_MESH) != 0, true, TieLog, false, true>(sc, o, d, true, limit, other, none, &unused, nullptr, doubted, n)) return true;
    for (uint32_t k = 0; k < n; ++k) {
        double world_t;
        if (chain_distance<F>(sc, o, d, doubted[k], -1, world_t) && ((sc.instances[doubted[k]].is_root & RT_INST_ROOT) ? world_t <= limit : world_t < limit)) return true;
    }
    return false;
}

// A lane's own notes, for a kernel whose lanes decide their ties themselves right after the query (the persistent
// kernel): the first two decisions of the query, in registers.  (The second look is inlined there as well: as a
// called function it made frames of mesh scenes under area lights differ from run to run.)
struct LaneTieLog {
    uint32_t count;            // 0: nothing is noted (shadow tests); else 1 + the notes made
    uint32_t i0, j0, i1, j1;
    RT_DEV void note(uint32_t i, uint32_t j) {
        if (count == 1u) { i0 = i; j0 = j; }
        else if (count == 2u) { i1 = i; j1 = j; }
        ++count;
    }
};
#define RT_TIE_FEATURES(F) ((F) & (RTF_FACED | RTF_MISC | RTF_WRAP))   // what resolve_tie needs of a kernel's features: chain_distance follows no mesh and no composite

// Rebuild the world-space Intersect of the winning hit (geometry/intersect.rs:31-45 applied
// through every level at once).
struct WorldHit {
    d3 p, n, tangent;
    double u, v;
    uint32_t inst;
};
template <uint32_t F, bool COUNT>
RT_DEV void world_hit(const DScene& sc, d3 o, d3 d, const SceneHit& h, WorldHit& w, Work<COUNT>& wk) {
    const DInstance& in = sc.instances[h.inst];
    const DPrim& p = sc.prims[in.prim];
    double s, inv_s;
    LRay lr = to_local(in.x, o, d, s, inv_s);
    Details dt;
    bool done = false;
    if constexpr (F & RTF_CSG) {
        if (p.kind == DP_CSG) { csg_details<F, COUNT>(sc, lr, h.c, dt, wk); done = true; }
    }
    if (!done) basic_details<F>(sc, p, lr, h.c, dt);
    w.p = xf_point(in.x.l2w, dt.p);
    w.n = xf_normal(in.x.nmat, dt.n);
    w.tangent = xf_vector(in.x.l2w, dt.tangent);
    w.u = dt.u; w.v = dt.v;
    w.inst = h.inst;
}
