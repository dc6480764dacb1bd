// rt_flatten.hpp — host side of rt_scene_create: turns the caller's RtSceneDesc
// (a description of the reference's Scene tree, src/scene.rs:47-53,142-148) into the
// flat arrays of rt_device_types.h.  Runs once per scene; nothing here is per ray.
//
// What replaces what:
//   * SceneNode::trace's per-level ray transform (scene.rs:198, ray.rs:90-97) and
//     per-hit `matrix.inverse()` (matrix.rs:55-57) → one composed world<->local pair
//     and a normal matrix per instance, computed here.
//   * The median-split kd-tree of src/mesh.rs:281-332 → a binned-SAH binary BVH whose
//     nodes carry both child boxes (any exact closest-hit structure gives the same
//     hits; the triangle test itself stays mesh.rs:416-466).
//   * Box<dyn Shadable> trees → index-linked DShader records plus, per ChainShader, the
//     flattened list of normal maps its modify_intersect applies (shader.rs:146-152).
//   * Box<dyn Compositable> trees → a postfix program over leaves with composed transforms.
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_device_types.h"

namespace rtflat {

struct Unsupported : std::runtime_error { using std::runtime_error::runtime_error; };
struct Invalid : std::runtime_error { using std::runtime_error::runtime_error; };
struct DeviceFailure : std::runtime_error { using std::runtime_error::runtime_error; };   // a HIP call failed while a scene was prepared on the device

// Puts the calling thread's current HIP device back when it goes out of scope: the entry points that pick a
// device of their own must not leave a caller that works on another GPU (a torch process, say) on theirs.
struct DeviceRestore {
    int saved = -1;
    DeviceRestore() { if (hipGetDevice(&saved) != hipSuccess) saved = -1; }
    ~DeviceRestore() { if (saved >= 0) (void)hipSetDevice(saved); }
    DeviceRestore(const DeviceRestore&) = delete;
    DeviceRestore& operator=(const DeviceRestore&) = delete;
};

// ---- 4x4 column-major helpers (m[c*4+r]) ----
struct Mat4 { double m[16]; };
inline Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r; }
inline Mat4 from_pod(const RtMat4& p) { Mat4 r; std::memcpy(r.m, p.m, sizeof r.m); return r; }
inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = acc;
        }
    return r;
}
inline Mat4 inverse(const Mat4& a) {   // Gauss-Jordan with partial pivoting
    double w[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { w[r][c] = a.m[c * 4 + r]; w[r][4 + c] = r == c ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int r = i + 1; r < 4; ++r) if (std::fabs(w[r][i]) > std::fabs(w[piv][i])) piv = r;
        if (w[piv][i] == 0.0) throw Invalid("singular transform");
        if (piv != i) for (int c = 0; c < 8; ++c) std::swap(w[i][c], w[piv][c]);
        double d = 1.0 / w[i][i];
        for (int c = 0; c < 8; ++c) w[i][c] *= d;
        for (int r = 0; r < 4; ++r) {
            if (r == i) continue;
            double f = w[r][i];
            if (f != 0.0) for (int c = 0; c < 8; ++c) w[r][c] -= f * w[i][c];
        }
    }
    Mat4 out;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out.m[c * 4 + r] = w[r][4 + c];
    return out;
}
// The inverse as the reference computes it (DMat4::inverse through cgmath: the adjugate - every cofactor a 3 x 3
// determinant a(ei - fh) - b(di - fg) + c(dh - eg) - times 1 / det, the determinant expanded along the first row).
// Used for the per-node chain of transforms that decides exact ties (DNodeXf): there the last bit matters.
inline Mat4 inverse_by_cofactors(const Mat4& m) {
    double a[4][4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) a[r][c] = m.m[c * 4 + r];
    auto det3 = [](double p, double q, double r, double s, double t, double u, double v, double w, double x) {
        return p * (t * x - u * w) - q * (s * x - u * v) + r * (s * w - t * v);
    };
    double cof[4][4];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double sub[9];
            int n = 0;
            for (int i = 0; i < 4; ++i) {
                if (i == r) continue;
                for (int j = 0; j < 4; ++j) if (j != c) sub[n++] = a[i][j];
            }
            const double d = det3(sub[0], sub[1], sub[2], sub[3], sub[4], sub[5], sub[6], sub[7], sub[8]);
            cof[r][c] = ((r + c) & 1) ? -d : d;
        }
    const double det = a[0][0] * cof[0][0] + a[0][1] * cof[0][1] + a[0][2] * cof[0][2] + a[0][3] * cof[0][3];
    const double inv_det = 1.0 / det;
    Mat4 out;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out.m[c * 4 + r] = cof[c][r] * inv_det;
    return out;
}
inline DAffine affine(const Mat4& a) {
    DAffine r;
    for (int row = 0; row < 3; ++row) for (int c = 0; c < 4; ++c) r.r[row * 4 + c] = a.m[c * 4 + row];
    return r;
}
inline DXform make_xform(const Mat4& l2w, const Mat4& w2l) {
    DXform x;
    x.w2l = affine(w2l);
    x.l2w = affine(l2w);
    // n_world ∝ (w2l_lin)^T n_local
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) x.nmat[r * 3 + c] = x.w2l.r[c * 4 + r];
    // uniform scale?  L^T L = s^2 I
    double g[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        g[i][j] = 0.0;
        for (int k = 0; k < 3; ++k) g[i][j] += x.w2l.r[k * 4 + i] * x.w2l.r[k * 4 + j];
    }
    double s2 = g[0][0];
    bool uniform = s2 > 0.0;
    for (int i = 0; i < 3 && uniform; ++i) for (int j = 0; j < 3; ++j) {
        double want = i == j ? s2 : 0.0;
        if (std::fabs(g[i][j] - want) > 1e-12 * s2) { uniform = false; break; }
    }
    x.uniform_inv_s = uniform ? 1.0 / std::sqrt(s2) : 0.0;
    x.uniform_s = uniform ? std::sqrt(s2) : 0.0;
    return x;
}
inline Mat4 rot_x(double deg) { double t = deg * (M_PI / 180.0), s = std::sin(t), c = std::cos(t); return Mat4{{1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1}}; }
inline Mat4 rot_y(double deg) { double t = deg * (M_PI / 180.0), s = std::sin(t), c = std::cos(t); return Mat4{{c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1}}; }
inline Mat4 translate(double x, double y, double z) { return Mat4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1}}; }
inline Mat4 scale(double x, double y, double z) { return Mat4{{x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1}}; }

struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double length(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0 / length(a)); }
inline void put(double* d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

// ---- the flattened scene (host copies; uploaded verbatim) ----
constexpr size_t RT_LDS_BLOB_MAX = 160u * 1024u;   // no CU holds more LDS than this: meshes beyond it are never walked out of LDS

// What a bounding-hierarchy build works on and produces: per triangle its box, the box's centre and the face it
// came from (in the order the leaves hold them); per node its box, its children (or -1) and its range of triangles.
struct BuildTri { V3 lo, hi, c; uint32_t face; };
struct BuildNode { V3 lo, hi; int left = -1, right = -1; uint32_t first = 0, count = 0; };
// A mesh build that runs somewhere else (on the device, rt_ingest.hip): same inputs and outputs as the host's
// build_bvh below, and the same tree, node for node.
using MeshBuilder = std::function<void(const double* positions, uint32_t num_positions, const uint32_t* faces, uint32_t num_faces,
                                       int max_depth, std::vector<uint32_t>& face_order, std::vector<BuildNode>& nodes)>;

struct FlatScene {
    std::vector<DInstance> instances;
    std::vector<uint32_t> mesh_instances;
    std::vector<DBvhNode> tlas_nodes;
    std::vector<uint32_t> tlas_index, tlas_unbounded;
    std::vector<double> inst_box;   // without a TLAS: world-space lo, hi of every instance (infinite where it has no finite box)
    uint32_t inst_box_mask = 0;     // ... and which of the first 32 instances are worth the box test
    std::vector<DPrim> prims;
    std::vector<DTri> tris;
    std::vector<DPoly> polys;
    std::vector<double> points;
    std::vector<DFaced> faced;
    std::vector<DMesh> meshes;
    std::vector<DBvhNode> bvh_nodes;
    std::vector<DMeshTri> mesh_tris;
    std::vector<DMeshAttr> mesh_attrs;
    std::vector<DNodeXf> node_xf;           // per scene node
    std::vector<DChain> inst_chain;         // per instance
    std::vector<uint32_t> chain_nodes;
    std::vector<DCsgRoot> csg_roots;
    std::vector<DCsgOp> csg_ops;
    std::vector<DCsgLeaf> csg_leaves;
    std::vector<DShader> shaders;
    std::vector<uint32_t> shader_bounds;
    std::vector<uint32_t> shader_child_idx;
    std::vector<double> shader_child_weight;
    std::vector<uint32_t> chain_nmaps;
    std::vector<DNormalMap> normal_maps;
    std::vector<DImage> images;
    std::vector<uint8_t> texels;
    std::vector<double> gamma_lut;
    std::vector<DLight> lights;
    double ambient[3] = {0, 0, 0};
    uint32_t has_background = 0, background_image = 0;
    DAffine background_w2l{};
    uint32_t max_mesh_bvh_bytes = 0;
    uint32_t features = 0, max_ray_spawn = 0, max_shadow_spawn = 0, inline_point_lights = 0;
    // What the materials in use contain, apart for those that spawn no rays ([0]) and those that do ([1]): bit k of
    // shade_kinds is set when a shader of DShaderKind k occurs in one of them, shade_depth is their deepest
    // nesting of group shaders (0: no groups).  The queue pipeline picks its shade kernels by these.
    uint32_t shade_kinds[2] = {0, 0}, shade_depth[2] = {0, 0};
    MeshBuilder mesh_builder;        // empty: the host builds the meshes' hierarchies
    bool flip_on_device = false;     // images are staged top row first and flipped once they are in device memory
    std::vector<uint8_t> lds_blob;   // LBvhNode + LMeshTri arrays of all meshes
    bool lds_exact = true;           // every mesh coordinate is an f32 value, so the f32 vertex copies lose nothing
};

// ---- BVH build: binned SAH, leaves of <= 2 triangles (measured: 4 costs 2 % more, 1 costs 18 % more on the north-star scene) ----

// `max_depth`: the walk that will use the tree keeps one stack entry per level, so a tree may not grow
// deeper than that stack; a subtree that reaches the limit becomes a leaf if its count still fits the
// 3-bit leaf count of a child reference, else the scene is refused.
inline void build_bvh(std::vector<BuildTri>& tris, std::vector<BuildNode>& nodes, int node_idx, uint32_t first, uint32_t count, int depth, int max_depth) {
    const double inf = std::numeric_limits<double>::infinity();
    V3 lo{inf, inf, inf}, hi{-inf, -inf, -inf}, clo = lo, chi = hi;
    for (uint32_t i = first; i < first + count; ++i) {
        const BuildTri& t = tris[i];
        lo = {std::min(lo.x, t.lo.x), std::min(lo.y, t.lo.y), std::min(lo.z, t.lo.z)};
        hi = {std::max(hi.x, t.hi.x), std::max(hi.y, t.hi.y), std::max(hi.z, t.hi.z)};
        clo = {std::min(clo.x, t.c.x), std::min(clo.y, t.c.y), std::min(clo.z, t.c.z)};
        chi = {std::max(chi.x, t.c.x), std::max(chi.y, t.c.y), std::max(chi.z, t.c.z)};
    }
    nodes[node_idx].lo = lo; nodes[node_idx].hi = hi;
    const uint32_t LEAF = 2;
    if (count > LEAF && depth >= max_depth && count > 7u) throw Unsupported("bounding hierarchy deeper than the device walk stack");
    if (count <= LEAF || depth >= max_depth) { nodes[node_idx].first = first; nodes[node_idx].count = count; return; }
    auto area = [](V3 a, V3 b) { V3 d = b - a; return 2.0 * (d.x * d.y + d.y * d.z + d.z * d.x); };
    const int BINS = 16;
    double best_cost = inf; int best_axis = -1, best_bin = -1;
    for (int axis = 0; axis < 3; ++axis) {
        double a0 = axis == 0 ? clo.x : axis == 1 ? clo.y : clo.z, a1 = axis == 0 ? chi.x : axis == 1 ? chi.y : chi.z;
        if (!(a1 > a0)) continue;
        struct Bin { V3 lo, hi; uint32_t n = 0; } bins[BINS];
        for (auto& b : bins) { b.lo = {inf, inf, inf}; b.hi = {-inf, -inf, -inf}; }
        double k = BINS / (a1 - a0);
        for (uint32_t i = first; i < first + count; ++i) {
            const BuildTri& t = tris[i];
            double c = axis == 0 ? t.c.x : axis == 1 ? t.c.y : t.c.z;
            int b = std::min(BINS - 1, (int)((c - a0) * k));
            bins[b].n++;
            bins[b].lo = {std::min(bins[b].lo.x, t.lo.x), std::min(bins[b].lo.y, t.lo.y), std::min(bins[b].lo.z, t.lo.z)};
            bins[b].hi = {std::max(bins[b].hi.x, t.hi.x), std::max(bins[b].hi.y, t.hi.y), std::max(bins[b].hi.z, t.hi.z)};
        }
        double right_area[BINS]; uint32_t right_n[BINS];
        V3 rl{inf, inf, inf}, rh{-inf, -inf, -inf}; uint32_t rn = 0;
        for (int b = BINS - 1; b > 0; --b) {
            rn += bins[b].n;
            rl = {std::min(rl.x, bins[b].lo.x), std::min(rl.y, bins[b].lo.y), std::min(rl.z, bins[b].lo.z)};
            rh = {std::max(rh.x, bins[b].hi.x), std::max(rh.y, bins[b].hi.y), std::max(rh.z, bins[b].hi.z)};
            right_area[b] = rn ? area(rl, rh) : 0.0; right_n[b] = rn;
        }
        V3 ll{inf, inf, inf}, lh{-inf, -inf, -inf}; uint32_t ln = 0;
        for (int b = 0; b < BINS - 1; ++b) {
            ln += bins[b].n;
            ll = {std::min(ll.x, bins[b].lo.x), std::min(ll.y, bins[b].lo.y), std::min(ll.z, bins[b].lo.z)};
            lh = {std::max(lh.x, bins[b].hi.x), std::max(lh.y, bins[b].hi.y), std::max(lh.z, bins[b].hi.z)};
            if (ln == 0 || right_n[b + 1] == 0) continue;
            double cost = ln * area(ll, lh) + right_n[b + 1] * right_area[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
        }
    }
    uint32_t mid;
    if (best_axis < 0) {   // all centroids coincide: split in the middle
        mid = first + count / 2;
    } else {
        double a0 = best_axis == 0 ? clo.x : best_axis == 1 ? clo.y : clo.z, a1 = best_axis == 0 ? chi.x : best_axis == 1 ? chi.y : chi.z;
        double k = BINS / (a1 - a0);
        auto it = std::stable_partition(tris.begin() + first, tris.begin() + first + count, [&](const BuildTri& t) {
            double c = best_axis == 0 ? t.c.x : best_axis == 1 ? t.c.y : t.c.z;
            return std::min(BINS - 1, (int)((c - a0) * k)) <= best_bin;
        });
        mid = (uint32_t)(it - tris.begin());
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int l = (int)nodes.size(); nodes.emplace_back();
    int r = (int)nodes.size(); nodes.emplace_back();
    nodes[node_idx].left = l; nodes[node_idx].right = r;
    build_bvh(tris, nodes, l, first, mid - first, depth + 1, max_depth);
    build_bvh(tris, nodes, r, mid, first + count - mid, depth + 1, max_depth);
}

inline uint32_t flatten_mesh(FlatScene& fs, const RtSceneDesc& d, const RtMesh& m) {
    if (m.num_faces == 0) throw Invalid("mesh without faces");
    if (m.num_faces >= (1u << 28)) throw Unsupported("mesh with 2^28 or more faces");
    const double* P = d.mesh_f64 + m.positions_off;
    const uint32_t* F = d.mesh_u32 + m.faces_off;
    auto pos = [&](uint32_t i) { if (i >= m.num_positions) throw Invalid("mesh face index out of range"); return V3{P[3 * i], P[3 * i + 1], P[3 * i + 2]}; };
    std::vector<uint32_t> face_order;                            // the faces in the order the leaves hold them
    std::vector<BuildNode> bn(1);
    if (fs.mesh_builder) {
        for (uint32_t f = 0; f < 3 * m.num_faces; ++f) if (F[f] >= m.num_positions) throw Invalid("mesh face index out of range");
        fs.mesh_builder(P, m.num_positions, F, m.num_faces, RT_BVH_STACK - 2, face_order, bn);
    } else {
        std::vector<BuildTri> bt(m.num_faces);
        for (uint32_t f = 0; f < m.num_faces; ++f) {
            V3 a = pos(F[3 * f]), b = pos(F[3 * f + 1]), c = pos(F[3 * f + 2]);
            bt[f].lo = {std::min(a.x, std::min(b.x, c.x)), std::min(a.y, std::min(b.y, c.y)), std::min(a.z, std::min(b.z, c.z))};
            bt[f].hi = {std::max(a.x, std::max(b.x, c.x)), std::max(a.y, std::max(b.y, c.y)), std::max(a.z, std::max(b.z, c.z))};
            bt[f].c = (bt[f].lo + bt[f].hi) * 0.5;
            bt[f].face = f;
        }
        bn.reserve(2 * m.num_faces + 2);
        build_bvh(bt, bn, 0, 0, m.num_faces, 0, RT_BVH_STACK - 2);
        face_order.resize(m.num_faces);
        for (uint32_t i = 0; i < m.num_faces; ++i) face_order[i] = bt[i].face;
    }
    // pad boxes so that slab-test rounding can never cull a triangle lying on a box face
    V3 ext = bn[0].hi - bn[0].lo;
    double pad = 1e-9 * std::max(ext.x, std::max(ext.y, ext.z)) + 1e-300;
    DMesh dm{};
    dm.node_off = (uint32_t)fs.bvh_nodes.size();
    dm.tri_off = (uint32_t)fs.mesh_tris.size();
    dm.num_tris = m.num_faces;
    put(dm.lo, bn[0].lo - V3{pad, pad, pad}); put(dm.hi, bn[0].hi + V3{pad, pad, pad});
    dm.max_abs = 0.0;
    for (int k = 0; k < 3; ++k) dm.max_abs = std::max(dm.max_abs, std::max(std::fabs(dm.lo[k]), std::fabs(dm.hi[k])));
    dm.max_abs *= 1.0 + 1e-6;   // (the f32 copies of the boxes are rounded outward by up to 2^-23 of their value)
    // triangles in BVH order; derived values as Mesh::from_path computes them (mesh.rs:396-404)
    fs.mesh_tris.reserve(fs.mesh_tris.size() + m.num_faces);
    for (uint32_t i = 0; i < m.num_faces; ++i) {
        uint32_t f = face_order[i];
        V3 a = pos(F[3 * f]), b = pos(F[3 * f + 1]), c = pos(F[3 * f + 2]);
        V3 cr = cross(b - a, c - b);
        DMeshTri t{};
        put(t.v1, a); put(t.v2, b); put(t.v3, c); put(t.n, normalize(cr));
        t.inv_area = 1.0 / length(cr);
        t.i1 = F[3 * f]; t.i2 = F[3 * f + 1]; t.i3 = F[3 * f + 2]; t.face = f;
        fs.mesh_tris.push_back(t);
    }
    // interior nodes only; node 0 = root.  A single-leaf mesh gets a root whose second child is empty.
    std::vector<int> interior_index(bn.size(), -1);
    std::vector<int> order;
    for (size_t i = 0; i < bn.size(); ++i) if (bn[i].left >= 0) { interior_index[i] = (int)order.size(); order.push_back((int)i); }
    auto child_ref = [&](int b, uint32_t& ref, uint32_t& cnt) {
        if (bn[b].left >= 0) { ref = (uint32_t)interior_index[b]; cnt = 0; }
        else { ref = 0x80000000u | (bn[b].count << 28) | bn[b].first; cnt = bn[b].count; }
    };
    auto set_box = [&](DBvhNode& n, int k, const BuildNode& b) {
        n.lo[k][0] = b.lo.x - pad; n.lo[k][1] = b.lo.y - pad; n.lo[k][2] = b.lo.z - pad;
        n.hi[k][0] = b.hi.x + pad; n.hi[k][1] = b.hi.y + pad; n.hi[k][2] = b.hi.z + pad;
    };
    if (order.empty()) {
        DBvhNode n{};
        set_box(n, 0, bn[0]);
        n.child[0] = 0x80000000u | (bn[0].count << 28) | bn[0].first; n.count[0] = bn[0].count;
        const double inf = std::numeric_limits<double>::infinity();
        for (int k = 0; k < 3; ++k) { n.lo[1][k] = inf; n.hi[1][k] = -inf; }
        n.child[1] = 0x80000000u; n.count[1] = 0;
        fs.bvh_nodes.push_back(n);
    } else {
        for (int b : order) {
            DBvhNode n{};
            set_box(n, 0, bn[bn[b].left]); set_box(n, 1, bn[bn[b].right]);
            child_ref(bn[b].left, n.child[0], n.count[0]);
            child_ref(bn[b].right, n.child[1], n.count[1]);
            fs.bvh_nodes.push_back(n);
        }
    }
    dm.num_nodes = (uint32_t)fs.bvh_nodes.size() - dm.node_off;
    dm.normal_off = 0; dm.num_normals = m.num_normals;
    dm.uv_off = 0; dm.num_uvs = m.num_texcoords;
    // the corners' normals and texture coordinates per face, in BVH order (a vertex index beyond an array reads as
    // zeros and is never used: the reference indexes with all three once the first is in range, mesh.rs:449-461)
    fs.mesh_attrs.reserve(fs.mesh_attrs.size() + m.num_faces);
    for (uint32_t i = 0; i < m.num_faces; ++i) {
        const DMeshTri& t = fs.mesh_tris[dm.tri_off + i];
        const uint32_t idx[3] = {t.i1, t.i2, t.i3};
        DMeshAttr at{};
        double* nn[3] = {at.n1, at.n2, at.n3};
        if (m.num_normals > t.i1) {
            at.flags |= 1u;
            for (int c = 0; c < 3; ++c) {
                if (idx[c] >= m.num_normals) throw Invalid("mesh face refers to a normal beyond the mesh's normals");
                for (int k = 0; k < 3; ++k) nn[c][k] = d.mesh_f64[m.normals_off + 3 * (size_t)idx[c] + k];
            }
        }
        if (m.num_texcoords > t.i1) {
            at.flags |= 2u;
            for (int c = 0; c < 3; ++c) {
                if (idx[c] >= m.num_texcoords) throw Invalid("mesh face refers to a texture coordinate beyond the mesh's");
                for (int k = 0; k < 2; ++k) at.uv[2 * c + k] = d.mesh_f64[m.texcoords_off + 2 * (size_t)idx[c] + k];
            }
        }
        fs.mesh_attrs.push_back(at);
    }
    // LDS copies: f32 boxes rounded outward, triangles without indices (only while the scene's meshes can still fit
    // a CU's LDS together: a larger scene walks the f64 records in global memory and needs no copies)
    if (fs.lds_blob.size() + ((size_t)dm.num_nodes + dm.num_tris) * 64u > RT_LDS_BLOB_MAX) { fs.lds_exact = false; fs.lds_blob.clear(); }
    else if (fs.lds_exact) {
        auto down = [](double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; };
        auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; };
        while (fs.lds_blob.size() % 16) fs.lds_blob.push_back(0);
        dm.lds_node_off = (uint32_t)fs.lds_blob.size();
        for (uint32_t i = 0; i < dm.num_nodes; ++i) {
            const DBvhNode& n = fs.bvh_nodes[dm.node_off + i];
            LBvhNode l{};
            for (int c = 0; c < 2; ++c) {
                for (int k = 0; k < 3; ++k) { l.lo[c][k] = down(n.lo[c][k]); l.hi[c][k] = up(n.hi[c][k]); }
                l.child[c] = n.child[c];
            }
            const uint8_t* b = reinterpret_cast<const uint8_t*>(&l);
            fs.lds_blob.insert(fs.lds_blob.end(), b, b + sizeof l);
        }
        dm.lds_tri_off = (uint32_t)fs.lds_blob.size();
        fs.lds_blob.reserve(fs.lds_blob.size() + (size_t)dm.num_tris * sizeof(LMeshTri));
        for (uint32_t i = 0; i < dm.num_tris; ++i) {
            const DMeshTri& t = fs.mesh_tris[dm.tri_off + i];
            LMeshTri l{};
            for (int k = 0; k < 3; ++k) {
                l.v1[k] = (float)t.v1[k]; l.v2[k] = (float)t.v2[k]; l.v3[k] = (float)t.v3[k];
                if ((double)l.v1[k] != t.v1[k] || (double)l.v2[k] != t.v2[k] || (double)l.v3[k] != t.v3[k]) fs.lds_exact = false;
            }
            std::memcpy(l.n, t.n, sizeof l.n);
            const uint8_t* b = reinterpret_cast<const uint8_t*>(&l);
            fs.lds_blob.insert(fs.lds_blob.end(), b, b + sizeof l);
        }
    }
    fs.meshes.push_back(dm);
    uint32_t bytes = dm.num_nodes * (uint32_t)sizeof(DBvhNode) + dm.num_tris * (uint32_t)sizeof(DMeshTri);
    fs.max_mesh_bvh_bytes = std::max(fs.max_mesh_bvh_bytes, bytes);
    return (uint32_t)fs.meshes.size() - 1;
}

inline DTri make_tri(V3 v1, V3 v2, V3 v3) {   // Triangle::from_vertices, plane.rs:111-126
    DTri t{};
    V3 e0 = v2 - v1, e1 = v3 - v2, e2 = v1 - v3;
    put(t.v, v1); put(t.v + 3, v2); put(t.v + 6, v3);
    put(t.e, e0); put(t.e + 3, e1); put(t.e + 6, e2);
    V3 a_side = e2 * -1.0, b_side = e0;
    V3 h = a_side - b_side * (dot(a_side, b_side) / dot(b_side, b_side));
    put(t.tangent, normalize(h));
    V3 w = a_side - h;
    t.sc_height = length(h);
    t.sc_width = std::max(length(w), length(b_side));
    put(t.n, normalize(cross(e0, e1)));
    t.side_c = length(e0);
    return t;
}

class Flattener {
public:
    Flattener(const RtSceneDesc& d, FlatScene& fs) : d_(d), fs_(fs), prim_map_(d.num_prims, RT_NONE), mesh_map_(d.num_meshes, RT_NONE) {}

    void run() {
        if (d_.abi_version != RT_ABI_VERSION) throw Invalid("RtSceneDesc.abi_version mismatch");
        if (d_.num_nodes == 0 || !d_.nodes) throw Invalid("scene has no nodes");
        images();
        shaders();
        // lights
        for (uint32_t i = 0; i < d_.num_lights; ++i) {
            const RtLight& l = d_.lights[i];
            if (l.kind > RT_LIGHT_DIRECTION) throw Invalid("unknown light kind");
            DLight o{};
            o.kind = l.kind; o.index = i;
            std::memcpy(o.position, l.position, sizeof o.position);
            std::memcpy(o.color, l.color, sizeof o.color);
            std::memcpy(o.falloff, l.falloff, sizeof o.falloff);
            o.power = l.power; o.size = l.size; o.inv_area = l.size != 0.0 ? 1.0 / (l.size * l.size) : 0.0;
            fs_.lights.push_back(o);
        }
        for (int k = 0; k < 3; ++k) fs_.ambient[k] = d_.ambient_color[k] * d_.ambient_power;   // light.rs:77-79
        if (d_.has_background) {
            if (d_.background_image >= d_.num_images) throw Invalid("background image out of range");
            fs_.has_background = 1; fs_.background_image = d_.background_image;
            fs_.background_w2l = affine(inverse(from_pod(d_.background_trans)));
        }
        // instances: DFS pre-order with composed transforms
        walk(0, identity(), identity());
        analyse();
    }

private:
    const RtSceneDesc& d_;
    FlatScene& fs_;
    std::vector<uint32_t> prim_map_, mesh_map_;

    // Which kernel features the scene needs and how many rays one hit can push.
    uint32_t point_lights_ = 0;
    bool mix_needs_inline(uint32_t i) {   // a Mix whose product has two factors that depend on traced rays
        const RtShader& s = d_.shaders[i];
        bool need = false;
        if (s.kind == RT_SHADER_MIX) {
            int pending = 0;
            for (uint32_t k = 0; k < s.num_children; ++k) {
                uint32_t c = d_.shader_child_idx[s.first_child + k];
                if (spawns(c, 0) || has_phong(c)) ++pending;
            }
            need = pending > 1;
        }
        if (s.kind == RT_SHADER_MIX || s.kind == RT_SHADER_COMPOSITE || s.kind == RT_SHADER_CHAIN)
            for (uint32_t k = 0; k < s.num_children; ++k) need |= mix_needs_inline(d_.shader_child_idx[s.first_child + k]);
        return need;
    }
    bool has_phong(uint32_t i) {
        const RtShader& s = d_.shaders[i];
        if (s.kind == RT_SHADER_PHONG) return point_lights_ > 0;
        if (s.kind == RT_SHADER_MIX || s.kind == RT_SHADER_COMPOSITE) {
            for (uint32_t k = 0; k < s.num_children; ++k) if (has_phong(d_.shader_child_idx[s.first_child + k])) return true;
            return false;
        }
        if (s.kind == RT_SHADER_CHAIN) return has_phong(d_.shader_child_idx[s.first_child + s.num_children - 1]);
        return false;
    }
    uint32_t count_phongs(uint32_t i) {
        const RtShader& s = d_.shaders[i];
        if (s.kind == RT_SHADER_PHONG) return 1;
        if (s.kind == RT_SHADER_MIX || s.kind == RT_SHADER_COMPOSITE) {
            uint32_t n = 0;
            for (uint32_t k = 0; k < s.num_children; ++k) n += count_phongs(d_.shader_child_idx[s.first_child + k]);
            return n;
        }
        if (s.kind == RT_SHADER_CHAIN) return count_phongs(d_.shader_child_idx[s.first_child + s.num_children - 1]);
        return 0;
    }
    // Shader kinds below shader i (collected in `kinds`, one bit per DShaderKind) and the nesting depth of its groups.
    uint32_t material_shape(uint32_t i, uint32_t& kinds) {
        const DShader& s = fs_.shaders[i];
        kinds |= 1u << s.kind;
        if (s.kind != DS_MIX && s.kind != DS_COMPOSITE && s.kind != DS_CHAIN) return 0;
        uint32_t deepest = 0;
        for (uint32_t k = 0; k < s.num_children; ++k) deepest = std::max(deepest, material_shape(fs_.shader_child_idx[s.first_child + k], kinds));
        return deepest + 1;
    }
    void analyse() {
        uint32_t f = 0;
        for (const DPrim& p : fs_.prims) {
            switch (p.kind) {
                case DP_MESH: f |= RTF_MESH; break;
                case DP_CSG: f |= RTF_CSG; break;
                case DP_FACED: f |= RTF_FACED; break;
                case DP_PLANE: case DP_TRIANGLE: case DP_POLYGON: case DP_POLYHEDRON: f |= RTF_MISC; break;
                default: break;
            }
            if (p.wrap) f |= RTF_WRAP;
        }
        if (!fs_.images.empty()) f |= RTF_IMAGES;
        point_lights_ = 0;
        for (const DLight& l : fs_.lights) { if (l.kind == RT_LIGHT_POINT) ++point_lights_; else f |= RTF_INLINE_LIGHTS; }
        bool inline_points = false;
        uint32_t max_r = 0, max_s = 0;
        for (const DInstance& in : fs_.instances) {
            if (mix_needs_inline(in.shader)) inline_points = true;
            uint32_t r = count_spawns(in.shader), s = count_phongs(in.shader) * point_lights_;
            max_r = std::max(max_r, r); max_s = std::max(max_s, s);
        }
        if (max_r + max_s > RT_SHADE_MAX_SPAWN) inline_points = true;
        if (inline_points) { f |= RTF_INLINE_LIGHTS; max_s = 0; }
        for (uint32_t i = 0; i < fs_.instances.size(); ++i) {
            const DPrim& p = fs_.prims[fs_.instances[i].prim];
            if (p.kind == DP_MESH && !p.wrap) fs_.mesh_instances.push_back(i);
        }
        build_tlas();
        build_instance_boxes((f & RTF_INLINE_LIGHTS) != 0);
        if (!fs_.tlas_nodes.empty()) f |= RTF_TLAS;
        fs_.shader_bounds.assign(d_.num_shaders, 0u);
        for (uint32_t i = 0; i < d_.num_shaders; ++i) {
            uint32_t r = std::min<uint32_t>(count_spawns(i), RT_SHADE_MAX_SPAWN);
            uint32_t sh = inline_points ? 0u : std::min<uint32_t>(count_phongs(i) * point_lights_, RT_SHADE_MAX_SPAWN);
            fs_.shader_bounds[i] = r | (sh << 16);
        }
        for (const DInstance& in : fs_.instances) {
            const int cls = (fs_.shader_bounds[in.shader] & 0xFFFFu) ? 1 : 0;
            fs_.shade_depth[cls] = std::max(fs_.shade_depth[cls], material_shape(in.shader, fs_.shade_kinds[cls]));
        }
        fs_.features = f; fs_.max_ray_spawn = max_r; fs_.max_shadow_spawn = max_s; fs_.inline_point_lights = inline_points ? 1 : 0;
    }

    // Local-space box of a primitive; false when it has none (infinite or not worth bounding).
    bool prim_bounds(const DPrim& p, V3& lo, V3& hi) const {
        switch (p.kind) {
            case DP_SPHERE: { double r = std::fabs(p.p[0]); lo = {-r, -r, -r}; hi = {r, r, r}; return true; }
            case DP_RECTANGLE: { double w = std::fabs(p.p[0]) / 2, h = std::fabs(p.p[1]) / 2; lo = {-w, -h, 0.0}; hi = {w, h, 0.0}; return true; }
            case DP_TRIANGLE: {
                const DTri& t = fs_.tris[p.ref];
                lo = hi = V3{t.v[0], t.v[1], t.v[2]};
                for (int k = 1; k < 3; ++k) {
                    lo = {std::min(lo.x, t.v[3 * k]), std::min(lo.y, t.v[3 * k + 1]), std::min(lo.z, t.v[3 * k + 2])};
                    hi = {std::max(hi.x, t.v[3 * k]), std::max(hi.y, t.v[3 * k + 1]), std::max(hi.z, t.v[3 * k + 2])};
                }
                return true;
            }
            case DP_FACED: {   // cube: +-length/2; tetrahedron: every vertex lies within `size` of the centre
                const DFaced& f = fs_.faced[p.ref];
                double r = f.is_cube ? std::fabs(f.base_w) / 2 : 0.0;
                if (!f.is_cube) { const DTri& t = fs_.tris[f.base_tri]; r = std::fabs(t.v[3] - t.v[0]) * 1.01; }
                lo = {-r, -r, -r}; hi = {r, r, r};
                return true;
            }
            case DP_MESH: { const DMesh& m = fs_.meshes[p.ref]; lo = {m.lo[0], m.lo[1], m.lo[2]}; hi = {m.hi[0], m.hi[1], m.hi[2]}; return true; }
            default: return false;   // Plane, Polygon, Polyhedron, composite shapes
        }
    }
    // World-space box of instance i, padded generously (the hit distance is measured in instance space and
    // scaled back, not in world space); false when its primitive has no finite box.
    bool instance_world_box(uint32_t i, V3& blo, V3& bhi) const {
        const DInstance& in = fs_.instances[i];
        V3 lo, hi;
        if (!prim_bounds(fs_.prims[in.prim], lo, hi)) return false;
        const double inf = std::numeric_limits<double>::infinity();
        V3 wlo{inf, inf, inf}, whi{-inf, -inf, -inf};
        const double* m = in.x.l2w.r;
        for (int c = 0; c < 8; ++c) {
            V3 q{c & 1 ? hi.x : lo.x, c & 2 ? hi.y : lo.y, c & 4 ? hi.z : lo.z};
            V3 w{m[0] * q.x + m[1] * q.y + m[2] * q.z + m[3], m[4] * q.x + m[5] * q.y + m[6] * q.z + m[7], m[8] * q.x + m[9] * q.y + m[10] * q.z + m[11]};
            wlo = {std::min(wlo.x, w.x), std::min(wlo.y, w.y), std::min(wlo.z, w.z)};
            whi = {std::max(whi.x, w.x), std::max(whi.y, w.y), std::max(whi.z, w.z)};
        }
        V3 ext = whi - wlo;
        double pad = 1e-6 * std::max(ext.x, std::max(ext.y, ext.z)) + 1e-9 * (std::fabs(wlo.x) + std::fabs(wlo.y) + std::fabs(wlo.z) + std::fabs(whi.x) + std::fabs(whi.y) + std::fabs(whi.z)) + 1e-12;
        blo = wlo - V3{pad, pad, pad}; bhi = whi + V3{pad, pad, pad};
        return true;
    }
    // Scenes too small for a TLAS: the flat loop over the instances skips those whose box the ray misses, where
    // that saves more than the test costs.  In the persistent kernel (scenes whose lights are sampled inside the
    // shader: `every_kind`) an instance costs two dependent table reads before any arithmetic, and its shadow
    // tests end at the light, before the walls: every bounded instance gets the test (measured: light_room
    // 35.5 -> 32 ms).  In the queue pipeline's analytic kernel, which is bound by its arithmetic, a rectangle in a
    // uniformly scaled frame is tested almost as cheaply as its box (the test made glass_monkey's walls 2 %
    // slower): only spheres, cubes, tetrahedra, triangles and anything under a non-uniform scale (whose ray
    // direction has to be normalised again) get it.
    void build_instance_boxes(bool every_kind) {
        fs_.inst_box.clear();
        fs_.inst_box_mask = 0;
        if (!fs_.tlas_nodes.empty()) return;
        const double inf = std::numeric_limits<double>::infinity();
        for (uint32_t i = 0; i < fs_.instances.size(); ++i) {
            const DInstance& in = fs_.instances[i];
            const uint32_t kind = fs_.prims[in.prim].kind;
            const bool costly = every_kind || kind == DP_SPHERE || kind == DP_FACED || kind == DP_TRIANGLE || (in.x.uniform_s == 0.0 && kind != DP_MESH);
            V3 lo, hi;
            if (costly && i < 32 && instance_world_box(i, lo, hi)) fs_.inst_box_mask |= 1u << i;
            else { lo = {-inf, -inf, -inf}; hi = {inf, inf, inf}; }
            for (double v : {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z}) fs_.inst_box.push_back(v);
        }
        if (!fs_.inst_box_mask) fs_.inst_box.clear();
    }
    void build_tlas() {
        if (fs_.instances.size() < RT_TLAS_MIN_INSTANCES) return;
        std::vector<BuildTri> bt;
        for (uint32_t i = 0; i < fs_.instances.size(); ++i) {
            V3 plo, phi;
            if (!instance_world_box(i, plo, phi)) { fs_.tlas_unbounded.push_back(i); continue; }
            BuildTri b;
            b.lo = plo; b.hi = phi; b.c = (b.lo + b.hi) * 0.5; b.face = i;
            bt.push_back(b);
        }
        if (bt.size() < RT_TLAS_MIN_INSTANCES) { fs_.tlas_unbounded.clear(); return; }
        std::vector<BuildNode> bn(1);
        bn.reserve(2 * bt.size() + 2);
        build_bvh(bt, bn, 0, 0, (uint32_t)bt.size(), 0, RT_TLAS_STACK - 2);
        for (const BuildTri& b : bt) fs_.tlas_index.push_back(b.face);
        std::vector<int> interior_index(bn.size(), -1);
        std::vector<int> order;
        for (size_t i = 0; i < bn.size(); ++i) if (bn[i].left >= 0) { interior_index[i] = (int)order.size(); order.push_back((int)i); }
        auto child_ref = [&](int b) -> uint32_t {
            if (bn[b].left >= 0) return (uint32_t)interior_index[b];
            return 0x80000000u | (bn[b].count << 28) | bn[b].first;
        };
        auto set_box = [&](DBvhNode& n, int k, const BuildNode& b) {
            n.lo[k][0] = b.lo.x; n.lo[k][1] = b.lo.y; n.lo[k][2] = b.lo.z;
            n.hi[k][0] = b.hi.x; n.hi[k][1] = b.hi.y; n.hi[k][2] = b.hi.z;
        };
        for (int b : order) {
            DBvhNode n{};
            set_box(n, 0, bn[bn[b].left]); set_box(n, 1, bn[bn[b].right]);
            n.child[0] = child_ref(bn[b].left); n.child[1] = child_ref(bn[b].right);
            fs_.tlas_nodes.push_back(n);
        }
        if (fs_.tlas_nodes.empty()) { fs_.tlas_index.clear(); fs_.tlas_unbounded.clear(); }   // a single leaf: not worth it
    }

    std::vector<uint32_t> path_;       // the nodes from the root to the one being walked
    void walk(uint32_t i, const Mat4& parent_l2w, const Mat4& parent_w2l) {
        if (i >= d_.num_nodes) throw Invalid("node index out of range");
        const RtNode& n = d_.nodes[i];
        Mat4 t = from_pod(n.trans);
        Mat4 l2w = mul(parent_l2w, t);
        Mat4 w2l = mul(inverse(t), parent_w2l);
        if (fs_.node_xf.size() < d_.num_nodes) fs_.node_xf.resize(d_.num_nodes);
        fs_.node_xf[i].trans = affine(t);
        fs_.node_xf[i].inv = affine(inverse_by_cofactors(t));
        path_.push_back(i);
        if (n.prim != RT_NONE) {
            DInstance inst{};
            inst.x = make_xform(l2w, w2l);
            inst.prim = prim(n.prim);
            if (n.shader >= d_.num_shaders) throw Invalid("node shader out of range");
            inst.shader = n.shader;
            inst.node_id = i;
            inst.is_root = i == 0 ? RT_INST_ROOT : 0u;
            // (an earlier revision of the reference let shadow rays through surfaces of opacity BLACK; RtShader.p[5] of a
            // TranslucentShader asks for that, see rt_amd.h)
            const RtShader& sh = d_.shaders[n.shader];
            if (sh.kind == RT_SHADER_TRANSLUCENT && sh.p[5] != 0.0) inst.is_root |= RT_INST_CLEAR_TO_LIGHT;
            fs_.instances.push_back(inst);
            fs_.inst_chain.push_back(DChain{(uint32_t)fs_.chain_nodes.size(), (uint32_t)path_.size()});
            fs_.chain_nodes.insert(fs_.chain_nodes.end(), path_.begin(), path_.end());
        }
        if (n.subtree_size == 0 || i + n.subtree_size > d_.num_nodes) throw Invalid("bad subtree_size");
        for (uint32_t c = i + 1; c < i + n.subtree_size; c += d_.nodes[c].subtree_size) walk(c, l2w, w2l);
        path_.pop_back();
    }

    uint32_t prim(uint32_t i) {
        if (i >= d_.num_prims) throw Invalid("primitive index out of range");
        if (prim_map_[i] != RT_NONE) return prim_map_[i];
        const RtPrim& p = d_.prims[i];
        DPrim o{};
        switch (p.kind) {
            case RT_PRIM_SPHERE: o.kind = DP_SPHERE; o.p[0] = p.p[0]; break;
            case RT_PRIM_RECTANGLE: o.kind = DP_RECTANGLE; o.p[0] = p.p[0]; o.p[1] = p.p[1]; break;
            case RT_PRIM_PLANE: o.kind = DP_PLANE; std::memcpy(o.p, p.p, sizeof o.p); break;
            case RT_PRIM_TRIANGLE:
                o.kind = DP_TRIANGLE; o.ref = (uint32_t)fs_.tris.size();
                fs_.tris.push_back(make_tri({p.p[0], p.p[1], p.p[2]}, {p.p[3], p.p[4], p.p[5]}, {p.p[6], p.p[7], p.p[8]}));
                break;
            case RT_PRIM_POLYGON: {   // Polygon::from_vertices, plane.rs:181-202
                if (p.count < 3 || (uint64_t)p.ref + p.count > d_.num_points) throw Invalid("polygon vertices out of range");
                std::vector<V3> v(p.count);
                for (uint32_t k = 0; k < p.count; ++k) v[k] = {d_.points[(p.ref + k) * 3], d_.points[(p.ref + k) * 3 + 1], d_.points[(p.ref + k) * 3 + 2]};
                std::vector<V3> e;
                e.push_back(v[0] - v[p.count - 1]);
                e.push_back(v[1] - v[0]);
                V3 n = normalize(cross(e[0], e[1]));
                for (uint32_t k = 2; k + 1 < p.count; ++k) v[k] = v[k] - n * dot(v[k] - v[0], n);
                for (uint32_t k = 2; k < p.count; ++k) e.push_back(v[k] - v[(k - 1) % p.count]);
                DPoly poly{};
                poly.count = p.count; put(poly.n, n);
                poly.first_vertex = (uint32_t)(fs_.points.size() / 3);
                for (auto& q : v) { fs_.points.push_back(q.x); fs_.points.push_back(q.y); fs_.points.push_back(q.z); }
                poly.first_edge = (uint32_t)(fs_.points.size() / 3);
                for (auto& q : e) { fs_.points.push_back(q.x); fs_.points.push_back(q.y); fs_.points.push_back(q.z); }
                o.kind = DP_POLYGON; o.ref = (uint32_t)fs_.polys.size();
                fs_.polys.push_back(poly);
                break;
            }
            case RT_PRIM_CUBE: {      // Cube::new, cube.rs:14-31
                double length = p.p[0];
                DFaced f{};
                f.nfaces = 6; f.base_is_triangle = 0; f.is_cube = 1; f.base_w = length; f.base_h = length;
                Mat4 t = translate(0.0, 0.0, length / 2.0);
                Mat4 m[6] = {mul(rot_y(90.0), t), mul(rot_y(-90.0), t), t, mul(rot_y(180.0), t), mul(rot_x(-90.0), t), mul(rot_x(90.0), t)};
                for (int k = 0; k < 6; ++k) f.face[k] = make_xform(m[k], inverse(m[k]));
                o.kind = DP_FACED; o.ref = (uint32_t)fs_.faced.size();
                fs_.faced.push_back(f);
                break;
            }
            case RT_PRIM_TETRAHEDRON: {   // Tetrahedron::new, cube.rs:103-122
                double size = p.p[0];
                double tri_height = size * std::sin(M_PI / 3.0);
                DFaced f{};
                f.nfaces = 4; f.base_is_triangle = 1; f.is_cube = 0;
                f.base_tri = (uint32_t)fs_.tris.size();
                fs_.tris.push_back(make_tri({-size / 2.0, 0.0, 0.0}, {size / 2.0, 0.0, 0.0}, {0.0, 0.0, -tri_height}));
                Mat4 t = translate(0.0, 0.0, tri_height / 3.0);
                Mat4 m1 = mul(t, rot_x(std::acos(1.0 / 3.0) * (180.0 / M_PI)));
                Mat4 m[4] = {mul(scale(1.0, -1.0, 1.0), t), m1, mul(rot_y(-120.0), m1), mul(rot_y(120.0), m1)};
                for (int k = 0; k < 4; ++k) f.face[k] = make_xform(m[k], inverse(m[k]));
                o.kind = DP_FACED; o.ref = (uint32_t)fs_.faced.size();
                fs_.faced.push_back(f);
                break;
            }
            case RT_PRIM_POLYHEDRON: {
                if (p.count == 0 || p.count > 32 || (uint64_t)p.ref + p.count > d_.num_prims) throw Invalid("polyhedron planes out of range");
                // planes must be contiguous in the device array: emit them now
                uint32_t first = (uint32_t)fs_.prims.size();
                for (uint32_t k = 0; k < p.count; ++k) {
                    const RtPrim& pl = d_.prims[p.ref + k];
                    if (pl.kind != RT_PRIM_PLANE) throw Invalid("polyhedron entry is not a plane");
                    DPrim q{}; q.kind = DP_PLANE; std::memcpy(q.p, pl.p, sizeof q.p);
                    fs_.prims.push_back(q);
                }
                o.kind = DP_POLYHEDRON; o.ref = first; o.count = p.count;
                break;
            }
            case RT_PRIM_ONEWAY:
            case RT_PRIM_INVERTED: {
                uint32_t inner = prim(p.ref);
                o = fs_.prims[inner];
                uint32_t levels = 0;
                while (levels < 16 && ((o.wrap >> (2 * levels)) & 3u)) ++levels;
                if (levels >= 8) throw Unsupported("more than 8 nested OneWay/Inverted wrappers");
                if (o.kind == DP_CSG) throw Unsupported("OneWay/Inverted around a composite shape");
                o.wrap |= (p.kind == RT_PRIM_ONEWAY ? DWRAP_ONEWAY : DWRAP_INVERTED) << (2 * levels);
                break;
            }
            case RT_PRIM_MESH:
                if (p.ref >= d_.num_meshes) throw Invalid("mesh index out of range");
                if (mesh_map_[p.ref] == RT_NONE) mesh_map_[p.ref] = flatten_mesh(fs_, d_, d_.meshes[p.ref]);
                o.kind = DP_MESH; o.ref = mesh_map_[p.ref];
                break;
            case RT_PRIM_CSG: {
                if (p.ref >= d_.num_csg) throw Invalid("csg index out of range");
                DCsgRoot root{};
                root.op_begin = (uint32_t)fs_.csg_ops.size();
                uint32_t lists = 0;
                const size_t leaf0 = fs_.csg_leaves.size();
                csg(p.ref, identity(), identity(), 0, lists);
                root.op_count = (uint32_t)fs_.csg_ops.size() - root.op_begin;
                root.leaf_begin = (uint32_t)leaf0; root.num_leaves = (uint32_t)(fs_.csg_leaves.size() - leaf0);
                {   // box around the leaves, padded like the instance boxes of the top-level tree
                    const double inf = std::numeric_limits<double>::infinity();
                    V3 wlo{inf, inf, inf}, whi{-inf, -inf, -inf};
                    bool bounded = fs_.csg_leaves.size() > leaf0;
                    for (size_t li = leaf0; li < fs_.csg_leaves.size() && bounded; ++li) {
                        const DCsgLeaf& leaf = fs_.csg_leaves[li];
                        V3 lo, hi;
                        if (!prim_bounds(fs_.prims[leaf.prim], lo, hi)) { bounded = false; break; }
                        const double* m = leaf.x.l2w.r;
                        for (int c = 0; c < 8; ++c) {
                            V3 q{c & 1 ? hi.x : lo.x, c & 2 ? hi.y : lo.y, c & 4 ? hi.z : lo.z};
                            V3 w{m[0] * q.x + m[1] * q.y + m[2] * q.z + m[3], m[4] * q.x + m[5] * q.y + m[6] * q.z + m[7], m[8] * q.x + m[9] * q.y + m[10] * q.z + m[11]};
                            wlo = {std::min(wlo.x, w.x), std::min(wlo.y, w.y), std::min(wlo.z, w.z)};
                            whi = {std::max(whi.x, w.x), std::max(whi.y, w.y), std::max(whi.z, w.z)};
                        }
                    }
                    if (bounded) {
                        V3 ext = whi - wlo;
                        double pad = 1e-6 * std::max(ext.x, std::max(ext.y, ext.z)) + 1e-9 * (std::fabs(wlo.x) + std::fabs(wlo.y) + std::fabs(wlo.z) + std::fabs(whi.x) + std::fabs(whi.y) + std::fabs(whi.z)) + 1e-12;
                        root.bounded = 1;
                        put(root.lo, wlo - V3{pad, pad, pad}); put(root.hi, whi + V3{pad, pad, pad});
                    }
                }
                o.kind = DP_CSG; o.ref = (uint32_t)fs_.csg_roots.size();
                fs_.csg_roots.push_back(root);
                break;
            }
            default: throw Invalid("unknown primitive kind");
        }
        uint32_t idx = (uint32_t)fs_.prims.size();
        fs_.prims.push_back(o);
        prim_map_[i] = idx;
        return idx;
    }

    // Post-order emission; every transform above a leaf is folded into the leaf.
    void csg(uint32_t i, const Mat4& parent_l2w, const Mat4& parent_w2l, int depth, uint32_t& live_lists) {
        if (i >= d_.num_csg) throw Invalid("csg index out of range");
        if (depth > 16) throw Unsupported("composite shape nested deeper than 16");
        const RtCsgNode& n = d_.csg[i];
        Mat4 t = from_pod(n.trans);
        Mat4 l2w = mul(parent_l2w, t), w2l = mul(inverse(t), parent_w2l);
        DCsgOp op{};
        if (n.kind == RT_CSG_BASE) {
            DCsgLeaf leaf{};
            leaf.x = make_xform(l2w, w2l);
            leaf.prim = prim(n.prim);
            if (fs_.prims[leaf.prim].kind == DP_CSG) throw Unsupported("BaseShape wrapping a composite shape");
            op.kind = DC_BASE; op.leaf = (uint32_t)fs_.csg_leaves.size();
            fs_.csg_leaves.push_back(leaf);
            if (++live_lists > RT_CSG_MAX_LISTS) throw Unsupported("composite shape needs too many simultaneous hit lists");
        } else {
            if (n.kind > RT_CSG_MULTI_AND) throw Invalid("unknown csg kind");
            if (n.kind != RT_CSG_MULTI_AND && n.num_children != 2) throw Invalid("binary composite needs two children");
            if (n.num_children == 0 || n.num_children > 8) throw Unsupported("composite with more than 8 children");
            for (uint32_t k = 0; k < n.num_children; ++k) csg(d_.csg_children[n.first_child + k], l2w, w2l, depth + 1, live_lists);
            op.kind = n.kind;   // RT_CSG_* and DC_* share values
            op.num_children = n.num_children;
            live_lists -= n.num_children - 1;
        }
        fs_.csg_ops.push_back(op);
    }

    void images() {
        fs_.gamma_lut.resize(256);
        for (int i = 0; i < 256; ++i) fs_.gamma_lut[i] = std::pow((double)i / 255.0, 2.2);   // color.rs:78-84
        for (uint32_t i = 0; i < d_.num_images; ++i) {
            const RtImage& s = d_.images[i];
            if (s.width < 2 || s.height < 2 || (s.channels != 1 && s.channels != 3)) throw Invalid("image must be at least 2x2 with 1 or 3 channels");
            size_t row = (size_t)s.width * s.channels;
            if (s.offset + row * s.height > d_.texel_bytes) throw Invalid("image texels out of range");
            DImage o{};
            o.width = s.width; o.height = s.height; o.channels = s.channels;
            while (fs_.texels.size() % 16) fs_.texels.push_back(0);
            o.offset = fs_.texels.size();
            fs_.texels.resize(o.offset + row * s.height);
            if (fs_.flip_on_device)                   // rows as decoded; the caller flips them after the upload
                std::memcpy(&fs_.texels[o.offset], d_.texels + s.offset, row * s.height);
            else
                for (uint32_t y = 0; y < s.height; ++y)   // flip_vertical: texture.rs:46, normal_map.rs:52,116
                    std::memcpy(&fs_.texels[o.offset + (size_t)y * row], d_.texels + s.offset + (size_t)(s.height - 1 - y) * row, row);
            fs_.images.push_back(o);
        }
        for (uint32_t i = 0; i < d_.num_normal_maps; ++i) {
            const RtNormalMap& s = d_.normal_maps[i];
            if (s.image >= d_.num_images) throw Invalid("normal map image out of range");
            if ((s.kind == RT_NMAP_NORMAL && d_.images[s.image].channels != 3) || (s.kind == RT_NMAP_BUMP && d_.images[s.image].channels != 1))
                throw Invalid("normal map image has the wrong channel count");
            fs_.normal_maps.push_back(DNormalMap{s.kind, s.image, s.depth});
        }
    }

    void collect_chain_nmaps(uint32_t i, int depth) {   // ChainShader::modify_intersect, shader.rs:146-152
        if (depth > 16) throw Unsupported("shader chain nested deeper than 16");
        const RtShader& s = d_.shaders[i];
        if (s.kind == RT_SHADER_NORMAL_MAP) fs_.chain_nmaps.push_back(s.ref);
        else if (s.kind == RT_SHADER_CHAIN)
            for (uint32_t k = 0; k < s.num_children; ++k) collect_chain_nmaps(d_.shader_child_idx[s.first_child + k], depth + 1);
    }
    // true if evaluating the shader can spawn secondary rays
    bool spawns(uint32_t i, int depth) {
        if (depth > RT_SHADE_MAX_FRAMES) throw Unsupported("shader tree nested deeper than the interpreter's frame stack");
        const RtShader& s = d_.shaders[i];
        switch (s.kind) {
            case RT_SHADER_REFLECTION: case RT_SHADER_TRANSLUCENT: return true;
            case RT_SHADER_MIX: {
                int n = 0;
                for (uint32_t k = 0; k < s.num_children; ++k) n += spawns(d_.shader_child_idx[s.first_child + k], depth + 1);
                if (n > 1) throw Unsupported("MixShader with more than one ray-spawning factor (product of two ray colours)");
                return n > 0;
            }
            case RT_SHADER_COMPOSITE: {
                bool any = false;
                for (uint32_t k = 0; k < s.num_children; ++k) any |= spawns(d_.shader_child_idx[s.first_child + k], depth + 1);
                return any;
            }
            case RT_SHADER_CHAIN:
                if (s.num_children == 0) throw Invalid("Chain Shader needs at least one shader to function");
                return spawns(d_.shader_child_idx[s.first_child + s.num_children - 1], depth + 1);
            default: return false;
        }
    }
    uint32_t count_spawns(uint32_t i) {
        const RtShader& s = d_.shaders[i];
        switch (s.kind) {
            case RT_SHADER_REFLECTION: return 1;
            case RT_SHADER_TRANSLUCENT: return 2;
            case RT_SHADER_MIX: case RT_SHADER_COMPOSITE: {
                uint32_t n = 0;
                for (uint32_t k = 0; k < s.num_children; ++k) n += count_spawns(d_.shader_child_idx[s.first_child + k]);
                return n;
            }
            case RT_SHADER_CHAIN: return count_spawns(d_.shader_child_idx[s.first_child + s.num_children - 1]);
            default: return 0;
        }
    }
    void shaders() {
        for (uint32_t i = 0; i < d_.num_shaders; ++i) {
            const RtShader& s = d_.shaders[i];
            if (s.kind > RT_SHADER_NORMAL_MAP) throw Invalid("unknown shader kind");
            if ((uint64_t)s.first_child + s.num_children > d_.num_shader_children) throw Invalid("shader children out of range");
            for (uint32_t k = 0; k < s.num_children; ++k)
                if (d_.shader_child_idx[s.first_child + k] >= d_.num_shaders) throw Invalid("shader child out of range");
            if (s.kind == RT_SHADER_TEXTURE && (s.ref >= d_.num_images || d_.images[s.ref].channels != 3)) throw Invalid("texture shader image invalid");
            if (s.kind == RT_SHADER_NORMAL_MAP && s.ref >= d_.num_normal_maps) throw Invalid("normal map index out of range");
        }
        for (uint32_t i = 0; i < d_.num_shaders; ++i) {
            const RtShader& s = d_.shaders[i];
            DShader o{};
            o.kind = s.kind; o.ref = s.ref; o.first_child = s.first_child; o.num_children = s.num_children;
            std::memcpy(o.p, s.p, sizeof o.p);
            if (s.kind == RT_SHADER_CHAIN) {
                o.nm_begin = (uint32_t)fs_.chain_nmaps.size();
                collect_chain_nmaps(i, 0);
                o.nm_count = (uint32_t)fs_.chain_nmaps.size() - o.nm_begin;
            }
            spawns(i, 0);
            if (count_spawns(i) > RT_SHADE_MAX_SPAWN / 2) throw Unsupported("shader spawns more secondary rays per hit than the device path buffers");
            fs_.shaders.push_back(o);
        }
        fs_.shader_child_idx.assign(d_.shader_child_idx, d_.shader_child_idx + d_.num_shader_children);
        fs_.shader_child_weight.assign(d_.shader_child_weight, d_.shader_child_weight + d_.num_shader_children);
    }
};

inline void flatten(const RtSceneDesc& d, FlatScene& fs) { Flattener(d, fs).run(); }

}  // namespace rtflat
