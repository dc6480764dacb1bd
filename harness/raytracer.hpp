// raytracer.hpp — host-side mirror of the reference crate's public API
// (src/lib.rs:24-38 re-exports), written in C++ because no Rust toolchain exists
// where this is built.  Scene scripts written against it read like the reference's
// tests/*.rs: same type names, same helper names (src/util.rs), same argument
// meaning.  The one addition is `describe()`: every trait object can write itself
// into a flat RtSceneDesc (include/rt_amd.h), which is what crosses the C ABI.
//
// Nothing here traces rays.  All ray work happens behind rt_render().
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "rt_amd.h"

namespace raytracer {

// ---------------------------------------------------------------- euler / cgmath subset
struct DVec2 { double x, y; };
struct DVec3 {
    double x, y, z;
    DVec3 operator+(DVec3 o) const { return {x + o.x, y + o.y, z + o.z}; }
    DVec3 operator-(DVec3 o) const { return {x - o.x, y - o.y, z - o.z}; }
    DVec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    double dot(DVec3 o) const { return x * o.x + y * o.y + z * o.z; }
    DVec3 cross(DVec3 o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double length() const { return std::sqrt(dot(*this)); }
    DVec3 normalize() const { double s = 1.0 / length(); return {x * s, y * s, z * s}; }
    double angle(DVec3 o) const { return std::acos(dot(o) / (length() * o.length())); }
};
inline DVec3 dvec3(double x, double y, double z) { return {x, y, z}; }
inline DVec2 dvec2(double x, double y) { return {x, y}; }

// Column-major 4x4, element (row r, col c) at m[c*4+r]; `dmat4(...)` takes the 16
// values in the same order as the reference's dmat4! macro.
struct DMat4 {
    double m[16];
    static DMat4 identity() {
        DMat4 r{};
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
        return r;
    }
    DMat4 operator*(const DMat4& b) const {
        DMat4 r{};
        for (int c = 0; c < 4; ++c)
            for (int row = 0; row < 4; ++row) {
                // cgmath: result column c = self * b.column(c) = sum_k self.col(k) * b[c][k]
                double acc = m[0 * 4 + row] * b.m[c * 4 + 0];
                acc = acc + m[1 * 4 + row] * b.m[c * 4 + 1];
                acc = acc + m[2 * 4 + row] * b.m[c * 4 + 2];
                acc = acc + m[3 * 4 + row] * b.m[c * 4 + 3];
                r.m[c * 4 + row] = acc;
            }
        return r;
    }
    RtMat4 pod() const { RtMat4 r; std::memcpy(r.m, m, sizeof m); return r; }
};

enum class Axis { X, Y, Z };

namespace matrix {
inline double to_radians(double deg) { return deg * (3.14159265358979323846 / 180.0); }
inline double to_degrees(double rad) { return rad * (180.0 / 3.14159265358979323846); }

// src/geometry/matrix.rs:63-83
inline DMat4 rotation(Axis axis, double degree) {
    double t = to_radians(degree), s = std::sin(t), c = std::cos(t);
    switch (axis) {
        case Axis::X: return DMat4{{1, 0, 0, 0, 0, c, s, 0, 0, -s, c, 0, 0, 0, 0, 1}};
        case Axis::Y: return DMat4{{c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, 0, 0, 0, 1}};
        default:      return DMat4{{c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    }
}
// src/geometry/matrix.rs:85-90
inline DMat4 translation(double x, double y, double z) {
    return DMat4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1}};
}
// src/geometry/matrix.rs:92-97
inline DMat4 scaling(double x, double y, double z) {
    return DMat4{{x, 0, 0, 0, 0, y, 0, 0, 0, 0, z, 0, 0, 0, 0, 1}};
}
// src/geometry/matrix.rs:99-110
inline DMat4 reflection(Axis axis) {
    double x = 1, y = 1, z = 1;
    if (axis == Axis::X) x = -x; else if (axis == Axis::Y) y = -y; else z = -z;
    return scaling(x, y, z);
}
}  // namespace matrix
using namespace matrix;

// ---------------------------------------------------------------- Color (src/color.rs)
struct Color {
    double red, green, blue;
    static Color new_(double r, double g, double b) { return {r, g, b}; }
    static Color from_f64(double v) { return {v, v, v}; }
    Color operator*(double s) const { return {red * s, green * s, blue * s}; }
    Color operator*(Color o) const { return {red * o.red, green * o.green, blue * o.blue}; }
    Color operator+(Color o) const { return {red + o.red, green + o.green, blue + o.blue}; }
    Color operator-(Color o) const { return {red - o.red, green - o.green, blue - o.blue}; }
    // src/color.rs:59-66
    Color normalize() const {
        double mx = std::fmax(std::fmax(red, green), blue);
        return {red / mx, green / mx, blue / mx};
    }
    // the constants the reference's scenes use (src/color/consts.rs)
    static const Color MAROON, FIREBRICK, RED, YELLOW, GREEN, LIME, CYAN, NAVY, BLUE, PURPLE, MAGENTA,
        CHOCOLATE, ROSY_BROWN, MISTY_ROSE, BLACK, ALMOST_BLACK, LIGHT_BLACK, GRAY, WHITE, ORANGE, TEAL, GOLD, BROWN, LAVENDER;
};
inline constexpr Color Color::MAROON{0.5, 0, 0}, Color::FIREBRICK{0.695312, 0.132812, 0.132812},
    Color::RED{0.996094, 0, 0}, Color::YELLOW{0.996094, 0.996094, 0}, Color::GREEN{0, 0.5, 0},
    Color::LIME{0, 0.996094, 0}, Color::CYAN{0, 0.996094, 0.996094}, Color::NAVY{0, 0, 0.5},
    Color::BLUE{0, 0, 0.996094}, Color::PURPLE{0.5, 0, 0.5}, Color::MAGENTA{0.996094, 0, 0.996094},
    Color::CHOCOLATE{0.820312, 0.410156, 0.117188}, Color::ROSY_BROWN{0.734375, 0.558594, 0.558594},
    Color::MISTY_ROSE{0.996094, 0.890625, 0.878906}, Color::BLACK{0, 0, 0},
    Color::ALMOST_BLACK{0.004, 0.004, 0.004}, Color::LIGHT_BLACK{0.1, 0.1, 0.1}, Color::GRAY{0.5, 0.5, 0.5},
    Color::WHITE{1, 1, 1}, Color::ORANGE{0.996094, 0.644531, 0}, Color::TEAL{0, 0.5, 0.5},
    Color::GOLD{0.996094, 0.839844, 0}, Color::BROWN{0.644531, 0.164062, 0.164062}, Color::LAVENDER{0.898438, 0.898438, 0.976562};

// ---------------------------------------------------------------- description builder
// Accumulates the flat arrays of an RtSceneDesc.  Shared sub-objects (the same
// Box used twice, the same image path) are written once.
class SceneDescBuilder {
public:
    std::vector<RtNode> nodes;
    std::vector<RtPrim> prims;
    std::vector<double> points;
    std::vector<RtCsgNode> csg;
    std::vector<uint32_t> csg_children;
    std::vector<RtShader> shaders;
    std::vector<uint32_t> shader_child_idx;
    std::vector<double> shader_child_weight;
    std::vector<RtLight> lights;
    std::vector<RtImage> images;
    std::vector<uint8_t> texels;
    std::vector<RtNormalMap> normal_maps;
    std::vector<RtMesh> meshes;
    std::vector<double> mesh_f64;
    std::vector<uint32_t> mesh_u32;
    std::map<const void*, uint32_t> seen;            // object identity → index
    std::map<std::string, uint32_t> image_by_key;    // "path|channels" → image index
    std::string asset_root;                           // prefix for "assets/..." paths

    uint32_t load_image(const std::string& path, uint32_t channels);
    RtSceneDesc desc() const;   // pointers into this builder; valid while it lives unchanged
    double ambient_color[3] = {0, 0, 0};
    double ambient_power = 0;
    bool has_background = false;
    uint32_t background_image = 0;
    RtMat4 background_trans{};
};

// ---------------------------------------------------------------- Intersectable (src/geometry.rs:15-30)
struct Intersectable {
    virtual ~Intersectable() = default;
    virtual uint32_t describe(SceneDescBuilder& b) const = 0;
};
using IntersectableBox = std::shared_ptr<Intersectable>;

inline uint32_t push_prim(SceneDescBuilder& b, const void* self, RtPrim p) {
    auto it = b.seen.find(self);
    if (it != b.seen.end()) return it->second;
    uint32_t idx = (uint32_t)b.prims.size();
    b.prims.push_back(p);
    b.seen[self] = idx;
    return idx;
}

struct Sphere : Intersectable {
    double radius;
    explicit Sphere(double r) : radius(r) {}
    static std::shared_ptr<Sphere> from_radius(double r) { return std::make_shared<Sphere>(r); }
    static std::shared_ptr<Sphere> from_diameter(double d) { return std::make_shared<Sphere>(d / 2.0); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtPrim p{}; p.kind = RT_PRIM_SPHERE; p.ref = RT_NONE; p.p[0] = radius;
        return push_prim(b, this, p);
    }
};

struct Rectangle : Intersectable {
    double width, height;
    Rectangle(double w, double h) : width(w), height(h) {}
    static std::shared_ptr<Rectangle> new_(double w, double h) { return std::make_shared<Rectangle>(w, h); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtPrim p{}; p.kind = RT_PRIM_RECTANGLE; p.ref = RT_NONE; p.p[0] = width; p.p[1] = height;
        return push_prim(b, this, p);
    }
};

struct Plane : Intersectable {
    DVec3 origin, normal, tangent;
    DVec2 surface_scale;
    // src/primitive/plane.rs:15-35
    static std::shared_ptr<Plane> new_(DVec3 origin, DVec3 normal) {
        DVec3 tangent = dvec3(0.0, 1.0, 0.0);
        if (!(std::fabs(normal.angle(tangent)) > 0.0)) tangent = dvec3(0.0, 0.0, -1.0);
        return with_tangent(origin, normal, tangent);
    }
    static std::shared_ptr<Plane> with_tangent(DVec3 origin, DVec3 normal, DVec3 tangent) {
        return with_surface_scale(origin, normal, tangent, dvec2(1.0, 1.0));
    }
    static std::shared_ptr<Plane> with_surface_scale(DVec3 origin, DVec3 normal, DVec3 tangent, DVec2 scale) {
        if (!(std::fabs(normal.angle(tangent)) > 0.0)) throw std::invalid_argument("Plane: tangent parallel to normal");
        tangent = tangent.cross(normal).normalize();
        tangent = normal.cross(tangent).normalize();
        normal = normal.normalize();
        auto p = std::make_shared<Plane>();
        p->origin = origin; p->normal = normal; p->tangent = tangent; p->surface_scale = scale;
        return p;
    }
    RtPrim pod() const {
        RtPrim p{}; p.kind = RT_PRIM_PLANE; p.ref = RT_NONE;
        p.p[0] = origin.x; p.p[1] = origin.y; p.p[2] = origin.z;
        p.p[3] = normal.x; p.p[4] = normal.y; p.p[5] = normal.z;
        p.p[6] = tangent.x; p.p[7] = tangent.y; p.p[8] = tangent.z;
        p.p[9] = surface_scale.x; p.p[10] = surface_scale.y;
        return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override { return push_prim(b, this, pod()); }
};

struct Triangle : Intersectable {
    DVec3 v[3];
    static std::shared_ptr<Triangle> from_vertices(DVec3 v1, DVec3 v2, DVec3 v3) {
        auto t = std::make_shared<Triangle>(); t->v[0] = v1; t->v[1] = v2; t->v[2] = v3; return t;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtPrim p{}; p.kind = RT_PRIM_TRIANGLE; p.ref = RT_NONE;
        for (int i = 0; i < 3; ++i) { p.p[3 * i] = v[i].x; p.p[3 * i + 1] = v[i].y; p.p[3 * i + 2] = v[i].z; }
        return push_prim(b, this, p);
    }
};

struct Polygon : Intersectable {
    std::vector<DVec3> vertices;
    static std::shared_ptr<Polygon> from_vertices(std::vector<DVec3> v) {
        if (v.size() < 3) throw std::invalid_argument("Polygon needs >= 3 vertices");
        auto p = std::make_shared<Polygon>(); p->vertices = std::move(v); return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        RtPrim p{}; p.kind = RT_PRIM_POLYGON; p.ref = (uint32_t)(b.points.size() / 3); p.count = (uint32_t)vertices.size();
        for (auto& q : vertices) { b.points.push_back(q.x); b.points.push_back(q.y); b.points.push_back(q.z); }
        return push_prim(b, this, p);
    }
};

struct Cube : Intersectable {
    double length;
    explicit Cube(double l) : length(l) {}
    static std::shared_ptr<Cube> new_(double l) { return std::make_shared<Cube>(l); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtPrim p{}; p.kind = RT_PRIM_CUBE; p.ref = RT_NONE; p.p[0] = length;
        return push_prim(b, this, p);
    }
};

struct Tetrahedron : Intersectable {
    double size;
    explicit Tetrahedron(double s) : size(s) {}
    static std::shared_ptr<Tetrahedron> new_(double s) { return std::make_shared<Tetrahedron>(s); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtPrim p{}; p.kind = RT_PRIM_TETRAHEDRON; p.ref = RT_NONE; p.p[0] = size;
        return push_prim(b, this, p);
    }
};

struct Polyhedron : Intersectable {
    std::vector<Plane> planes;
    static std::shared_ptr<Polyhedron> from_planes(std::vector<Plane> planes) {
        auto p = std::make_shared<Polyhedron>(); p->planes = std::move(planes); return p;
    }
    // src/primitive/polyhedron.rs:13-54 (deltohedron :56-97 builds the same eight planes)
    static std::shared_ptr<Polyhedron> octahedron(double size) {
        double h = size / 2.0;
        DVec3 n[8] = {dvec3(-1, 1, 1).normalize(), dvec3(1, 1, 1).normalize(), dvec3(-1, 1, -1).normalize(),
                      dvec3(1, 1, -1).normalize(), dvec3(-1, -1, 1).normalize(), dvec3(1, -1, 1).normalize(),
                      dvec3(-1, -1, -1).normalize(), dvec3(1, -1, -1).normalize()};
        DVec3 o[8] = {dvec3(-h, 0, 0), dvec3(0, 0, h), dvec3(0, 0, -h), dvec3(h, 0, 0),
                      dvec3(0, 0, h), dvec3(h, 0, 0), dvec3(-h, 0, 0), dvec3(0, 0, -h)};
        DVec3 up = dvec3(0, 1, 0), down = dvec3(0, -1, 0);
        DVec2 scale = dvec2(size / std::sqrt(2.0), size / std::sqrt(2.0));
        std::vector<Plane> planes;
        for (int i = 0; i < 8; ++i) planes.push_back(*Plane::with_surface_scale(o[i], n[i], i < 4 ? up : down, scale));
        return from_planes(std::move(planes));
    }
    static std::shared_ptr<Polyhedron> deltohedron(double size) { return octahedron(size); }
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        uint32_t first = (uint32_t)b.prims.size();
        for (auto& pl : planes) b.prims.push_back(pl.pod());
        RtPrim p{}; p.kind = RT_PRIM_POLYHEDRON; p.ref = first; p.count = (uint32_t)planes.size();
        return push_prim(b, this, p);
    }
};

struct OneWay : Intersectable {
    IntersectableBox primitive;
    explicit OneWay(IntersectableBox p) : primitive(std::move(p)) {}
    static std::shared_ptr<OneWay> new_(IntersectableBox p) { return std::make_shared<OneWay>(std::move(p)); }
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        RtPrim p{}; p.kind = RT_PRIM_ONEWAY; p.ref = primitive->describe(b);
        return push_prim(b, this, p);
    }
};

struct Inverted : Intersectable {
    IntersectableBox primitive;
    explicit Inverted(IntersectableBox p) : primitive(std::move(p)) {}
    static std::shared_ptr<Inverted> new_(IntersectableBox p) { return std::make_shared<Inverted>(std::move(p)); }
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        RtPrim p{}; p.kind = RT_PRIM_INVERTED; p.ref = primitive->describe(b);
        return push_prim(b, this, p);
    }
};

// Triangle mesh (src/mesh.rs:342-414).  from_path reads a Wavefront OBJ the way
// tobj 0.1.7's load_obj does: every distinct (v, vt, vn) corner becomes one unified
// vertex, numbered by first appearance; floats are parsed as f32 and widened.
struct Mesh : Intersectable {
    std::vector<double> positions, vertex_normals, tex_coords;  // 3,3,2 per vertex
    std::vector<uint32_t> faces;                                 // 3 per face
    static std::shared_ptr<Mesh> from_path(const std::string& path);
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        RtMesh m{};
        m.num_positions = (uint32_t)(positions.size() / 3);
        m.num_normals = (uint32_t)(vertex_normals.size() / 3);
        m.num_texcoords = (uint32_t)(tex_coords.size() / 2);
        m.num_faces = (uint32_t)(faces.size() / 3);
        m.positions_off = b.mesh_f64.size(); b.mesh_f64.insert(b.mesh_f64.end(), positions.begin(), positions.end());
        m.normals_off = b.mesh_f64.size();   b.mesh_f64.insert(b.mesh_f64.end(), vertex_normals.begin(), vertex_normals.end());
        m.texcoords_off = b.mesh_f64.size(); b.mesh_f64.insert(b.mesh_f64.end(), tex_coords.begin(), tex_coords.end());
        m.faces_off = b.mesh_u32.size();     b.mesh_u32.insert(b.mesh_u32.end(), faces.begin(), faces.end());
        uint32_t mi = (uint32_t)b.meshes.size();
        b.meshes.push_back(m);
        RtPrim p{}; p.kind = RT_PRIM_MESH; p.ref = mi;
        return push_prim(b, this, p);
    }
};

// ---------------------------------------------------------------- Compositable (src/composite_shape.rs:125-127)
struct Compositable : Intersectable {
    DMat4 transform = DMat4::identity();
    void set_transform(DMat4 t) { transform = t; }
    virtual uint32_t describe_csg(SceneDescBuilder& b) const = 0;
    // As a SceneNode primitive the composite is referenced through an RT_PRIM_CSG entry.
    uint32_t describe(SceneDescBuilder& b) const override {
        auto it = b.seen.find(this);
        if (it != b.seen.end()) return it->second;
        RtPrim p{}; p.kind = RT_PRIM_CSG; p.ref = describe_csg(b);
        return push_prim(b, this, p);
    }
};
using CompositableBox = std::shared_ptr<Compositable>;

inline uint32_t push_csg(SceneDescBuilder& b, uint32_t kind, uint32_t prim, const DMat4& t,
                         const std::vector<uint32_t>& kids) {
    RtCsgNode n{}; n.kind = kind; n.prim = prim; n.trans = t.pod();
    n.first_child = (uint32_t)b.csg_children.size(); n.num_children = (uint32_t)kids.size();
    b.csg_children.insert(b.csg_children.end(), kids.begin(), kids.end());
    b.csg.push_back(n);
    return (uint32_t)b.csg.size() - 1;
}

struct BaseShape : Compositable {
    IntersectableBox primitive;
    BaseShape(DMat4 m, IntersectableBox p) : primitive(std::move(p)) { transform = m; }
    static std::shared_ptr<BaseShape> new_(DMat4 m, IntersectableBox p) { return std::make_shared<BaseShape>(m, std::move(p)); }
    uint32_t describe_csg(SceneDescBuilder& b) const override {
        uint32_t p = primitive->describe(b);
        return push_csg(b, RT_CSG_BASE, p, transform, {});
    }
};

template <uint32_t KIND>
struct BinaryShape : Compositable {
    CompositableBox first, second;
    BinaryShape(CompositableBox a, CompositableBox c) : first(std::move(a)), second(std::move(c)) {}
    static std::shared_ptr<BinaryShape> new_(CompositableBox a, CompositableBox c) {
        return std::make_shared<BinaryShape>(std::move(a), std::move(c));
    }
    uint32_t describe_csg(SceneDescBuilder& b) const override {
        uint32_t a = first->describe_csg(b), c = second->describe_csg(b);
        return push_csg(b, KIND, RT_NONE, transform, {a, c});
    }
};
using SubtractShape = BinaryShape<RT_CSG_SUBTRACT>;   // (positive, negative)
using AndShape = BinaryShape<RT_CSG_AND>;
using OrShape = BinaryShape<RT_CSG_OR>;
using XorShape = BinaryShape<RT_CSG_XOR>;

struct MultiAndShape : Compositable {
    std::vector<CompositableBox> shapes;
    static std::shared_ptr<MultiAndShape> from_vec(std::vector<CompositableBox> s) {
        auto m = std::make_shared<MultiAndShape>(); m->shapes = std::move(s); return m;
    }
    uint32_t describe_csg(SceneDescBuilder& b) const override {
        std::vector<uint32_t> kids;
        for (auto& s : shapes) kids.push_back(s->describe_csg(b));
        return push_csg(b, RT_CSG_MULTI_AND, RT_NONE, transform, kids);
    }
};

// ---------------------------------------------------------------- textures / normal maps
struct ImageTexture {
    std::string path;
    static std::shared_ptr<ImageTexture> from_path(const std::string& p) {
        auto t = std::make_shared<ImageTexture>(); t->path = p; return t;
    }
};
struct NormalMappable {
    virtual ~NormalMappable() = default;
    virtual uint32_t describe(SceneDescBuilder& b) const = 0;
};
struct NormalMap : NormalMappable {
    std::string path;
    static std::shared_ptr<NormalMap> from_path(const std::string& p) {
        auto t = std::make_shared<NormalMap>(); t->path = p; return t;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtNormalMap n{}; n.kind = RT_NMAP_NORMAL; n.image = b.load_image(path, 3); n.depth = 0;
        b.normal_maps.push_back(n);
        return (uint32_t)b.normal_maps.size() - 1;
    }
};
struct BumpMap : NormalMappable {
    std::string path; double depth = 1.0;
    static std::shared_ptr<BumpMap> from_path(const std::string& p, double depth) {
        auto t = std::make_shared<BumpMap>(); t->path = p; t->depth = depth; return t;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RtNormalMap n{}; n.kind = RT_NMAP_BUMP; n.image = b.load_image(path, 1); n.depth = depth;
        b.normal_maps.push_back(n);
        return (uint32_t)b.normal_maps.size() - 1;
    }
};

// ---------------------------------------------------------------- Shadable (src/shader.rs:20-30)
struct Shadable {
    virtual ~Shadable() = default;
    virtual uint32_t describe(SceneDescBuilder& b) const = 0;
};
using ShadableBox = std::shared_ptr<Shadable>;

inline uint32_t push_shader(SceneDescBuilder& b, const void* self, RtShader s) {
    uint32_t idx = (uint32_t)b.shaders.size();
    b.shaders.push_back(s);
    b.seen[self] = idx;
    return idx;
}
#define RT_SEEN_OR(b, self)                                \
    do { auto it_ = (b).seen.find(self);                   \
         if (it_ != (b).seen.end()) return it_->second; } while (0)

struct PhongShader : Shadable {
    Color diffuse, specular, ambient; double shininess;
    static std::shared_ptr<PhongShader> new_(Color d, Color s, Color a, double sh) {
        auto p = std::make_shared<PhongShader>(); p->diffuse = d; p->specular = s; p->ambient = a; p->shininess = sh; return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        RtShader s{}; s.kind = RT_SHADER_PHONG; s.ref = RT_NONE;
        const double v[10] = {diffuse.red, diffuse.green, diffuse.blue, specular.red, specular.green, specular.blue,
                              ambient.red, ambient.green, ambient.blue, shininess};
        std::memcpy(s.p, v, sizeof v);
        return push_shader(b, this, s);
    }
};
struct TextureShader : Shadable {
    std::shared_ptr<ImageTexture> texture;
    static std::shared_ptr<TextureShader> new_(std::shared_ptr<ImageTexture> t) {
        auto p = std::make_shared<TextureShader>(); p->texture = std::move(t); return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        RtShader s{}; s.kind = RT_SHADER_TEXTURE; s.ref = b.load_image(texture->path, 3);
        return push_shader(b, this, s);
    }
};
struct ReflectionShader : Shadable {
    Color reflectivity;
    static std::shared_ptr<ReflectionShader> new_(Color r) {
        auto p = std::make_shared<ReflectionShader>(); p->reflectivity = r; return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        RtShader s{}; s.kind = RT_SHADER_REFLECTION; s.ref = RT_NONE;
        s.p[0] = reflectivity.red; s.p[1] = reflectivity.green; s.p[2] = reflectivity.blue;
        return push_shader(b, this, s);
    }
};
struct TranslucentShader : Shadable {
    Color translucency; double refractive_index;
    static std::shared_ptr<TranslucentShader> new_(Color t, double ior) {
        auto p = std::make_shared<TranslucentShader>(); p->translucency = t; p->refractive_index = ior; return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        RtShader s{}; s.kind = RT_SHADER_TRANSLUCENT; s.ref = RT_NONE;
        s.p[0] = translucency.red; s.p[1] = translucency.green; s.p[2] = translucency.blue; s.p[3] = refractive_index;
        return push_shader(b, this, s);
    }
};
inline uint32_t push_group_shader(SceneDescBuilder& b, const void* self, uint32_t kind,
                                  const std::vector<ShadableBox>& kids, const std::vector<double>* weights) {
    std::vector<uint32_t> idx;
    for (auto& k : kids) idx.push_back(k->describe(b));
    RtShader s{}; s.kind = kind; s.ref = RT_NONE;
    s.first_child = (uint32_t)b.shader_child_idx.size(); s.num_children = (uint32_t)idx.size();
    for (size_t i = 0; i < idx.size(); ++i) {
        b.shader_child_idx.push_back(idx[i]);
        b.shader_child_weight.push_back(weights ? (*weights)[i] : 1.0);
    }
    return push_shader(b, self, s);
}
struct MixShader : Shadable {
    std::vector<ShadableBox> shaders;
    static std::shared_ptr<MixShader> from_shaders(std::vector<ShadableBox> s) {
        auto p = std::make_shared<MixShader>(); p->shaders = std::move(s); return p;
    }
    void add_shader(ShadableBox s) { shaders.push_back(std::move(s)); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        return push_group_shader(b, this, RT_SHADER_MIX, shaders, nullptr);
    }
};
struct CompositeShader : Shadable {
    std::vector<std::pair<double, ShadableBox>> shaders;
    static std::shared_ptr<CompositeShader> new_() { return std::make_shared<CompositeShader>(); }
    static std::shared_ptr<CompositeShader> from_shaders(std::vector<std::pair<double, ShadableBox>> s) {
        auto p = std::make_shared<CompositeShader>(); p->shaders = std::move(s); return p;
    }
    void add_shader(double w, ShadableBox s) { shaders.emplace_back(w, std::move(s)); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        std::vector<ShadableBox> kids; std::vector<double> w;
        for (auto& p : shaders) { w.push_back(p.first); kids.push_back(p.second); }
        return push_group_shader(b, this, RT_SHADER_COMPOSITE, kids, &w);
    }
};
struct ChainShader : Shadable {
    std::vector<ShadableBox> shaders;
    static std::shared_ptr<ChainShader> new_() { return std::make_shared<ChainShader>(); }
    static std::shared_ptr<ChainShader> from_shaders(std::vector<ShadableBox> s) {
        auto p = std::make_shared<ChainShader>(); p->shaders = std::move(s); return p;
    }
    void push_shader(ShadableBox s) { shaders.push_back(std::move(s)); }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        if (shaders.empty()) throw std::invalid_argument("Chain Shader needs at least one shader to function");
        return push_group_shader(b, this, RT_SHADER_CHAIN, shaders, nullptr);
    }
};
struct NormalMapShader : Shadable {
    std::shared_ptr<NormalMappable> normal_map;
    static std::shared_ptr<NormalMapShader> new_(std::shared_ptr<NormalMappable> m) {
        auto p = std::make_shared<NormalMapShader>(); p->normal_map = std::move(m); return p;
    }
    uint32_t describe(SceneDescBuilder& b) const override {
        RT_SEEN_OR(b, this);
        RtShader s{}; s.kind = RT_SHADER_NORMAL_MAP; s.ref = normal_map->describe(b);
        return push_shader(b, this, s);
    }
};

// ---------------------------------------------------------------- Lightable (src/light.rs)
struct Lightable {
    virtual ~Lightable() = default;
    virtual RtLight pod() const = 0;
};
using LightableBox = std::shared_ptr<Lightable>;
using Falloff = std::tuple<double, double, double>;

struct AmbientLight {
    Color color; double power;
    static AmbientLight new_(Color c, double p) { return {c, p}; }
};
struct PointLight : Lightable {
    DVec3 position; Color color; double power; Falloff falloff;
    static std::shared_ptr<PointLight> new_(DVec3 pos, Color c, double power, Falloff f) {
        auto p = std::make_shared<PointLight>(); p->position = pos; p->color = c.normalize(); p->power = power; p->falloff = f; return p;
    }
    RtLight pod() const override {
        RtLight l{}; l.kind = RT_LIGHT_POINT;
        l.position[0] = position.x; l.position[1] = position.y; l.position[2] = position.z;
        l.color[0] = color.red; l.color[1] = color.green; l.color[2] = color.blue; l.power = power;
        l.falloff[0] = std::get<0>(falloff); l.falloff[1] = std::get<1>(falloff); l.falloff[2] = std::get<2>(falloff);
        return l;
    }
};
struct SquareLight : Lightable {
    DVec3 position; double size; Color color; double power; Falloff falloff;
    static std::shared_ptr<SquareLight> new_(DVec3 pos, double size, Color c, double power, Falloff f) {
        auto p = std::make_shared<SquareLight>(); p->position = pos; p->size = size; p->color = c.normalize(); p->power = power; p->falloff = f; return p;
    }
    RtLight pod() const override {
        RtLight l{}; l.kind = RT_LIGHT_SQUARE;
        l.position[0] = position.x; l.position[1] = position.y; l.position[2] = position.z;
        l.color[0] = color.red; l.color[1] = color.green; l.color[2] = color.blue; l.power = power;
        l.falloff[0] = std::get<0>(falloff); l.falloff[1] = std::get<1>(falloff); l.falloff[2] = std::get<2>(falloff);
        l.size = size;
        return l;
    }
};
struct DirectionLight : Lightable {
    DVec3 direction; Color color; double power;
    static std::shared_ptr<DirectionLight> new_(DVec3 dir, Color c, double power) {
        auto p = std::make_shared<DirectionLight>(); p->direction = dir.normalize(); p->color = c; p->power = power; return p;
    }
    RtLight pod() const override {
        RtLight l{}; l.kind = RT_LIGHT_DIRECTION;
        l.position[0] = direction.x; l.position[1] = direction.y; l.position[2] = direction.z;
        l.color[0] = color.red; l.color[1] = color.green; l.color[2] = color.blue; l.power = power;
        return l;
    }
};

// ---------------------------------------------------------------- Scene, SceneNode, SkyBox (src/scene.rs)
struct SkyBox {
    std::string path; DMat4 transform;
    static SkyBox from_path(const std::string& p, DMat4 m) { return {p, m}; }
};

struct SceneNode;
using SceneNodeBox = std::shared_ptr<SceneNode>;
using TraceableBox = SceneNodeBox;   // SceneNode is the only Traceable impl

struct SceneNode {
    IntersectableBox primitive;     // Option<Box<Intersectable>>
    ShadableBox material;
    DMat4 transform = DMat4::identity();
    std::vector<SceneNodeBox> children;
    // src/scene.rs:152-161: default material PhongShader(WHITE*0.5, WHITE*0.5, WHITE*0.1, 1.0)
    SceneNode() : material(PhongShader::new_(Color::WHITE * 0.5, Color::WHITE * 0.5, Color::WHITE * 0.1, 1.0)) {}
    static SceneNodeBox new_() { return std::make_shared<SceneNode>(); }
    void set_primitive(IntersectableBox p) { primitive = std::move(p); }
    void set_material(ShadableBox m) { material = std::move(m); }
    void set_transform(DMat4 t) { transform = t; }
    void add_child(SceneNodeBox c) { children.push_back(std::move(c)); }
    SceneNodeBox clone() const { return std::make_shared<SceneNode>(*this); }  // shallow: children shared, immutable
    uint32_t describe(SceneDescBuilder& b, uint32_t parent) const {
        uint32_t me = (uint32_t)b.nodes.size();
        b.nodes.push_back(RtNode{});
        uint32_t prim = primitive ? primitive->describe(b) : RT_NONE;
        uint32_t sh = material->describe(b);
        for (auto& c : children) c->describe(b, me);
        RtNode& n = b.nodes[me];
        n.parent = parent; n.prim = prim; n.shader = sh; n.trans = transform.pod();
        n.subtree_size = (uint32_t)b.nodes.size() - me;
        return me;
    }
};

struct Scene {
    SceneNodeBox root = SceneNode::new_();
    std::vector<LightableBox> lights;
    AmbientLight ambient_light = AmbientLight::new_(Color::BLACK, 0.0);
    bool has_background = false;
    SkyBox background;
    static Scene new_() { return Scene{}; }
    void add_light(LightableBox l) { lights.push_back(std::move(l)); }
    void set_background_from_path(const std::string& p) { background = SkyBox::from_path(p, DMat4::identity()); has_background = true; }
    void set_background(SkyBox s) { background = std::move(s); has_background = true; }

    // Write the whole scene into `b` (the describe() visitor's entry point).
    void describe(SceneDescBuilder& b) const {
        root->describe(b, RT_NONE);
        for (auto& l : lights) b.lights.push_back(l->pod());
        b.ambient_color[0] = ambient_light.color.red; b.ambient_color[1] = ambient_light.color.green;
        b.ambient_color[2] = ambient_light.color.blue; b.ambient_power = ambient_light.power;
        if (has_background) {
            b.has_background = true;
            b.background_image = b.load_image(background.path, 3);
            b.background_trans = background.transform.pod();
        }
    }
};

// ---------------------------------------------------------------- render types (src/render.rs:28-86)
struct RenderConfig {
    size_t num_threads; uint32_t workload_split; bool anti_alias; double aa_threshold; uint32_t aa_rays;
    uint32_t recursion_depth; bool interactive;
    static RenderConfig default_() { return {8, 500, true, 0.08, 5, 20, false}; }
    RtRenderConfig pod(uint64_t seed = 1) const {
        RtRenderConfig c{};
        c.num_threads = num_threads; c.workload_split = workload_split; c.anti_alias = anti_alias;
        c.aa_threshold = aa_threshold; c.aa_rays = aa_rays; c.recursion_depth = recursion_depth;
        c.interactive = interactive; c.seed = seed; c.rank = 0; c.nranks = 1; c.tile_rows = 16;
        return c;
    }
};
struct ImageDimension {
    uint32_t width, height;
    static ImageDimension new_(uint32_t w, uint32_t h) { return {w, h}; }
    uint32_t area() const { return width * height; }
};
struct CameraConfig {
    DVec3 origin, target, up; double fov_y;
    static CameraConfig default_() { return {dvec3(0, 0, 0), dvec3(0, 0, -100), dvec3(0, 1, 0), 90.0}; }
    RtCamera pod() const {
        RtCamera c{};
        c.origin[0] = origin.x; c.origin[1] = origin.y; c.origin[2] = origin.z;
        c.target[0] = target.x; c.target[1] = target.y; c.target[2] = target.z;
        c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z; c.fov_y = fov_y;
        return c;
    }
};

// RgbImage stand-in: W*H*3 bytes, row-major, top row first.
struct RgbImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
};

// ---------------------------------------------------------------- src/util.rs helpers
ShadableBox texture_phong_material(const std::string& path, double diffuse, double specular, double ambient, double shininess);
inline AmbientLight no_ambient() { return AmbientLight::new_(Color::WHITE, 0.0); }
Scene build_scene(std::vector<LightableBox> lights, AmbientLight ambient, const SkyBox* background, SceneNodeBox root);
inline SceneNodeBox default_node() { return SceneNode::new_(); }
SceneNodeBox geometry_node(DMat4 transform, ShadableBox material, IntersectableBox primitive, std::vector<SceneNodeBox> children);
SceneNodeBox scene_node(DMat4 transform, std::vector<SceneNodeBox> children);
inline ImageDimension image(uint32_t w, uint32_t h) { return {w, h}; }
inline ImageDimension square_image(uint32_t s) { return {s, s}; }
inline CameraConfig camera(DVec3 origin, DVec3 target) { return {origin, target, dvec3(0, 1, 0), 90.0}; }
inline ShadableBox basic_diffuse(Color c) { return PhongShader::new_(c * 1.0, c * 0.0, c * 0.0, 1.0); }
inline ShadableBox slightly_shiney(Color c) { return PhongShader::new_(c * 1.0, c * 0.0, c * 0.0, 1.0); }
SceneNodeBox create_wall_from_material(double size, ShadableBox material, DMat4 transform);
SceneNodeBox create_wall(double size, Color color, DMat4 transform);
SceneNodeBox create_floor(double size, Color color);
SceneNodeBox create_floor_from_material(double size, ShadableBox material);
SceneNodeBox create_mirror(double size, Color color, DMat4 transform);
struct RoomMaterialScheme { ShadableBox ceiling, floor, front, back, left, right; };
struct RoomColorScheme { Color ceiling, floor, front, back, left, right; };
SceneNodeBox create_room_from_material(double size, RoomMaterialScheme m);
SceneNodeBox create_room(double size, RoomColorScheme c);
SceneNodeBox create_interior_box(double size);
SceneNodeBox create_interior_mirror_box(double size);
SceneNodeBox create_cube(double size, DMat4 transform, Color color);
SceneNodeBox create_translucent_cube(double size, DMat4 transform, Color color, double refractive_index);
SceneNodeBox create_translucent_sphere(double size, DMat4 transform, Color color, double refractive_index);
SceneNodeBox create_mirror_sphere(double size, DMat4 transform, Color color);
SceneNodeBox create_comp_sphere(double size, DMat4 transform, Color color);
SceneNodeBox create_sphere(double size, DMat4 transform, Color color);
CompositableBox subtract_shape(DMat4 transform, CompositableBox positive, CompositableBox negative);
CompositableBox or_shape(DMat4 transform, CompositableBox a, CompositableBox b);
CompositableBox xor_shape(DMat4 transform, CompositableBox a, CompositableBox b);
CompositableBox and_shape(DMat4 transform, CompositableBox a, CompositableBox b);
CompositableBox reuleaux_tetrahedron(DMat4 transform, double size);
inline CompositableBox base_shape(DMat4 transform, IntersectableBox prim) { return BaseShape::new_(transform, std::move(prim)); }
inline IntersectableBox sphere(double radius) { return Sphere::from_radius(radius); }
inline IntersectableBox cube(double length) { return Cube::new_(length); }
CompositableBox create_weird(double size, double offset);
SceneNodeBox create_comp_weird(double size, DMat4 transform, Color color);

// Root directory that "assets/..." paths are resolved against (default: $RT_ASSET_ROOT or ".").
void set_asset_root(const std::string& root);
const std::string& asset_root();

}  // namespace raytracer
