// raytracer.cpp — host-side helpers of the reference API mirror: src/util.rs scene
// builders, the OBJ reader behind Mesh::from_path, raw image loading and the
// RtSceneDesc view of a SceneDescBuilder.
#include "raytracer.hpp"

#include <cstdlib>
#include <fstream>
#include <sstream>

namespace raytracer {

static std::string g_asset_root;
void set_asset_root(const std::string& root) { g_asset_root = root; }
const std::string& asset_root() {
    if (g_asset_root.empty()) {
        const char* e = std::getenv("RT_ASSET_ROOT");
        g_asset_root = e ? e : ".";
    }
    return g_asset_root;
}

// "assets/images/textures/x.png" → "<root>/assets/images/textures/x.rtim": images are
// decoded offline (tools/make_assets.py) into headered raw files.
static std::string raw_image_path(const std::string& path) {
    std::string p = path;
    size_t dot = p.find_last_of('.');
    if (dot != std::string::npos) p = p.substr(0, dot);
    return asset_root() + "/" + p + ".rtim";
}

uint32_t SceneDescBuilder::load_image(const std::string& path, uint32_t channels) {
    std::string key = path + "|" + std::to_string(channels);
    auto it = image_by_key.find(key);
    if (it != image_by_key.end()) return it->second;
    std::string file = raw_image_path(path);
    std::ifstream f(file, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open image " + file);
    char magic[4]; uint32_t hdr[3];
    f.read(magic, 4); f.read(reinterpret_cast<char*>(hdr), 12);
    if (!f || std::memcmp(magic, "RTIM", 4) != 0 || hdr[2] != 3) throw std::runtime_error("bad image file " + file);
    uint32_t w = hdr[0], h = hdr[1];
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    f.read(reinterpret_cast<char*>(rgb.data()), (std::streamsize)rgb.size());
    if (!f) throw std::runtime_error("truncated image file " + file);
    RtImage im{}; im.width = w; im.height = h; im.channels = channels;
    while (texels.size() % 16) texels.push_back(0);
    im.offset = texels.size();
    if (channels == 3) {
        texels.insert(texels.end(), rgb.begin(), rgb.end());
    } else {
        // `to_luma()` of the image crate (0.21): l = 0.2126 r + 0.7152 g + 0.0722 b in f32, truncated.
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            float l = 0.2126f * rgb[3 * i] + 0.7152f * rgb[3 * i + 1] + 0.0722f * rgb[3 * i + 2];
            texels.push_back((uint8_t)l);
        }
    }
    uint32_t idx = (uint32_t)images.size();
    images.push_back(im);
    image_by_key[key] = idx;
    return idx;
}

RtSceneDesc SceneDescBuilder::desc() const {
    RtSceneDesc d{};
    d.abi_version = RT_ABI_VERSION;
    d.num_nodes = (uint32_t)nodes.size(); d.nodes = nodes.data();
    d.num_prims = (uint32_t)prims.size(); d.prims = prims.data();
    d.num_points = (uint32_t)(points.size() / 3); d.points = points.data();
    d.num_csg = (uint32_t)csg.size(); d.csg = csg.data();
    d.num_csg_children = (uint32_t)csg_children.size(); d.csg_children = csg_children.data();
    d.num_shaders = (uint32_t)shaders.size(); d.shaders = shaders.data();
    d.num_shader_children = (uint32_t)shader_child_idx.size();
    d.shader_child_idx = shader_child_idx.data(); d.shader_child_weight = shader_child_weight.data();
    d.num_lights = (uint32_t)lights.size(); d.lights = lights.data();
    std::memcpy(d.ambient_color, ambient_color, sizeof ambient_color);
    d.ambient_power = ambient_power;
    d.num_images = (uint32_t)images.size(); d.images = images.data();
    d.texel_bytes = texels.size(); d.texels = texels.data();
    d.num_normal_maps = (uint32_t)normal_maps.size(); d.normal_maps = normal_maps.data();
    d.num_meshes = (uint32_t)meshes.size(); d.meshes = meshes.data();
    d.mesh_f64_count = mesh_f64.size(); d.mesh_f64 = mesh_f64.data();
    d.mesh_u32_count = mesh_u32.size(); d.mesh_u32 = mesh_u32.data();
    d.has_background = has_background ? 1 : 0;
    d.background_image = background_image;
    d.background_trans = background_trans;
    return d;
}

// ---------------------------------------------------------------- OBJ reader (tobj 0.1.7 load_obj semantics)
std::shared_ptr<Mesh> Mesh::from_path(const std::string& path) {
    std::string file = asset_root() + "/" + path;
    std::ifstream f(file);
    if (!f) throw std::runtime_error("cannot open model " + file);
    std::vector<float> v, vt, vn;   // global pools, 3/2/3 floats per entry
    auto mesh = std::make_shared<Mesh>();
    // per-model state: tobj exports one model per `o`/`g` block that has faces and
    // restarts its (v,vt,vn) → index map there; Mesh::from_path then concatenates the
    // models, offsetting face indices by the vertices already appended (mesh.rs:389-395).
    std::map<std::tuple<long, long, long>, uint32_t> index_map;
    uint32_t model_offset = 0, model_vertices = 0;
    bool model_has_faces = false;
    auto flush_model = [&]() {
        model_offset += model_vertices;
        model_vertices = 0;
        index_map.clear();
        model_has_faces = false;
    };
    auto add_vertex = [&](long iv, long it, long in) -> uint32_t {
        auto key = std::make_tuple(iv, it, in);
        auto found = index_map.find(key);
        if (found != index_map.end()) return found->second;
        for (int k = 0; k < 3; ++k) mesh->positions.push_back((double)v[(size_t)iv * 3 + k]);
        if (!vt.empty() && it >= 0) for (int k = 0; k < 2; ++k) mesh->tex_coords.push_back((double)vt[(size_t)it * 2 + k]);
        if (!vn.empty() && in >= 0) for (int k = 0; k < 3; ++k) mesh->vertex_normals.push_back((double)vn[(size_t)in * 3 + k]);
        uint32_t next = model_vertices++;
        index_map[key] = next;
        return next;
    };
    auto resolve = [](long idx, size_t count) -> long {   // OBJ indices: 1-based, negative = relative
        if (idx > 0) return idx - 1;
        if (idx < 0) return (long)count + idx;
        return -1;
    };
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            std::string a, b2, c; ss >> a >> b2 >> c;
            v.push_back(std::strtof(a.c_str(), nullptr)); v.push_back(std::strtof(b2.c_str(), nullptr)); v.push_back(std::strtof(c.c_str(), nullptr));
        } else if (tag == "vt") {
            std::string a, b2; ss >> a >> b2;
            vt.push_back(std::strtof(a.c_str(), nullptr)); vt.push_back(std::strtof(b2.c_str(), nullptr));
        } else if (tag == "vn") {
            std::string a, b2, c; ss >> a >> b2 >> c;
            vn.push_back(std::strtof(a.c_str(), nullptr)); vn.push_back(std::strtof(b2.c_str(), nullptr)); vn.push_back(std::strtof(c.c_str(), nullptr));
        } else if (tag == "f") {
            std::vector<uint32_t> corner;
            std::string tok;
            while (ss >> tok) {
                long iv = 0, it = 0, in = 0;
                size_t s1 = tok.find('/');
                if (s1 == std::string::npos) {
                    iv = std::strtol(tok.c_str(), nullptr, 10);
                } else {
                    iv = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t2 = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t2.empty()) it = std::strtol(t2.c_str(), nullptr, 10);
                    if (s2 != std::string::npos) {
                        std::string t3 = tok.substr(s2 + 1);
                        if (!t3.empty()) in = std::strtol(t3.c_str(), nullptr, 10);
                    }
                }
                corner.push_back(add_vertex(resolve(iv, v.size() / 3), resolve(it, vt.size() / 2), resolve(in, vn.size() / 3)));
            }
            for (size_t i = 1; i + 1 < corner.size(); ++i) {   // fan triangulation
                mesh->faces.push_back(model_offset + corner[0]);
                mesh->faces.push_back(model_offset + corner[i]);
                mesh->faces.push_back(model_offset + corner[i + 1]);
            }
            model_has_faces = true;
        } else if (tag == "o" || tag == "g") {
            if (model_has_faces) flush_model();
        }
    }
    if (mesh->faces.empty()) throw std::runtime_error("model has no faces: " + file);
    return mesh;
}

// ---------------------------------------------------------------- src/util.rs
ShadableBox texture_phong_material(const std::string& path, double diffuse, double specular, double ambient, double shininess) {
    return MixShader::from_shaders({
        TextureShader::new_(ImageTexture::from_path(path)),
        PhongShader::new_(Color::WHITE * diffuse, Color::WHITE * specular, Color::WHITE * ambient, shininess),
    });
}

Scene build_scene(std::vector<LightableBox> lights, AmbientLight ambient, const SkyBox* background, SceneNodeBox root) {
    Scene scene = Scene::new_();
    scene.ambient_light = ambient;
    for (auto& l : lights) scene.add_light(l);
    if (background) scene.set_background(*background);
    scene.root = std::move(root);
    return scene;
}

SceneNodeBox geometry_node(DMat4 transform, ShadableBox material, IntersectableBox primitive, std::vector<SceneNodeBox> children) {
    auto node = SceneNode::new_();
    node->set_transform(transform);
    node->set_material(std::move(material));
    node->set_primitive(std::move(primitive));
    for (auto& c : children) node->add_child(c);
    return node;
}

SceneNodeBox scene_node(DMat4 transform, std::vector<SceneNodeBox> children) {
    auto node = SceneNode::new_();
    node->set_transform(transform);
    for (auto& c : children) node->add_child(c);
    return node;
}

SceneNodeBox create_wall_from_material(double size, ShadableBox material, DMat4 transform) {
    auto wall = SceneNode::new_();
    wall->set_primitive(Rectangle::new_(size, size));
    wall->set_material(std::move(material));
    wall->set_transform(transform);
    return wall;
}

SceneNodeBox create_wall(double size, Color color, DMat4 transform) {
    return create_wall_from_material(size, PhongShader::new_(color * 1.0, color * 0.0, color * 0.0, 1.0), transform);
}

SceneNodeBox create_floor(double size, Color color) {
    auto floor = SceneNode::new_();
    floor->set_primitive(Rectangle::new_(size, size));
    auto comp = CompositeShader::new_();
    comp->add_shader(0.8, PhongShader::new_(color * 1.0, color * 0.0, color * 0.1, 1.0));
    comp->add_shader(0.2, ReflectionShader::new_(color * 1.0));
    floor->set_material(comp);
    floor->set_transform(rotation(Axis::X, -90.0));
    return floor;
}

SceneNodeBox create_floor_from_material(double size, ShadableBox material) {
    return geometry_node(rotation(Axis::X, -90.0), std::move(material), Rectangle::new_(size, size), {});
}

SceneNodeBox create_mirror(double size, Color color, DMat4 transform) {
    return create_wall_from_material(size, ReflectionShader::new_(color * 1.0), transform);
}

SceneNodeBox create_room_from_material(double size, RoomMaterialScheme m) {
    auto interior_box = SceneNode::new_();
    auto floor = create_wall_from_material(size * 1.01, m.floor, rotation(Axis::X, -90.0) * translation(0.0, 0.0, -size / 2.0));
    auto ceiling = create_wall_from_material(size * 1.01, m.ceiling, rotation(Axis::X, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto front = create_wall_from_material(size * 1.01, m.front, translation(0.0, 0.0, -size / 2.0));
    auto back = create_wall_from_material(size * 1.01, m.back, rotation(Axis::X, 180.0) * translation(0.0, 0.0, -size / 2.0));
    auto left = create_wall_from_material(size * 1.01, m.left, rotation(Axis::Y, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto right = create_wall_from_material(size * 1.01, m.right, rotation(Axis::Y, -90.0) * translation(0.0, 0.0, -size / 2.0));
    interior_box->add_child(ceiling);
    interior_box->add_child(floor);
    interior_box->add_child(front);
    interior_box->add_child(back);
    interior_box->add_child(left);
    interior_box->add_child(right);
    return interior_box;
}

SceneNodeBox create_room(double size, RoomColorScheme c) {
    auto mat = [](Color col) { return PhongShader::new_(col * 1.0, col * 0.0, col * 0.0, 1.0); };
    return create_room_from_material(size, RoomMaterialScheme{mat(c.ceiling), mat(c.floor), mat(c.front), mat(c.back), mat(c.left), mat(c.right)});
}

// src/util.rs:210-230: note the "ceiling" wall sits at y = -size/2 and the "floor" at +size/2.
SceneNodeBox create_interior_box(double size) {
    auto box = SceneNode::new_();
    auto ceiling = create_wall(size * 1.01, Color::GREEN, rotation(Axis::X, -90.0) * translation(0.0, 0.0, -size / 2.0));
    auto floor = create_wall(size * 1.01, Color::CHOCOLATE, rotation(Axis::X, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto front = create_wall(size * 1.01, Color::ROSY_BROWN, translation(0.0, 0.0, -size / 2.0));
    auto back = create_wall(size * 1.01, Color::MISTY_ROSE, rotation(Axis::X, 180.0) * translation(0.0, 0.0, -size / 2.0));
    auto left = create_wall(size * 1.01, Color::RED, rotation(Axis::Y, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto right = create_wall(size * 1.01, Color::BLUE, rotation(Axis::Y, -90.0) * translation(0.0, 0.0, -size / 2.0));
    box->add_child(ceiling); box->add_child(floor); box->add_child(front);
    box->add_child(back); box->add_child(left); box->add_child(right);
    return box;
}

SceneNodeBox create_interior_mirror_box(double size) {
    auto box = SceneNode::new_();
    auto ceiling = create_wall(size * 1.01, Color::GREEN, rotation(Axis::X, -90.0) * translation(0.0, 0.0, -size / 2.0));
    auto floor = create_wall(size * 1.01, Color::CHOCOLATE, rotation(Axis::X, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto front = create_mirror(size * 1.01, Color::WHITE, translation(0.0, 0.0, -size / 2.0));
    auto back = create_wall(size * 1.01, Color::MISTY_ROSE, rotation(Axis::X, 180.0) * translation(0.0, 0.0, -size / 2.0));
    auto left = create_mirror(size * 1.01, Color::WHITE, rotation(Axis::Y, 90.0) * translation(0.0, 0.0, -size / 2.0));
    auto right = create_mirror(size * 1.01, Color::WHITE, rotation(Axis::Y, -90.0) * translation(0.0, 0.0, -size / 2.0));
    box->add_child(ceiling); box->add_child(floor); box->add_child(front);
    box->add_child(back); box->add_child(left); box->add_child(right);
    return box;
}

static SceneNodeBox prim_node(IntersectableBox prim, ShadableBox material, DMat4 transform) {
    auto n = SceneNode::new_();
    n->set_primitive(std::move(prim));
    n->set_material(std::move(material));
    n->set_transform(transform);
    return n;
}

SceneNodeBox create_cube(double size, DMat4 transform, Color color) {
    return prim_node(Cube::new_(size), PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0), transform);
}
SceneNodeBox create_translucent_cube(double size, DMat4 transform, Color color, double refractive_index) {
    return prim_node(Cube::new_(size), TranslucentShader::new_(color * 1.0, refractive_index), transform);
}
SceneNodeBox create_translucent_sphere(double size, DMat4 transform, Color color, double refractive_index) {
    return prim_node(OneWay::new_(Sphere::from_radius(size)), TranslucentShader::new_(color * 1.0, refractive_index), transform);
}
SceneNodeBox create_mirror_sphere(double size, DMat4 transform, Color color) {
    return prim_node(OneWay::new_(Sphere::from_radius(size)), ReflectionShader::new_(color * 1.0), transform);
}
static ShadableBox comp_material(Color color) {
    auto comp = CompositeShader::new_();
    comp->add_shader(0.8, PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0));
    comp->add_shader(0.2, ReflectionShader::new_(Color::WHITE));
    return comp;
}
SceneNodeBox create_comp_sphere(double size, DMat4 transform, Color color) {
    return prim_node(OneWay::new_(Sphere::from_radius(size)), comp_material(color), transform);
}
SceneNodeBox create_sphere(double size, DMat4 transform, Color color) {
    return prim_node(Sphere::from_radius(size), PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0), transform);
}

CompositableBox subtract_shape(DMat4 transform, CompositableBox positive, CompositableBox negative) {
    auto s = SubtractShape::new_(std::move(positive), std::move(negative)); s->set_transform(transform); return s;
}
CompositableBox or_shape(DMat4 transform, CompositableBox a, CompositableBox b) {
    auto s = OrShape::new_(std::move(a), std::move(b)); s->set_transform(transform); return s;
}
CompositableBox xor_shape(DMat4 transform, CompositableBox a, CompositableBox b) {
    auto s = XorShape::new_(std::move(a), std::move(b)); s->set_transform(transform); return s;
}
CompositableBox and_shape(DMat4 transform, CompositableBox a, CompositableBox b) {
    auto s = AndShape::new_(std::move(a), std::move(b)); s->set_transform(transform); return s;
}
CompositableBox reuleaux_tetrahedron(DMat4 transform, double size) {
    double radius = std::sqrt(2.0 * size * size);
    auto tetra = MultiAndShape::from_vec({
        base_shape(translation(-size / 2.0, -size / 2.0, size / 2.0), Sphere::from_radius(radius)),
        base_shape(translation(size / 2.0, size / 2.0, size / 2.0), Sphere::from_radius(radius)),
        base_shape(translation(-size / 2.0, size / 2.0, -size / 2.0), Sphere::from_radius(radius)),
        base_shape(translation(size / 2.0, -size / 2.0, -size / 2.0), Sphere::from_radius(radius)),
    });
    tetra->set_transform(transform);
    return tetra;
}
CompositableBox create_weird(double size, double offset) {
    CompositableBox s1 = BaseShape::new_(DMat4::identity(), Sphere::from_radius(size));
    CompositableBox s2 = BaseShape::new_(translation(0.0, offset, 0.0), Sphere::from_radius(size));
    CompositableBox s3 = BaseShape::new_(translation(0.0, 0.0, offset), Sphere::from_radius(size));
    CompositableBox s4 = BaseShape::new_(translation(offset, 0.0, 0.0), Sphere::from_radius(size));
    s1 = SubtractShape::new_(s1, s2);
    s1 = SubtractShape::new_(s1, s3);
    return SubtractShape::new_(s1, s4);
}
SceneNodeBox create_comp_weird(double size, DMat4 transform, Color color) {
    return prim_node(create_weird(size, size), comp_material(color), transform);
}

}  // namespace raytracer
