// scenes.cpp — the reference's test scenes, transcribed against the API mirror in
// raytracer.hpp so they read like tests/*.rs, plus the small C interface the Python
// tests and the benchmark use to obtain an RtSceneDesc for a named scene.
//
// Each scene returns the same (scene, image size, camera, config) the reference
// test passes to render()/render_with_config().  Scenes whose background image is
// one of the reference's missing blobs use sky_ocean.jpg instead; that is noted
// at each site.
#include <functional>
#include <string>
#include <vector>

#include "raytracer.hpp"

using namespace raytracer;

namespace {

const double PI = 3.14159265358979323846;

struct View {
    Scene scene;
    ImageDimension dim;
    CameraConfig cam;
    RenderConfig cfg;
};

// =========================================================== tests/basic_test.rs
Scene setup_scene() {                                                    // :8-15
    Scene test_scene = Scene::new_();
    auto root = SceneNode::new_();
    root->set_primitive(Sphere::from_radius(100.0));
    test_scene.root = root;
    test_scene.add_light(PointLight::new_(dvec3(0.0, 1000.0, 200.0), Color::new_(0.0, 1.0, 1.0), 1.0, {1.0, 0.0, 0.0}));
    return test_scene;
}
CameraConfig setup_camera() { return {dvec3(0.0, 0.0, 200.0), dvec3(0.0, 0.0, 0.0), dvec3(0.0, 1.0, 0.0), 90.0}; }  // :17-22

View basic_setup_scene() {   // setup_scene() as written: the only basic scene whose root keeps its sphere
    return {setup_scene(), square_image(512), setup_camera(), RenderConfig::default_()};
}
View white_sphere_one_light() {                                          // :24-33 (root replaced: renders black)
    Scene s = setup_scene();
    auto root = SceneNode::new_();
    root->set_material(PhongShader::new_(Color::WHITE * 1.0, Color::WHITE * 0.0, Color::WHITE * 0.1, 1.0));
    s.root = root;
    return {s, square_image(512), setup_camera(), RenderConfig::default_()};
}
View plane_test_1() {                                                    // :59-74
    Scene s = setup_scene();
    auto root = SceneNode::new_();
    root->set_material(PhongShader::new_(Color::WHITE * 0.1, Color::WHITE * 0.9, Color::WHITE * 0.1, 4.0));
    root->set_primitive(Rectangle::new_(100.0, 100.0));
    s.root = root;
    s.add_light(PointLight::new_(dvec3(0.0, 0.0, 200.0), Color::new_(1.0, 1.0, 1.0), 1.0, {1.0, 0.0, 0.0}));
    CameraConfig cam{dvec3(50.0, 50.0, 200.0), dvec3(0.0, 0.0, 0.0), dvec3(0.0, 1.0, 0.0), 90.0};
    return {s, square_image(512), cam, RenderConfig::default_()};
}
Scene setup_diffuse_cube(DMat4 transform) {                              // :76-87
    Scene scene = Scene::new_();
    auto root = SceneNode::new_();
    root->set_transform(transform);
    root->set_material(PhongShader::new_(Color::WHITE * 0.9, Color::WHITE * 0.0, Color::WHITE * 0.1, 4.0));
    root->set_primitive(Cube::new_(100.0));
    scene.root = root;
    scene.add_light(PointLight::new_(dvec3(200.0, 300.0, 400.0), Color::new_(1.0, 1.0, 1.0), 2.0, {1.0, 0.0, 0.0}));
    return scene;
}
View cube_view(DMat4 t) {
    return {setup_diffuse_cube(t), square_image(512), camera(dvec3(50.0, 50.0, 200.0), dvec3(0, 0, 0)), RenderConfig::default_()};
}
View cube_test_1() { return cube_view(DMat4::identity()); }              // :89-95
View translate_test_1() { return cube_view(translation(1.0, 0.0, 0.0)); }   // :97-103
View rotate_test_1() { return cube_view(rotation(Axis::Z, 15.0)); }      // :105-111
View scale_test_1() { return cube_view(scaling(2.9, 1.5, 0.3)); }        // :113-119
View child_transform_test_1() {                                          // :121-148
    DMat4 transform = rotation(Axis::Z, 15.0) * translation(50.0, 0.0, 0.0);
    Scene scene = Scene::new_();
    scene.add_light(PointLight::new_(dvec3(200.0, 300.0, 400.0), Color::new_(1.0, 1.0, 1.0), 2.0, {1.0, 0.0, 0.0}));
    auto base_node = SceneNode::new_();
    base_node->set_transform(transform);
    base_node->set_material(PhongShader::new_(Color::WHITE * 0.9, Color::WHITE * 0.0, Color::WHITE * 0.1, 4.0));
    base_node->set_primitive(Cube::new_(50.0));
    SceneNodeBox prev_node;
    SceneNodeBox node = base_node->clone();
    for (int i = 1; i < 5; ++i) {
        prev_node = node;
        node = base_node->clone();
        node->add_child(prev_node->clone());
    }
    scene.root = node;
    return {scene, square_image(512), camera(dvec3(0.0, 0.0, 200.0), dvec3(0, 0, 0)), RenderConfig::default_()};
}
View shadow_test_1() {                                                   // :150-167
    Scene s = setup_scene();
    auto root = SceneNode::new_();
    root->set_material(PhongShader::new_(Color::WHITE * 0.1, Color::WHITE * 0.9, Color::WHITE * 0.1, 4.0));
    root->set_transform(translation(-100.0, -100.0, -100.0));
    auto small_sphere = SceneNode::new_();
    small_sphere->set_primitive(Sphere::from_radius(50.0));
    small_sphere->set_transform(translation(100.0, 100.0, 100.0));
    root->add_child(small_sphere);
    s.root = root;
    s.add_light(PointLight::new_(dvec3(200.0, 200.0, 200.0), Color::new_(1.0, 1.0, 1.0), 1.0, {1.0, 0.0, 0.0}));
    return {s, square_image(512), setup_camera(), RenderConfig::default_()};
}

// =========================================================== tests/mesh_test.rs
namespace mesh_test {
LightableBox light1() { return PointLight::new_(dvec3(-100.0, 300.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light2() { return PointLight::new_(dvec3(100.0, 300.0, -300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light3() { return PointLight::new_(dvec3(-300.0, 300.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light4() { return PointLight::new_(dvec3(0.0, -300.0, 0.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox square_light(DVec3 pos, double size) { return SquareLight::new_(pos, size, Color::WHITE, 1500000.0, {0.0, 0.0, 4.0 * PI}); }
ShadableBox default_material(Color color) { return PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0); }
ShadableBox glass() {                                                    // :37-42
    return CompositeShader::from_shaders({
        {0.1, ReflectionShader::new_(Color::WHITE)},
        {0.9, TranslucentShader::new_(Color::WHITE, 1.517)},
    });
}
ShadableBox frosted_glass() {                                            // :44-52
    return ChainShader::from_shaders({
        NormalMapShader::new_(NormalMap::from_path("assets/images/normal_maps/concrete.jpg")),
        CompositeShader::from_shaders({
            {0.1, ReflectionShader::new_(Color::WHITE)},
            {0.9, TranslucentShader::new_(Color::WHITE, 1.517)},
        }),
    });
}
ShadableBox ceramic_tile() {                                             // :136-148
    return ChainShader::from_shaders({
        NormalMapShader::new_(BumpMap::from_path("assets/images/bump_maps/tiles.jpg", 10.0)),
        CompositeShader::from_shaders({
            {0.8, MixShader::from_shaders({
                      PhongShader::new_(Color::WHITE * 0.5, Color::WHITE * 0.5, Color::BLACK, 2.0),
                      TextureShader::new_(ImageTexture::from_path("assets/images/textures/marble.png")),
                  })},
            {0.2, ReflectionShader::new_(Color::WHITE)},
        }),
    });
}
ShadableBox concrete(Color color) {                                      // :150-158
    return ChainShader::from_shaders({
        NormalMapShader::new_(NormalMap::from_path("assets/images/normal_maps/concrete.jpg")),
        MixShader::from_shaders({
            PhongShader::new_(color * 1.0, Color::WHITE * 0.0, Color::BLACK, 1.0),
            TextureShader::new_(ImageTexture::from_path("assets/images/textures/smooth_concrete.jpg")),
        }),
    });
}
ShadableBox metallic() {                                                 // :160-168
    return ChainShader::from_shaders({
        NormalMapShader::new_(NormalMap::from_path("assets/images/normal_maps/concrete.jpg")),
        CompositeShader::from_shaders({
            {0.3, ReflectionShader::new_(Color::WHITE)},
            {0.7, PhongShader::new_(Color::WHITE * 0.2, Color::BLACK, Color::BLACK, 2.0)},
        }),
    });
}
SceneNodeBox coloured_room() {
    return create_room_from_material(700.0, RoomMaterialScheme{
        default_material(Color::RED), default_material(Color::BLUE), default_material(Color::MAGENTA),
        default_material(Color::CYAN), default_material(Color::YELLOW), default_material(Color::GREEN)});
}
View mesh_basic() {                                                      // :54-93
    Scene scene = build_scene(
        {light1(), light2(), light3(), light4()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            coloured_room(),
            geometry_node(
                translation(0.0, 0.0, 0.0) * scaling(200.0, 200.0, 200.0) * rotation(Axis::Y, -30.0),
                texture_phong_material("assets/images/textures/test3.png", 1.0, 0.0, 0.0, 1.0),
                Mesh::from_path("assets/models/monkey2.obj"), {}),
        }));
    RenderConfig config = RenderConfig::default_();
    config.anti_alias = false;
    return {scene, image(256, 256), camera(dvec3(-310.0, 200.0, 300.0), dvec3(0.0, 0.0, 0.0)), config};
}
View glass_monkey() {                                                    // :95-134 — the north-star scene
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            coloured_room(),
            geometry_node(
                translation(0.0, 0.0, 0.0) * scaling(200.0, 200.0, 200.0) * rotation(Axis::Y, -30.0),
                glass(),
                Mesh::from_path("assets/models/monkey2.obj"), {}),
        }));
    RenderConfig config = RenderConfig::default_();
    config.anti_alias = true;
    return {scene, image(1000, 1000), camera(dvec3(-310.0, 200.0, 300.0), dvec3(0.0, 0.0, 0.0)), config};
}
View monkey_n(int n) {                                                   // :170-277
    RoomMaterialScheme room = (n == 1)
        ? RoomMaterialScheme{default_material(Color::WHITE), ceramic_tile(), default_material(Color::MAGENTA),
                             default_material(Color::CYAN), default_material(Color::YELLOW), default_material(Color::GREEN)}
        : RoomMaterialScheme{default_material(Color::WHITE), ceramic_tile(), concrete(Color::FIREBRICK),
                             default_material(Color::CYAN), default_material(Color::MAGENTA), metallic()};
    ShadableBox monkey_mat = n == 1 ? concrete(Color::WHITE) : (n == 2 ? metallic() : frosted_glass());
    Scene scene = build_scene(
        {square_light(dvec3(0.0, 340.0, 0.0), n == 1 ? 500.0 : 200.0)}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, room),
            geometry_node(
                translation(0.0, 0.0, 0.0) * scaling(150.0, 150.0, 150.0) * rotation(Axis::Y, -30.0),
                monkey_mat, Mesh::from_path("assets/models/monkey2.obj"), {}),
        }));
    ImageDimension dim = n == 1 ? image(1000, 1000) : image(1920, 1920);
    return {scene, dim, camera(dvec3(-310.0, 200.0, 300.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
}  // namespace mesh_test

// =========================================================== tests/square_light_test.rs
namespace square_light_test {
LightableBox square_light(DVec3 pos, double size) { return SquareLight::new_(pos, size, Color::WHITE, 1500000.0, {0.0, 0.0, 4.0 * PI}); }
LightableBox light2() { return PointLight::new_(dvec3(100.0, 300.0, -300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 4.0 * PI}); }
View square_light_basic() {                                              // :29-61
    Scene scene = build_scene(
        {light2()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 0.0),
                          PhongShader::new_(Color::RED * 0.5, Color::RED * 0.5, Color::BLACK, 4.0),
                          Sphere::from_radius(50.0), {}),
        }));
    RenderConfig config = RenderConfig::default_();
    config.anti_alias = true;
    return {scene, image(256, 256), camera(dvec3(0.0, 0.0, 100.0), dvec3(0.0, 0.0, 0.0)), config};
}
View light_room() {                                                      // :63-100
    Scene scene = build_scene(
        {square_light(dvec3(0.0, 340.0, 0.0), 500.0)}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                PhongShader::new_(Color::WHITE, Color::BLACK, Color::BLACK, 1.0),
                PhongShader::new_(Color::WHITE, Color::BLACK, Color::BLACK, 1.0),
                PhongShader::new_(Color::BLUE, Color::BLACK, Color::BLACK, 1.0),
                PhongShader::new_(Color::CYAN, Color::BLACK, Color::BLACK, 1.0),
                PhongShader::new_(Color::PURPLE, Color::BLACK, Color::BLACK, 1.0),
                ReflectionShader::new_(Color::WHITE)}),
            geometry_node(translation(-150.0, -270.0, 0.0),
                          PhongShader::new_(Color::WHITE * 0.3, Color::WHITE * 0.7, Color::BLACK, 4.0),
                          Sphere::from_radius(80.0), {}),
            geometry_node(translation(150.0, -270.0, 80.0),
                          texture_phong_material("assets/images/textures/cube_rgb_gradient.png", 0.5, 0.5, 0.0, 4.0),
                          Cube::new_(160.0), {}),
        }));
    RenderConfig config = RenderConfig::default_();
    config.anti_alias = true;
    return {scene, image(512, 512), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), config};
}
View subdivision_light() {                                               // :102-141
    auto white = [] { return PhongShader::new_(Color::WHITE, Color::BLACK, Color::BLACK, 1.0); };
    Scene scene = build_scene(
        {square_light(dvec3(0.0, 340.0, 0.0), 600.0)}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{white(), white(), white(), white(), white(), white()}),
        }));
    RenderConfig config = RenderConfig::default_();
    config.anti_alias = false;
    return {scene, image(256, 256), camera(dvec3(-300.0, -300.0, 300.0), dvec3(300.0, 300.0, -300.0)), config};
}
}  // namespace square_light_test

// =========================================================== tests/scene_test.rs
namespace scene_test {
const double room_size = 200.0;
LightableBox room_light() {
    return PointLight::new_(dvec3(0.0, (room_size / 2.0) * 0.6, (room_size / 2.0) * 0.6), Color::new_(1, 1, 1), 100000.0, {0.0, 0.0, 4.0 * PI});
}
View scene_test_n(int n) {                                               // :9-82
    Scene s = Scene::new_();
    s.root = n == 2 ? create_interior_mirror_box(room_size) : create_interior_box(room_size);
    s.add_light(room_light());
    s.ambient_light = AmbientLight::new_(Color::WHITE, 1.0);
    SceneNodeBox obj;
    if (n == 1 || n == 2) obj = create_sphere(25.0, translation(0.0, -75.0, 0.0), Color::NAVY);
    else if (n == 3) obj = create_sphere(50.0, translation(0.0, -50.0, 0.0), Color::NAVY);
    else if (n == 4) obj = create_translucent_sphere(50.0, translation(0.0, -50.0, 0.0), Color::WHITE, 1.52);
    else obj = create_cube(50.0, translation(0.0, -75.0, 0.0), Color::WHITE);
    s.root->add_child(obj);
    CameraConfig cam = n == 3 ? camera(dvec3(0.0, -50.0, 0.0), dvec3(0.0, 0.0, -(room_size / 2.0)))
                              : camera(dvec3(0.0, 0.0, room_size / 2.0), dvec3(0.0, -(room_size / 2.0) * 0.6, 0.0));
    ImageDimension dim = n == 5 ? image(512, 512) : image(1280, 720);
    return {s, dim, cam, RenderConfig::default_()};
}
View refraction_test(int i) {                                            // :84-99
    Scene s = Scene::new_();
    s.root = create_interior_box(room_size);
    s.add_light(room_light());
    s.ambient_light = AmbientLight::new_(Color::WHITE, 1.0);
    s.root->add_child(create_translucent_sphere(25.0, translation(0.0, -50.0, 50.0), Color::WHITE, 1.0 + i * 0.3));
    return {s, image(512, 512), camera(dvec3(0.0, -50.0, room_size / 2.0), dvec3(0.0, -50.0, 0.0)), RenderConfig::default_()};
}
View background_1() {                                                    // :117-131 (room.jpg is a missing blob → room_lores.jpg)
    Scene scene = Scene::new_();
    scene.root->add_child(create_translucent_sphere(50.0, translation(-60.0, 0.0, 0.0), Color::WHITE, 1.52));
    scene.root->add_child(create_mirror_sphere(50.0, translation(60.0, 0.0, 0.0), Color::WHITE));
    scene.add_light(PointLight::new_(dvec3(0.0, 60, 60), Color::new_(1, 1, 1), 100000.0, {0.0, 0.0, 4.0 * PI}));
    scene.ambient_light = AmbientLight::new_(Color::WHITE, 1.0);
    scene.set_background_from_path("assets/images/backgrounds/room_lores.jpg");
    return {scene, image(512, 512), camera(dvec3(0.0, 0.0, 200.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View many_balls() {                                                      // :133-153 (forest2.jpg is a missing blob → sky_ocean.jpg)
    Scene scene = Scene::new_();
    scene.root->add_child(create_comp_sphere(80.0, translation(-150.0, 80.0, 40.0), Color::RED));
    scene.root->add_child(create_comp_sphere(60.0, translation(80.0, 60.0, 0.0), Color::BLUE));
    scene.root->add_child(create_comp_sphere(40.0, translation(50.0, 40.0, 150.0), Color::GREEN));
    scene.root->add_child(create_comp_sphere(20.0, translation(-50.0, 20.0, 150.0), Color::PURPLE));
    scene.root->add_child(create_floor(600.0, Color::GRAY));
    scene.add_light(PointLight::new_(dvec3(-100.0, 300.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}));
    scene.ambient_light = AmbientLight::new_(Color::WHITE, 0.0);
    scene.set_background_from_path("assets/images/backgrounds/sky_ocean.jpg");
    return {scene, image(1920, 1080), camera(dvec3(0.0, 200.0, 400.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}

// tests/bump_map_test.rs:15-27
ShadableBox test_bump_shader(Color color, std::shared_ptr<NormalMappable> normal_map) {
    auto bump_shader = ChainShader::new_();
    bump_shader->push_shader(NormalMapShader::new_(std::move(normal_map)));
    bump_shader->push_shader(PhongShader::new_(color * 1.0, Color::BLACK, Color::BLACK, 1.0));
    return bump_shader;
}

// The 4K "forest" workload: many_balls plus 1024 seeded spheres on a jittered 32x32
// grid over the 600x600 floor, radii U[10,40), materials cycled over the reference's
// sphere materials (create_comp_sphere, texture_phong_material(marble), test_bump_shader
// (bumps), concrete).  SplitMix64, seed 42.
struct SplitMix64 {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
View many_balls_forest() {
    View v = many_balls();
    SplitMix64 rng{42};
    const Color palette[6] = {Color::RED, Color::BLUE, Color::GREEN, Color::PURPLE, Color::ORANGE, Color::TEAL};
    ShadableBox marble = texture_phong_material("assets/images/textures/marble.png", 0.7, 0.3, 0.0, 4.0);
    ShadableBox bumpy = test_bump_shader(Color::WHITE, BumpMap::from_path("assets/images/bump_maps/bumps.png", 1.0));
    ShadableBox conc = mesh_test::concrete(Color::WHITE);
    const double cell = 600.0 / 32.0;
    for (int gz = 0; gz < 32; ++gz)
        for (int gx = 0; gx < 32; ++gx) {
            int i = gz * 32 + gx;
            double radius = 10.0 + 30.0 * rng.uniform();
            double x = -300.0 + (gx + rng.uniform()) * cell;
            double z = -300.0 + (gz + rng.uniform()) * cell;
            DMat4 t = translation(x, radius, z);
            SceneNodeBox node;
            switch (i % 4) {
                case 0: node = create_comp_sphere(radius, t, palette[(i / 4) % 6]); break;
                case 1: node = geometry_node(t, marble, OneWay::new_(Sphere::from_radius(radius)), {}); break;
                case 2: node = geometry_node(t, bumpy, OneWay::new_(Sphere::from_radius(radius)), {}); break;
                default: node = geometry_node(t, conc, OneWay::new_(Sphere::from_radius(radius)), {}); break;
            }
            v.scene.root->add_child(node);
        }
    v.dim = image(3840, 2160);
    return v;
}
}  // namespace scene_test

// =========================================================== tests/composite_shape_test.rs
namespace composite_shape_test {
LightableBox light1() { return PointLight::new_(dvec3(-100.0, 300.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light2() { return PointLight::new_(dvec3(100.0, 300.0, -300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
ShadableBox default_material(Color color) { return PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0); }
ShadableBox clear_material() { return TranslucentShader::new_(Color::WHITE, 1.52); }
CompositableBox make_spikey_thing(double size) {                         // :72-98
    return subtract_shape(DMat4::identity(),
        subtract_shape(DMat4::identity(),
            subtract_shape(DMat4::identity(),
                base_shape(DMat4::identity(), sphere(size)),
                base_shape(translation(size, 0.0, 0.0), sphere(size))),
            base_shape(translation(0.0, size, 0.0), sphere(size))),
        base_shape(translation(0.0, 0.0, size), sphere(size)));
}
CompositableBox make_weird_cube(double size) {                           // :100-112
    return subtract_shape(DMat4::identity(), base_shape(DMat4::identity(), cube(size * 2.0)), base_shape(DMat4::identity(), sphere(size)));
}
CompositableBox subracted_sphere(double size) {                          // :114-126
    return subtract_shape(DMat4::identity(), base_shape(DMat4::identity(), sphere(size)), base_shape(DMat4::identity(), cube(size * 0.7)));
}
CompositableBox make_round_cube(double size) {                           // :243-255
    return and_shape(DMat4::identity(), base_shape(DMat4::identity(), cube(size * 1.5)), base_shape(DMat4::identity(), sphere(size)));
}
CompositableBox make_xor_sphere(double size) {                           // :293-305
    return xor_shape(DMat4::identity(), base_shape(translation(size / 4.0, 0.0, 0.0), sphere(size)),
                     base_shape(translation(-size / 4.0, 0.0, 0.0), sphere(size)));
}
ShadableBox shiney_material(Color color) {                               // :257-264
    auto comp = CompositeShader::new_();
    comp->add_shader(0.8, PhongShader::new_(color * 0.5, color * 0.5, color * 0.01, 4.0));
    comp->add_shader(0.2, ReflectionShader::new_(Color::WHITE));
    return comp;
}
const char* SKY = "assets/images/backgrounds/sky_ocean.jpg";   // building.jpg / forest2.jpg are missing blobs
CameraConfig outside_cam() { return camera(dvec3(0.0, 0.0, 350.0), dvec3(0.0, -100.0, -350.0)); }

View subtract_sphere() {                                                 // :9-29
    Scene scene = Scene::new_();
    auto root = SceneNode::new_();
    CompositableBox s1 = BaseShape::new_(DMat4::identity(), Sphere::from_radius(50.0));
    CompositableBox s2 = BaseShape::new_(translation(0.0, 50.0, 20.0), Sphere::from_radius(40.0));
    root->set_primitive(SubtractShape::new_(s1, s2));
    root->set_material(slightly_shiney(Color::LIME));
    scene.root = root;
    scene.add_light(PointLight::new_(dvec3(100.0, 300.0, 200.0), Color::WHITE, 100000.0, {0.0, 0.0, 4.0 * PI}));
    scene.add_light(PointLight::new_(dvec3(-100.0, 300.0, 200.0), Color::WHITE, 100000.0, {0.0, 0.0, 4.0 * PI}));
    return {scene, image(1280, 720), camera(dvec3(0.0, 150.0, 100.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View many_weirds() {                                                     // :39-62
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 180.0));
    Scene scene = build_scene({light1()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            create_comp_weird(80.0, translation(-150.0, 80.0, 40.0), Color::RED),
            create_comp_weird(60.0, translation(80.0, 60.0, 0.0), Color::BLUE),
            create_comp_weird(40.0, translation(50.0, 40.0, 150.0), Color::GREEN),
            create_comp_weird(20.0, translation(-50.0, 20.0, 150.0), Color::PURPLE),
            create_floor(600.0, Color::GRAY),
        }));
    return {scene, image(1920, 1080), camera(dvec3(0.0, 300.0, 300.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View subtraction_room() {                                                // :139-177
    double sphere_size = 80.0;
    Scene scene = build_scene({light1(), light2()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room(700.0, RoomColorScheme{Color::RED, Color::GREEN, Color::BLUE, Color::CYAN, Color::MAGENTA, Color::YELLOW}),
            geometry_node(translation(-150.0, 0.0, 40.0) * rotation(Axis::Z, 60.0), clear_material(), make_spikey_thing(sphere_size), {}),
            geometry_node(translation(100.0, 0.0, 40.0) * rotation(Axis::Y, 45.0), clear_material(), make_weird_cube(sphere_size), {}),
        }));
    return {scene, image(1920, 1080), outside_cam(), RenderConfig::default_()};
}
View subtraction_outside1() {                                            // :179-212
    double sphere_size = 80.0;
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 0.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(-150.0, 0.0, 40.0) * rotation(Axis::Z, 60.0), clear_material(), make_spikey_thing(sphere_size), {}),
            geometry_node(translation(100.0, 0.0, 40.0) * rotation(Axis::Y, 45.0), clear_material(), make_weird_cube(sphere_size), {}),
        }));
    return {scene, image(5000, 5000), outside_cam(), RenderConfig::default_()};
}
View subtraction_outside3() {                                            // :214-241
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 11.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 40.0) * rotation(Axis::Y, 45.0), clear_material(), subracted_sphere(100.0), {}),
        }));
    return {scene, image(5000, 5000), outside_cam(), RenderConfig::default_()};
}
View rounded_cube() {                                                    // :266-291
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 0.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 40.0) * rotation(Axis::Y, 45.0) * rotation(Axis::Z, 15.0),
                          shiney_material(Color::MAROON), make_round_cube(100.0), {}),
        }));
    return {scene, image(1920, 1080), outside_cam(), RenderConfig::default_()};
}
View xor_test() {                                                        // :307-334
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 77.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 40.0), clear_material(), make_xor_sphere(100.0), {}),
        }));
    return {scene, image(1920, 1080), outside_cam(), RenderConfig::default_()};
}
View subtraction_outside4() {                                            // :336-367
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 130.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(10.0, 0.0, 40.0) * rotation(Axis::Y, 45.0), clear_material(),
                          subtract_shape(DMat4::identity(), make_round_cube(100.0), make_xor_sphere(50.0)), {}),
        }));
    return {scene, image(5000, 5000), outside_cam(), RenderConfig::default_()};
}
View reuleaux() {                                                        // :369-400
    SkyBox sky = SkyBox::from_path(SKY, rotation(Axis::Y, 200.0));
    Scene scene = build_scene({light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(10.0, 0.0, 40.0), default_material(Color::RED),
                          reuleaux_tetrahedron(DMat4::identity(), 100.0), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(-200.0, 350.0, 50.0), dvec3(10.0, 0.0, 40.0)), RenderConfig::default_()};
}
}  // namespace composite_shape_test

// =========================================================== tests/bump_map_test.rs, polyhedra_test.rs
View bump_map_basic_1() {                                                // bump_map_test.rs:29-53
    auto dot = [] { return scene_test::test_bump_shader(Color::WHITE, BumpMap::from_path("assets/images/bump_maps/dot.png", 1.0)); };
    Scene scene = build_scene(
        {PointLight::new_(dvec3(100.0, 200.0, 200.0), Color::new_(1, 1, 1), 300000.0, {0.0, 0.0, 1.0 * PI})}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                dot(), dot(), dot(), dot(), dot(),
                scene_test::test_bump_shader(Color::WHITE, BumpMap::from_path("assets/images/bump_maps/bumps.png", 1.0))}),
        }));
    return {scene, image(500, 500), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}
View octahedron_basic_n(int n) {                                         // polyhedra_test.rs:70-101: two cameras
    using namespace mesh_test;
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::WHITE), default_material(Color::WHITE), default_material(Color::RED),
                default_material(Color::CYAN), default_material(Color::GREEN), ReflectionShader::new_(Color::WHITE)}),
            geometry_node(translation(0.0, 0.0, 0.0),
                          texture_phong_material("assets/images/textures/d8_num.png", 1.0, 0.0, 0.0, 2.0),
                          Polyhedron::octahedron(200.0), {}),
        }));
    return {scene, image(512, 512), n == 1 ? camera(dvec3(-300.0, 0.0, 300.0), dvec3(0.0, 0.0, 0.0)) : camera(dvec3(300.0, 0.0, -300.0), dvec3(0.0, 0.0, 0.0)),
            RenderConfig::default_()};
}
View polyhedron_basic() {                                                // polyhedra_test.rs:26-68 (granite.jpg → marble.png)
    using namespace mesh_test;
    auto prism = Polyhedron::from_planes({
        *Plane::new_(dvec3(0.0, -1.0, 0.0), dvec3(0.0, -1.0, 0.0)),
        *Plane::new_(dvec3(1.0, 0.0, 0.0), dvec3(1.0, 0.0, 0.0)),
        *Plane::new_(dvec3(-1.0, 0.0, 0.0), dvec3(-1.0, 0.0, 0.0)),
        *Plane::new_(dvec3(0.0, -1.0, -1.0), dvec3(0.0, 1.0, -1.0)),
        *Plane::new_(dvec3(0.0, -1.0, 1.0), dvec3(0.0, 1.0, 1.0)),
    });
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::WHITE), default_material(Color::WHITE), default_material(Color::RED),
                default_material(Color::CYAN), default_material(Color::GREEN), ReflectionShader::new_(Color::WHITE)}),
            geometry_node(scaling(50.0, 200.0, 100.0) * rotation(Axis::Y, -45.0),
                          texture_phong_material("assets/images/textures/marble.png", 0.5, 0.5, 0.0, 2.0), prism, {}),
        }));
    return {scene, image(512, 512), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}

// =========================================================== tests/basic_test.rs (the two Phong tests)
View phong_test_n(int n) {                                               // :36-58 (root replaced by a primitive-less node: renders black)
    Scene test_scene = setup_scene();
    test_scene.add_light(PointLight::new_(dvec3(200.0, 200.0, 200.0), Color::new_(1.0, 1.0, 1.0), 1.0, {1.0, 0.0, 0.0}));
    auto root = SceneNode::new_();
    root->set_material(PhongShader::new_(Color::WHITE * 0.1, Color::WHITE * 0.9, Color::WHITE * 0.1, n == 1 ? 2.0 : 4.0));
    test_scene.root = root;
    return {test_scene, square_image(512), setup_camera(), RenderConfig::default_()};
}

// =========================================================== tests/scene_test.rs (the rest)
namespace scene_test {
View scene_light_1(int i) {                                              // :101-115, i = 1..10
    Scene s = Scene::new_();
    s.root = create_interior_box(room_size);
    const double a = room_size / 3.0, h = (room_size / 2.0) * 0.6;
    for (DVec3 p : {dvec3(a, h, a), dvec3(-a, h, a), dvec3(a, h, -a), dvec3(-a, h, -a)})
        s.add_light(PointLight::new_(p, Color::new_(1, 1, 1), 2000.0 * i, {0.0, 0.0, 4.0 * PI}));
    return {s, image(512, 512), camera(dvec3(0.0, 0.0, 200.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View translucent_shadow() {                                              // :155-171, i = 9 (forest2.jpg is a missing blob → sky_ocean.jpg)
    Scene scene = Scene::new_();
    scene.root->add_child(create_translucent_cube(160.0, translation(0.0, 100.0, 0.0) * scaling(1.0, 1.0, 1.0 / 160.0 * 9.0),
                                                  Color::WHITE - Color::LIGHT_BLACK, 1.52));
    scene.root->add_child(create_floor(600.0, Color::WHITE));
    scene.add_light(PointLight::new_(dvec3(-100.0, 300.0, -300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}));
    scene.set_background_from_path("assets/images/backgrounds/sky_ocean.jpg");
    return {scene, image(1920, 1080), camera(dvec3(0.0, 200.0, 400.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
}  // namespace scene_test

// =========================================================== tests/bump_map_test.rs:55-79
View bump_map_basic_2() {
    auto dot = [] { return scene_test::test_bump_shader(Color::WHITE, BumpMap::from_path("assets/images/bump_maps/dot.png", 1.0)); };
    Scene scene = build_scene(
        {PointLight::new_(dvec3(100.0, 200.0, -200.0), Color::new_(1, 1, 1), 300000.0, {0.0, 0.0, 1.0 * PI})}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                dot(), dot(), dot(), dot(), dot(),
                scene_test::test_bump_shader(Color::WHITE, BumpMap::from_path("assets/images/bump_maps/bumps.png", 1.0))}),
        }));
    return {scene, image(500, 500), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}

// =========================================================== tests/transparency_test.rs
namespace transparency_test {
LightableBox light1() { return PointLight::new_(dvec3(0.0, 300.0, 0.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light2() { return PointLight::new_(dvec3(-300.0, 300.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
LightableBox light3() { return PointLight::new_(dvec3(-300.0, -340.0, 300.0), Color::new_(1, 1, 1), 150000.0, {0.0, 0.0, 1.0 * PI}); }
View transparent_shapes() {                                              // :26-60
    using mesh_test::default_material;
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::GOLD), default_material(Color::BROWN), default_material(Color::FIREBRICK),
                default_material(Color::CYAN), default_material(Color::ORANGE), default_material(Color::LAVENDER)}),
            geometry_node(translation(-150.0, 0.0, 0.0), TranslucentShader::new_(Color::WHITE, 1.52), Sphere::from_radius(160.0), {}),
            geometry_node(translation(150.0, -220.0, 80.0), TranslucentShader::new_(Color::WHITE, 2.42), Cube::new_(160.0), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}
View transparent_orb() {                                                 // :62-85
    SkyBox sky = SkyBox::from_path("assets/images/backgrounds/sky_ocean.jpg", rotation(Axis::Y, 20.0));
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 0.0), TranslucentShader::new_(Color::WHITE, 1.52), Sphere::from_radius(160.0), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(0.0, 0.0, 300.0), dvec3(0.0, 0.0, -0.0)), RenderConfig::default_()};
}
}  // namespace transparency_test

// =========================================================== tests/normal_map_test.rs
// brick_shader (src/util.rs:13-29); normal_maps/brick_wall.jpg is a missing blob → brick_wall_01.png
ShadableBox brick_shader() {
    auto brick = ChainShader::new_();
    brick->push_shader(NormalMapShader::new_(NormalMap::from_path("assets/images/normal_maps/brick_wall_01.png")));
    brick->push_shader(PhongShader::new_(Color::WHITE * 0.9, Color::WHITE * 0.1, Color::WHITE * 0.0, 2.0));
    return brick;
}
View normal_map_room_n(int n) {                                          // :26-76
    using mesh_test::default_material;
    LightableBox light = n == 1 ? PointLight::new_(dvec3(200.0, 200.0, -200.0), Color::new_(1, 1, 1), 300000.0, {0.0, 0.0, 1.0 * PI})
                                : PointLight::new_(dvec3(200.0, 200.0, 200.0), Color::new_(1, 1, 1), 300000.0, {0.0, 0.0, 1.0 * PI});
    ShadableBox brick = brick_shader();
    Scene scene = build_scene(
        {light}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::WHITE), default_material(Color::WHITE), brick,
                default_material(Color::CYAN), default_material(Color::WHITE), brick}),
        }));
    return {scene, image(1920, 1080), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}

// =========================================================== tests/texture_test.rs
// Large or missing textures are replaced by images that are already committed: granite.jpg (12 MB raw)
// → marble.png, wood_boards.jpg (35 MB) → light_wood.jpg, brick_wall.jpg (missing blob) → smooth_concrete.jpg,
// orange_leather.jpg (12 MB) → smooth_concrete.jpg.
namespace texture_test {
using namespace mesh_test;
View texture_room() {                                                    // :34-69
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                texture_phong_material("assets/images/textures/marble.png", 1.0, 0.0, 0.0, 2.0),
                texture_phong_material("assets/images/textures/light_wood.jpg", 1.0, 0.0, 0.0, 1.0),
                texture_phong_material("assets/images/textures/smooth_concrete.jpg", 1.0, 0.0, 0.0, 1.0),
                default_material(Color::CYAN),
                texture_phong_material("assets/images/textures/smooth_concrete.jpg", 1.0, 0.0, 0.0, 1.0),
                ReflectionShader::new_(Color::WHITE)}),
            geometry_node(translation(-150.0, -270.0, 0.0),
                          texture_phong_material("assets/images/textures/smooth_concrete.jpg", 1.0, 0.0, 0.0, 1.0), Sphere::from_radius(80.0), {}),
            geometry_node(translation(150.0, -270.0, 80.0),
                          texture_phong_material("assets/images/textures/cube_rgb_gradient.png", 0.5, 0.5, 0.0, 4.0), Cube::new_(160.0), {}),
        }));
    return {scene, image(5000, 5000), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}
View texture_cube() {                                                    // :71-98
    Scene scene = build_scene(
        {light1(), light2(), light3(), light4()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::WHITE), ReflectionShader::new_(Color::WHITE), ReflectionShader::new_(Color::WHITE),
                default_material(Color::CYAN), default_material(Color::YELLOW), ReflectionShader::new_(Color::WHITE)}),
            geometry_node(translation(0.0, 0.0, 0.0),
                          texture_phong_material("assets/images/textures/cube_rgb_numbers.png", 0.5, 0.5, 0.0, 4.0), Cube::new_(160.0), {}),
        }));
    return {scene, image(1920, 1080), camera(dvec3(-310.0, 300.0, 300.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View skybox_test() {                                                     // :100-114: nothing but the sky box
    SkyBox sky = SkyBox::from_path("assets/images/living_room.jpg", DMat4::identity());
    Scene scene = build_scene({}, no_ambient(), &sky, scene_node(DMat4::identity(), {}));
    return {scene, image(512, 512), camera(dvec3(-310.0, 300.0, 300.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
View correlated_texture() {                                              // :116-138
    auto tiny = [](double d, double s, double sh) { return texture_phong_material("assets/images/tiny.png", d, s, 0.0, sh); };
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                tiny(4.0, 6.0, 2.0), tiny(1.0, 0.0, 1.0), tiny(1.0, 0.0, 1.0), default_material(Color::CYAN), tiny(1.0, 0.0, 1.0), tiny(1.0, 0.0, 1.0)}),
        }));
    return {scene, image(512, 512), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}
View plane_texture() {                                                   // :140-160
    Scene scene = build_scene(
        {PointLight::new_(dvec3(0.0, 0.0, 1000.0), Color::new_(1, 1, 1), 1.0, {1.0, 0.0, 0.0})}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            geometry_node(translation(0.0, 0.0, 0.0), texture_phong_material("assets/images/textures/keanu.jpg", 1.0, 0.0, 0.0, 4.0),
                          Plane::with_tangent(dvec3(0.0, 0.0, 0.0), dvec3(0.0, 0.0, 1.0), dvec3(0.0, 1.0, 0.0)), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(0.0, 0.0, 1.0), dvec3(1.0, 0.0, 1.0)), RenderConfig::default_()};
}
}  // namespace texture_test

// =========================================================== tests/dice_scene.rs
// nice_room.jpg (23 MB raw) → sky_ocean.jpg, gurf-1.jpg (11 MB) → keanu.jpg, granite.jpg → marble.png
namespace dice_scene {
LightableBox light1() { return PointLight::new_(dvec3(0.0, 100.0, 0.0), Color::new_(1, 1, 1), 90000.0, {0.0, 0.0, 4.0 * PI}); }
LightableBox light2() { return PointLight::new_(dvec3(-50.0, 100.0, 100.0), Color::new_(1, 1, 1), 75000.0, {0.0, 0.0, 4.0 * PI}); }
LightableBox light3() { return PointLight::new_(dvec3(50.0, 100.0, -100.0), Color::new_(1, 1, 1), 75000.0, {0.0, 0.0, 4.0 * PI}); }
LightableBox front_light() { return PointLight::new_(dvec3(0.0, 0.0, 100.0), Color::new_(1, 1, 1), 1.0, {1.0, 0.0, 0.0}); }
ShadableBox wood_material() {                                            // :26-39
    return CompositeShader::from_shaders({
        {0.8, MixShader::from_shaders({
                  PhongShader::new_(Color::WHITE * 0.5, Color::WHITE * 0.5, Color::BLACK, 2.0),
                  TextureShader::new_(ImageTexture::from_path("assets/images/textures/light_wood.jpg")),
              })},
        {0.2, ReflectionShader::new_(Color::WHITE)},
    });
}
SceneNodeBox make_character_sheet(double size, DMat4 transform) {       // :41-48
    return geometry_node(transform * translation(0.0, 0.01, 0.0) * rotation(Axis::X, -90.0),
                         texture_phong_material("assets/images/textures/keanu.jpg", 1.0, 0.0, 0.0, 1.0), Rectangle::new_(1.7 * size, 2.2 * size), {});
}
SceneNodeBox make_d6(double size, DMat4 transform) {                     // :50-63
    return geometry_node(transform * translation(0.0, size / 2.0, 0.0),
                         ChainShader::from_shaders({
                             NormalMapShader::new_(BumpMap::from_path("assets/images/bump_maps/d6_num.png", 10.0)),
                             CompositeShader::from_shaders({
                                 {0.05, ReflectionShader::new_(Color::WHITE)},
                                 {0.95, texture_phong_material("assets/images/textures/d6_num.png", 0.9, 0.1, 0.0, 2.0)},
                             }),
                         }),
                         Cube::new_(size), {});
}
SceneNodeBox make_d4(double size, DMat4 transform) {                     // :65-72
    return geometry_node(transform, texture_phong_material("assets/images/textures/d4_num.png", 1.0, 0.0, 0.0, 1.0), Tetrahedron::new_(size), {});
}
View dice_scene_view(bool hi_res) {                                      // :74-107
    SkyBox sky = SkyBox::from_path("assets/images/backgrounds/sky_ocean.jpg", rotation(Axis::Y, 20.0));
    Scene scene = build_scene(
        {light1(), light2()}, no_ambient(), &sky,
        scene_node(DMat4::identity(), {
            create_floor_from_material(200.0, wood_material()),
            make_character_sheet(30.0, translation(-20.0, 0.0, 50.0) * rotation(Axis::Y, 230.0)),
            make_d6(5.0, translation(-30.0, 0.0, 50.0)),
            make_d4(5.0, translation(-20.0, 0.0, 50.0)),
        }));
    RenderConfig config = RenderConfig::default_();
    if (hi_res) config.anti_alias = false;
    return {scene, hi_res ? image(1920, 1080) : image(192, 108), camera(dvec3(-30.0, 20.0, 70.0), dvec3(50.0, -50.0, -80.0)), config};
}
View polygon_basic() {                                                   // :109-133
    Scene scene = build_scene(
        {front_light()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            geometry_node(rotation(Axis::Y, 0.0), texture_phong_material("assets/images/textures/marble.png", 1.0, 0.0, 0.0, 1.0),
                          Polygon::from_vertices({dvec3(-10.0, -10.0, 0.0), dvec3(0.0, -10.0, 0.0), dvec3(10.0, 0.0, 0.0), dvec3(10.0, 10.0, 0.0)}), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(0.0, 0.0, 10.0), dvec3(0.0, 0.0, -10.0)), RenderConfig::default_()};
}
View triangle_basic() {                                                  // :135-155
    Scene scene = build_scene(
        {front_light()}, no_ambient(), nullptr,
        scene_node(DMat4::identity(), {
            geometry_node(rotation(Axis::Y, 0.0), texture_phong_material("assets/images/textures/marble.png", 1.0, 0.0, 0.0, 1.0),
                          Triangle::from_vertices(dvec3(-10.0, -10.0, 0.0), dvec3(0.0, -10.0, 0.0), dvec3(10.0, 10.0, 0.0)), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(0.0, 0.0, 10.0), dvec3(0.0, 0.0, -10.0)), RenderConfig::default_()};
}
View tetrahedron_basic() {                                               // :157-173
    Scene scene = build_scene(
        {light1(), light2(), light3()}, no_ambient(), nullptr,
        scene_node(rotation(Axis::X, -50.0), {make_d4(30.0, translation(0.0, 0.0, 0.0))}));
    return {scene, image(512, 512), camera(dvec3(0.0, 20.0, 40.0), dvec3(0.0, 0.0, 0.0)), RenderConfig::default_()};
}
}  // namespace dice_scene

// =========================================================== DirectionLight (src/light.rs:82-120): no reference test uses it
// The room of polyhedron_basic lit by one directional light (plus a weak point light, so that both light
// paths meet in one Phong evaluation), a sphere and a cube to cast shadows, a mirror wall.
View direction_light_room() {
    using namespace mesh_test;
    Scene scene = build_scene(
        {DirectionLight::new_(dvec3(0.3, 1.0, 0.5), Color::new_(1.0, 0.9, 0.8), 0.8), light2()}, AmbientLight::new_(Color::WHITE, 0.1), nullptr,
        scene_node(DMat4::identity(), {
            create_room_from_material(700.0, RoomMaterialScheme{
                default_material(Color::WHITE), default_material(Color::WHITE), default_material(Color::RED),
                default_material(Color::CYAN), default_material(Color::GREEN), ReflectionShader::new_(Color::WHITE)}),
            geometry_node(translation(-120.0, -250.0, 0.0), default_material(Color::YELLOW), Sphere::from_radius(100.0), {}),
            geometry_node(translation(150.0, -270.0, 80.0) * rotation(Axis::Y, 30.0), default_material(Color::WHITE), Cube::new_(160.0), {}),
        }));
    return {scene, image(512, 512), camera(dvec3(-300.0, 0.0, 300.0), dvec3(350.0, -350.0, -350.0)), RenderConfig::default_()};
}

// A room holding the primitives no reference test instantiates directly as a node
// (Triangle, Polygon, Tetrahedron, Inverted, Plane) so that every Intersectable is covered.
View primitive_zoo() {
    using namespace mesh_test;
    Scene scene = build_scene(
        {light1(), light2(), light3()}, AmbientLight::new_(Color::WHITE, 0.2), nullptr,
        scene_node(DMat4::identity(), {
            coloured_room(),
            geometry_node(translation(-200.0, -100.0, -100.0) * scaling(100.0, 100.0, 100.0), default_material(Color::WHITE),
                          Triangle::from_vertices(dvec3(-1, 0, 0), dvec3(1, 0, 0), dvec3(0, 1.5, 0.3)), {}),
            geometry_node(translation(150.0, -150.0, -150.0), default_material(Color::ORANGE),
                          Polygon::from_vertices({dvec3(-80, 0, 0), dvec3(0, -60, 10), dvec3(80, 0, 0), dvec3(50, 90, 0), dvec3(-50, 90, 0)}), {}),
            geometry_node(translation(0.0, -250.0, 100.0) * rotation(Axis::Y, 20.0), default_material(Color::TEAL), Tetrahedron::new_(150.0), {}),
            geometry_node(translation(150.0, 100.0, 0.0), glass(), Inverted::new_(Sphere::from_radius(60.0)), {}),
            geometry_node(DMat4::identity(), default_material(Color::GRAY), Plane::new_(dvec3(0.0, -340.0, 0.0), dvec3(0.0, 1.0, 0.0)), {}),
        }));
    RenderConfig cfg = RenderConfig::default_();
    return {scene, image(512, 512), camera(dvec3(-310.0, 200.0, 300.0), dvec3(0.0, 0.0, 0.0)), cfg};
}

struct Entry { const char* name; std::function<View()> make; };
const std::vector<Entry>& registry() {
    static const std::vector<Entry> r = {
        {"basic_setup_scene", basic_setup_scene},
        {"white_sphere_one_light", white_sphere_one_light},
        {"plane_test_1", plane_test_1},
        {"cube_test_1", cube_test_1},
        {"translate_test_1", translate_test_1},
        {"rotate_test_1", rotate_test_1},
        {"scale_test_1", scale_test_1},
        {"child_transform_test_1", child_transform_test_1},
        {"shadow_test_1", shadow_test_1},
        {"mesh_basic", mesh_test::mesh_basic},
        {"glass_monkey", mesh_test::glass_monkey},
        {"monkey_1", [] { return mesh_test::monkey_n(1); }},
        {"monkey_2", [] { return mesh_test::monkey_n(2); }},
        {"monkey_3", [] { return mesh_test::monkey_n(3); }},
        {"square_light_basic", square_light_test::square_light_basic},
        {"light_room", square_light_test::light_room},
        {"subdivision_light", square_light_test::subdivision_light},
        {"scene_test_1", [] { return scene_test::scene_test_n(1); }},
        {"scene_test_2", [] { return scene_test::scene_test_n(2); }},
        {"scene_test_3", [] { return scene_test::scene_test_n(3); }},
        {"scene_test_4", [] { return scene_test::scene_test_n(4); }},
        {"scene_test_5", [] { return scene_test::scene_test_n(5); }},
        {"refraction_03", [] { return scene_test::refraction_test(3); }},
        {"background_1", scene_test::background_1},
        {"many_balls", scene_test::many_balls},
        {"many_balls_forest", scene_test::many_balls_forest},
        {"subtract_sphere", composite_shape_test::subtract_sphere},
        {"many_weirds", composite_shape_test::many_weirds},
        {"subtraction_room", composite_shape_test::subtraction_room},
        {"subtraction_outside1", composite_shape_test::subtraction_outside1},
        {"subtraction_outside3", composite_shape_test::subtraction_outside3},
        {"subtraction_outside4", composite_shape_test::subtraction_outside4},
        {"rounded_cube", composite_shape_test::rounded_cube},
        {"xor_test", composite_shape_test::xor_test},
        {"reuleaux", composite_shape_test::reuleaux},
        {"bump_map_basic_1", bump_map_basic_1},
        {"octahedron_basic", [] { return octahedron_basic_n(1); }},
        {"octahedron_basic_02", [] { return octahedron_basic_n(2); }},
        {"phong_test_1", [] { return phong_test_n(1); }},
        {"phong_test_2", [] { return phong_test_n(2); }},
        {"anti_alias_comparison", [] { return scene_test::scene_test_n(5); }},   // anti_alias_test.rs:10-27: scene_test_5's scene and camera
        {"scene_light_1_05", [] { return scene_test::scene_light_1(5); }},
        {"translucent_shadow", scene_test::translucent_shadow},
        {"bump_map_basic_2", bump_map_basic_2},
        {"transparent_shapes", transparency_test::transparent_shapes},
        {"transparent_orb", transparency_test::transparent_orb},
        {"normal_map_room_1", [] { return normal_map_room_n(1); }},
        {"normal_map_room_2", [] { return normal_map_room_n(2); }},
        {"texture_room", texture_test::texture_room},
        {"texture_cube", texture_test::texture_cube},
        {"skybox_test", texture_test::skybox_test},
        {"correlated_texture", texture_test::correlated_texture},
        {"plane_texture", texture_test::plane_texture},
        {"dice_scene_lo_res", [] { return dice_scene::dice_scene_view(false); }},
        {"dice_scene_hi_res", [] { return dice_scene::dice_scene_view(true); }},
        {"polygon_basic", dice_scene::polygon_basic},
        {"triangle_basic", dice_scene::triangle_basic},
        {"tetrahedron_basic", dice_scene::tetrahedron_basic},
        {"direction_light_room", direction_light_room},
        {"polyhedron_basic", polyhedron_basic},
        {"primitive_zoo", primitive_zoo},
    };
    return r;
}

thread_local std::string g_error;

}  // namespace

// ---------------------------------------------------------------- C interface for tests / bench
struct RthScene {
    SceneDescBuilder builder;
    RtSceneDesc desc;
    RtImageDim dim;
    RtCamera cam;
    RtRenderConfig cfg;
};

extern "C" {

const char* rth_last_error() { return g_error.c_str(); }
int rth_scene_count() { return (int)registry().size(); }
const char* rth_scene_name(int i) { return (i >= 0 && i < (int)registry().size()) ? registry()[i].name : nullptr; }

// Build the named scene and describe it.  Returns 0 on success.
int rth_scene_build(const char* name, const char* asset_root_dir, RthScene** out) {
    if (!name || !out) { g_error = "null argument"; return RT_ERR_INVALID_ARG; }
    try {
        if (asset_root_dir && *asset_root_dir) set_asset_root(asset_root_dir);
        for (auto& e : registry()) {
            if (std::string(name) != e.name) continue;
            View v = e.make();
            auto* s = new RthScene();
            v.scene.describe(s->builder);
            s->desc = s->builder.desc();
            s->dim = RtImageDim{v.dim.width, v.dim.height};
            s->cam = v.cam.pod();
            s->cfg = v.cfg.pod();
            *out = s;
            return RT_OK;
        }
        g_error = std::string("unknown scene: ") + name;
        return RT_ERR_INVALID_ARG;
    } catch (const std::exception& e) {
        g_error = e.what();
        return RT_ERR_INVALID_ARG;
    }
}
const RtSceneDesc* rth_scene_desc(const RthScene* s) { return s ? &s->desc : nullptr; }
// The image size, camera and config the reference test itself uses for this scene.
int rth_scene_view(const RthScene* s, RtImageDim* dim, RtCamera* cam, RtRenderConfig* cfg) {
    if (!s) return RT_ERR_INVALID_ARG;
    if (dim) *dim = s->dim;
    if (cam) *cam = s->cam;
    if (cfg) *cfg = s->cfg;
    return RT_OK;
}
void rth_scene_free(RthScene* s) { delete s; }

}  // extern "C"
