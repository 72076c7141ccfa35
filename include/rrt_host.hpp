// =============================================================================
// rrt_host.hpp -- header-only C++ host mirror of the reference's scene API over
// the C ABI (include/rrt.h, include/rrt_host.h), so that src/main.rs can be
// restated line for line without Rust (examples/main.cpp).  Names and argument
// order follow the reference; each item cites what it mirrors.  No tracing
// happens here: everything below RayTracer::render is librrt.so on the GPU.
// =============================================================================
#pragma once
#include <chrono>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

#include "rrt_host.h"

namespace rrt_host {

struct Vector3 { double x, y, z; };
using Point3 = Vector3;

// common.rs:87-93
inline Vector3 color_vec(unsigned char r, unsigned char g, unsigned char b) {
    return Vector3{(double)r / 255.0, (double)g / 255.0, (double)b / 255.0};
}

inline void check(int rc) {
    if (rc < 0) throw std::runtime_error(std::string("rrt: ") + rrt_last_error());   // the reference unwrap()s
}

// material.rs:13-43.  A value type, like the reference's (Material: Clone).
struct Material {
    Vector3 color; double albedo, lambert, reflective, phong; int alpha;
    int lobe = 0; double ior = 1.0;            // extension (rrt_add_material_ext)
    // Material::new_lambert_material(color, albedo, lambert, reflective, phong, alpha)   material.rs:20-33
    static Material new_lambert_material(Vector3 color, double albedo, double lambert, double reflective,
                                         double phong, int alpha) {
        return Material{color, albedo, lambert, reflective, phong, alpha};
    }
};

// common.rs:41-47: the closed set of entities the reference ships.
struct Sphere {                                  // sphere.rs:17-24
    Point3 position; double radius; Material material;
    static Sphere new_(Point3 position, double radius, Material material) { return Sphere{position, radius, material}; }
};
struct Model {                                   // model.rs:24-56
    std::string path; Material material; Point3 position; Vector3 scale;
    static Model new_(const std::string& path, Material material, Point3 position, Vector3 scale) {
        std::printf("Opening model @ %s\n", path.c_str());   // model.rs:25
        return Model{path, material, position, scale};
    }
};

struct Triangle {                                // triangle.rs:19-23: an Entity of its own (no kd-tree around it)
    Point3 v0, v1, v2; Vector3 normal; Material material;
    static Triangle new_(Point3 v0, Point3 v1, Point3 v2, Vector3 normal, Material material) {
        return Triangle{v0, v1, v2, normal, material};
    }
};

// scene.rs:31-47 with the scene!/entity_enum! macros (scene.rs:12-29) and #[derive(Entity)]
// (derive_entity/src/lib.rs) folded into one closed sum type: the enum those macros generate for
// `entity_enum!(SceneType, Sphere, Model)` is exactly std::variant<Sphere, Model>, and the derived
// `match self { V(a) => a.collide(ray) }` forwarding is what the kernels do on the entity kind.
//     Scene::new_(Point3{0,0,0}, {Model(burger), Sphere(sphere), Sphere(sphere2)})   main.rs:57-58
// A Scene is an Entity too (scene.rs:50-75), so Scene<Scene<...>> is a legal member type: SceneBox carries one
// (one level deep on the GPU: a member scene may not hold scenes itself).
struct Scene;
struct SceneBox { std::shared_ptr<const Scene> scene; };
struct Scene {
    using Member = std::variant<Sphere, Model, Triangle, SceneBox>;
    std::vector<Member> models; Point3 position;
    static Scene new_(Point3 position, std::vector<Member> models) { return Scene{std::move(models), position}; }
    SceneBox boxed() const { return SceneBox{std::make_shared<const Scene>(*this)}; }
};

// lighting.rs:19-32
struct DirectionalLight {
    Vector3 direction, color; double intensity;
    static DirectionalLight new_(Vector3 direction, Vector3 color, double intensity) { return DirectionalLight{direction, color, intensity}; }
};

// lighting.rs:52-68.  The reference never builds one (dead code); pushed to world.light_sources it is summed at its
// place in the list, see rrt_add_point_light / rrt_set_shadow_mode.
struct PointLight {
    Point3 position; Vector3 color; double brightness, radius;
    static PointLight new_(Point3 position, Vector3 color, double brightness, double radius) {
        return PointLight{position, color, brightness, radius};
    }
};

// common.rs:10-15 + RayTracer::new_empty_world (tracer.rs:53-75)
class World {
  public:
    World() : h_(rrt_scene_create()) { if (!h_) throw std::runtime_error("rrt_scene_create failed"); }
    ~World() { rrt_scene_destroy(h_); }
    World(const World&) = delete;
    World& operator=(const World&) = delete;
    World(World&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }

    // world.entities.push(Box::new(sphere))     main.rs:62
    int push(const Sphere& s) {
        const double c[3] = {s.position.x, s.position.y, s.position.z};
        const int id = rrt_add_sphere(h_, c, s.radius, material_id(s.material));
        check(id);
        return id;
    }
    // world.entities.push(Box::new(model))      main.rs:61
    int push(const Model& m) {
        const double p[3] = {m.position.x, m.position.y, m.position.z}, sc[3] = {m.scale.x, m.scale.y, m.scale.z};
        const int id = rrt_host_add_model(h_, m.path.c_str(), material_id(m.material), p, sc);
        check(id);
        return id;
    }
    // world.entities.push(Box::new(triangle))   Triangle implements Entity (triangle.rs:25-83)
    int push(const Triangle& t) {
        const double a[3] = {t.v0.x, t.v0.y, t.v0.z}, b[3] = {t.v1.x, t.v1.y, t.v1.z}, c[3] = {t.v2.x, t.v2.y, t.v2.z},
                     n[3] = {t.normal.x, t.normal.y, t.normal.z};
        const int id = rrt_add_triangle(h_, a, b, c, n, material_id(t.material));
        check(id);
        return id;
    }
    // world.entities.push(scene)                main.rs:59: the members are reachable through the
    // scene's kd-tree only (rrt_add_group); returns the scene's entity id
    int push(const SceneBox& b) { return push(*b.scene); }
    int push(const Scene& sc) {
        std::vector<int32_t> ids;
        for (const Scene::Member& m : sc.models)
            ids.push_back(std::visit([this](const auto& e) { return this->push(e); }, m));
        const int id = rrt_add_group(h_, ids.data(), (int32_t)ids.size());
        check(id);
        return id;
    }
    // world.light_sources.push(Box::new(light)) for a DirectionalLight (the list keeps push order, common.rs:12)
    int push(const DirectionalLight& l) {
        const double d[3] = {l.direction.x, l.direction.y, l.direction.z}, c[3] = {l.color.x, l.color.y, l.color.z};
        const int id = rrt_add_directional_light(h_, d, c, l.intensity);
        check(id);
        return id;
    }
    // world.light_sources.push(Box::new(light)) for a PointLight; shadow_mode 0 = PointLight::visible as written
    // (lighting.rs:70-84), 1 = the occlusion test it was meant to be
    void push(const PointLight& l, int shadow_mode = 0) {
        const double p[3] = {l.position.x, l.position.y, l.position.z}, c[3] = {l.color.x, l.color.y, l.color.z};
        check(rrt_add_point_light(h_, p, c, l.brightness, l.radius));
        check(rrt_set_shadow_mode(h_, shadow_mode));
    }
    // entity.translate(vec)                     common.rs:46 / model.rs:80-86 / kdtree.rs:35-42: as written the
    // boxes of a model's tree stay where they were (refit = false); refit = true moves them too
    void translate(int entity, Vector3 vec, bool refit = false) {
        const double v[3] = {vec.x, vec.y, vec.z};
        check(rrt_translate_entity(h_, entity, v, refit ? 1 : 0));
    }
    rrt_scene* raw() const { return h_; }

  private:
    friend class RayTracer;
    int material_id(const Material& m) {
        const double c[3] = {m.color.x, m.color.y, m.color.z};
        const int id = m.lobe == 0 ? rrt_add_material(h_, c, m.albedo, m.lambert, m.reflective, m.phong, m.alpha)
                                   : rrt_add_material_ext(h_, c, m.albedo, m.lambert, m.reflective, m.phong, m.alpha,
                                                          m.lobe, m.ior);
        check(id);
        return id;
    }
    rrt_scene* h_;
};

// tracer.rs:11-24
class RayTracer {
  public:
    // RayTracer::new_default_renderer((w, h))   tracer.rs:38-51
    static RayTracer new_default_renderer(unsigned w, unsigned h) {
        RayTracer r;
        rrt_default_camera(&r.camera_);
        rrt_default_settings(&r.settings_, w, h);
        return r;
    }
    // RayTracer::new_empty_world(skybox)        tracer.rs:53-75
    static World new_empty_world(const std::string& skybox) {
        World w;
        const double dir[3] = {1.0, -1.0, 1.0};
        const Vector3 lc = color_vec(230, 230, 230);
        const double col[3] = {lc.x, lc.y, lc.z};
        check(rrt_set_directional_light(w.raw(), dir, col, 2.0));
        std::printf("\nLoading assets...");                   // cubemap.rs:17
        check(rrt_host_load_cubemap(w.raw(), skybox.c_str()));
        std::printf(".......\n");
        check(rrt_set_ambient(w.raw(), 0.15));
        return w;
    }
    rrt_settings& settings() { return settings_; }
    // raytracer.render(output, world)           tracer.rs:77-175
    void render(const std::string& output, World& world) {
        std::printf("Rendering...\n");
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<unsigned char> img((size_t)settings_.width * settings_.height * 3);
        check(rrt_render(world.raw(), &camera_, &settings_, img.data(), nullptr, nullptr, nullptr));
        if (rrt_host_save_png_rgb8(output.c_str(), img.data(), settings_.width, settings_.height) == 0)
            std::printf("Saved!\n");
        else
            std::printf("%s\n", rrt_last_error());            // tracer.rs:167-170: the error is printed and swallowed
        std::printf("\n\n");
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0);
        std::printf("Finished in %lldms\n", (long long)ms.count());
    }

  private:
    rrt_camera camera_{};
    rrt_settings settings_{};
};

}  // namespace rrt_host
