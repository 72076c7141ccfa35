// src/main.rs restated on the C++ host mirror: same scene literals, same call order, output
// ./bruh.png.  Everything under raytracer.render() runs in librrt.so on the MI355X.
//   build:  make -C examples          run (from the repo root):  examples/rrt_main [out.png] [W H] [scene]
// A fourth argument "scene" takes main.rs's commented-out lines 57-59 instead of 61-63: the three
// entities go into one Scene<T> (kd-tree over entities) and that is the world's only entity.
#include <cstdlib>

#include "../include/rrt_host.hpp"

using namespace rrt_host;

int main(int argc, char** argv) {
    std::printf("MAIN!\n");
    const unsigned w = argc > 3 ? (unsigned)std::atoi(argv[2]) : 3840, h = argc > 3 ? (unsigned)std::atoi(argv[3]) : 2160;

    RayTracer raytracer = RayTracer::new_default_renderer(w, h);                                    // main.rs:22
    World world = RayTracer::new_empty_world("./assets/cubemaps/hd_blue_sunset");                   // main.rs:24

    Material mat1 = Material::new_lambert_material(color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20);   // :26
    Material mat2 = Material::new_lambert_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20);          // :27

    Sphere sphere = Sphere::new_(Point3{-3.0, 0.0, 5.0}, 1.0, mat1);                                // :29-37
    Sphere sphere2 = Sphere::new_(Point3{2.0, 0.0, 8.0}, 1.0, mat2);                                // :38-46

    Model burger = Model::new_("./assets/obj/ufo_fix.obj",                                           // :48-53
                               Material::new_lambert_material(color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20),
                               Point3{0.0, 30.0, 70.0}, Vector3{1.0, -1.0, 1.0});

    if (argc > 4 && std::string(argv[4]) == "scene") {
        // entity_enum!(SceneType, Sphere, Model);                                                  // :57
        Scene scene = Scene::new_(Point3{0., 0., 0.}, {burger, sphere, sphere2});                   // :58
        world.push(scene);                                                                          // :59
    } else {
        world.push(burger);                                                                         // :61
        world.push(sphere);                                                                         // :62
        world.push(sphere2);                                                                        // :63
    }

    raytracer.render(argc > 1 ? argv[1] : "./bruh.png", world);                                      // :65
    return 0;
}
