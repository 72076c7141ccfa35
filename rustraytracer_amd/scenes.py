"""Scene definitions shared by tests and bench.py.

Each function drives a *builder* (rustraytracer_amd.host.World for the HIP
path, oracle.oracle.OracleScene for the CPU oracle -- same method names) and a
*loaders* object providing load_model(path, position, scale) and
load_cubemap(folder), so both sides are fed bit-identical inputs from
independent loader implementations.
"""
import os

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def color_vec(r, g, b):
    """src/common.rs:87-93"""
    return (float(r) / 255.0, float(g) / 255.0, float(b) / 255.0)


def empty_world(b, loaders, skybox):
    """RayTracer::new_empty_world (src/tracer.rs:53-75)."""
    b.set_directional_light((1.0, -1.0, 1.0), color_vec(230, 230, 230), 2.0)
    b.set_sky(loaders.load_cubemap(os.path.join(ASSETS, "cubemaps", skybox)))
    b.set_ambient(0.15)


def main_rs_scene(b, loaders, skybox="hd_blue_sunset"):
    """The reference's one concrete scene (src/main.rs:22-65).
    Entity ids follow world.entities.push order: 0 = UFO model, 1 = sphere,
    2 = sphere2 (main.rs:61-63)."""
    empty_world(b, loaders, skybox)
    mat1 = b.add_material(color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
    mat2 = b.add_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    mat3 = b.add_material(color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20)
    verts, normals = loaders.load_model(os.path.join(ASSETS, "obj", "ufo_fix.obj"),
                                        (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
    b.add_mesh(verts, normals, mat3)
    b.add_sphere((-3.0, 0.0, 5.0), 1.0, mat1)
    b.add_sphere((2.0, 0.0, 8.0), 1.0, mat2)
    return b


def cfg1_scene(b, loaders):
    """BASELINE.json configs[0]: obj/cube.obj + 3 spheres, `sky` cubemap.
    The reference fixes no placement for this config; frozen here:
    cube at (0,0,12) scale 2 (11 triangles after model.rs:31 drops the last
    face), the two main.rs spheres plus a third half-mirror sphere."""
    empty_world(b, loaders, "sky")
    mat1 = b.add_material(color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
    mat2 = b.add_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    mat3 = b.add_material(color_vec(200, 120, 60), 1.0, 1.0, 0.0, 0.3, 20)
    mat4 = b.add_material(color_vec(40, 160, 80), 0.9, 0.6, 0.4, 0.2, 20)
    verts, normals = loaders.load_model(os.path.join(ASSETS, "obj", "cube.obj"),
                                        (0.0, 0.0, 12.0), (2.0, 2.0, 2.0))
    b.add_mesh(verts, normals, mat3)
    b.add_sphere((-3.0, 0.0, 5.0), 1.0, mat1)
    b.add_sphere((2.0, 0.0, 8.0), 1.0, mat2)
    b.add_sphere((0.5, -2.5, 9.0), 1.5, mat4)
    return b


def cfg2_scene(b, loaders):
    """BASELINE.json configs[1] (the bench workload): the main.rs world
    (ufo_fix.obj kd-tree + the two spheres; diffuse+specular material of
    main.rs:50) under the blue_sunset cubemap (6 x 2048^2)."""
    return main_rs_scene(b, loaders, skybox="blue_sunset")
