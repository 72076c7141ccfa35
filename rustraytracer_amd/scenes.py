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


# ---------------------------------------------------------------------------
# Extension configs (BASELINE configs[2..4]).  The reference has no refraction, no stochastic
# bounce and no synthetic mesh; these scenes exercise the build's documented extensions
# (include/rrt.h: rrt_add_material_ext) and are checked GPU-vs-oracle only.
# ---------------------------------------------------------------------------
LOBE_REFLECT, LOBE_REFRACT, LOBE_DIFFUSE = 0, 1, 2


def cfg3_scene(b, loaders):
    """configs[2]: reflect/refract glass-sphere scene, deterministic bounces (render with
    max_bounce=8, spp=1): a mirror sphere, two glass spheres (ior 1.5 / 1.33), a tinted
    half-mirror and a large matte ground sphere, blue_sunset sky."""
    empty_world(b, loaders, "blue_sunset")
    mirror = b.add_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    glass = b.add_material_ext(color_vec(10, 10, 10), 0.8, 0.05, 0.95, 0.3, 20, LOBE_REFRACT, 1.5)
    water = b.add_material_ext(color_vec(10, 20, 30), 0.8, 0.05, 0.9, 0.2, 20, LOBE_REFRACT, 1.33)
    tinted = b.add_material(color_vec(200, 80, 60), 0.9, 0.7, 0.3, 0.2, 20)
    matte = b.add_material(color_vec(90, 110, 90), 1.0, 1.0, 0.0, 0.0, 20)
    b.add_sphere((0.0, 0.0, 9.0), 1.6, glass)
    b.add_sphere((-3.6, 0.3, 8.0), 1.3, mirror)
    b.add_sphere((3.4, 0.6, 8.5), 1.2, water)
    b.add_sphere((1.2, -1.2, 5.5), 0.6, tinted)
    b.add_sphere((0.0, 1002.0, 9.0), 1000.0, matte)      # world +y is image-down: a floor
    return b


def cfg4_scene(b, loaders):
    """configs[3]: the UFO scene with a stochastic diffuse bounce on the UFO (path tracing)."""
    empty_world(b, loaders, "blue_sunset")
    mat1 = b.add_material(color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
    mat2 = b.add_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    ufo = b.add_material_ext(color_vec(100, 100, 50), 1.0, 0.6, 0.5, 0.3, 20, LOBE_DIFFUSE)
    verts, normals = loaders.load_model(os.path.join(ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
    b.add_mesh(verts, normals, ufo)
    b.add_sphere((-3.0, 0.0, 5.0), 1.0, mat1)
    b.add_sphere((2.0, 0.0, 8.0), 1.0, mat2)
    return b


def heightfield_mesh(n=1000, seed=1):
    """configs[4]'s synthetic mesh: an n x n jittered height-field grid (2 triangles per cell,
    2*n*n in total) over x,y in [-50,50], facing the camera around z = 90; heights from a seeded
    integer hash, vertices rounded to f32 like an OBJ loader would, flat geometric normals."""
    import numpy as np
    g = n + 1
    ix, iy = np.meshgrid(np.arange(g, dtype=np.uint64), np.arange(g, dtype=np.uint64), indexing="xy")

    def hash01(a, b, s):
        h = (a * np.uint64(0x9E3779B97F4A7C15) + b * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(s) * np.uint64(0x165667B19E3779F9))
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        return (h & np.uint64(0xFFFFFF)).astype(np.float64) / float(1 << 24)

    cell = 100.0 / n
    x = -50.0 + ix.astype(np.float64) * cell + (hash01(ix, iy, seed) - 0.5) * 0.3 * cell
    y = -50.0 + iy.astype(np.float64) * cell + (hash01(ix, iy, seed + 1) - 0.5) * 0.3 * cell
    # smooth bumps from a coarse lattice + fine roughness
    cx, cy = ix // np.uint64(max(n // 16, 1)), iy // np.uint64(max(n // 16, 1))
    fx = (ix % np.uint64(max(n // 16, 1))).astype(np.float64) / max(n // 16, 1)
    fy = (iy % np.uint64(max(n // 16, 1))).astype(np.float64) / max(n // 16, 1)
    c00, c10 = hash01(cx, cy, seed + 2), hash01(cx + np.uint64(1), cy, seed + 2)
    c01, c11 = hash01(cx, cy + np.uint64(1), seed + 2), hash01(cx + np.uint64(1), cy + np.uint64(1), seed + 2)
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    coarse = (c00 * (1 - sx) + c10 * sx) * (1 - sy) + (c01 * (1 - sx) + c11 * sx) * sy
    z = 90.0 + 12.0 * (coarse - 0.5) + 0.5 * cell * (hash01(ix, iy, seed + 3) - 0.5)
    P = np.stack([x, y, z], axis=-1).astype(np.float32).astype(np.float64)
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    t1 = np.stack([p00, p10, p11], axis=2)
    t2 = np.stack([p00, p11, p01], axis=2)
    verts = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], axis=0)
    nrm = np.cross(verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0])
    nrm = nrm * (1.0 / np.sqrt((nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1]) + nrm[:, 2] * nrm[:, 2]))[:, None]
    nrm[nrm[:, 2] > 0] *= -1.0          # face the camera (which looks along +z)
    return np.ascontiguousarray(verts), np.ascontiguousarray(nrm)


def cfg5_scene(b, loaders, n=1000):
    """configs[4]: synthetic 2*n*n-triangle mesh (n=1000 -> 2M), diffuse bounces, sky cubemap."""
    empty_world(b, loaders, "sky")
    rough = b.add_material_ext(color_vec(150, 140, 120), 1.0, 0.7, 0.5, 0.1, 20, LOBE_DIFFUSE)
    verts, normals = heightfield_mesh(n)
    b.add_mesh(verts, normals, rough)
    return b


# ---------------------------------------------------------------------------
# Scene<T> (src/geometry/scene.rs): entities gathered under a kd-tree over whole entities.
# ---------------------------------------------------------------------------
def main_rs_as_scene(b, loaders, skybox="hd_blue_sunset"):
    """main.rs with its commented-out lines 57-59 enabled instead of 61-63:
    scene!(SceneType, origin, Model(burger), Sphere(sphere), Sphere(sphere2)) is the world's only
    entity.  Three members < 5, so the group's tree is a single leaf behind the union box."""
    main_rs_scene(b, loaders, skybox)
    b.add_group([0, 1, 2])
    return b


def many_entity_scene(b, loaders, n_spheres=60, seed=7, grouped=True):
    """A many-entity world (what Scene<T> is for): the UFO, the cube, a seeded cloud of matte,
    mirror and glass spheres, all under one group; one more sphere stays in the flat entity list."""
    import numpy as np
    empty_world(b, loaders, "sky")
    matte = b.add_material(color_vec(180, 90, 60), 1.0, 0.9, 0.1, 0.3, 20)
    mirror = b.add_material(color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    glass = b.add_material_ext(color_vec(10, 10, 10), 0.8, 0.05, 0.95, 0.3, 20, LOBE_REFRACT, 1.5)
    ufo = b.add_material(color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20)
    members = []
    v, n = loaders.load_model(os.path.join(ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
    members.append(b.add_mesh(v, n, ufo))
    v, n = loaders.load_model(os.path.join(ASSETS, "obj", "cube.obj"), (-9.0, 2.0, 20.0), (2.0, 2.0, 2.0))
    members.append(b.add_mesh(v, n, matte))
    rng = np.random.default_rng(seed)
    for i in range(n_spheres):
        c = rng.uniform((-14.0, -8.0, 8.0), (14.0, 8.0, 40.0))
        r = float(rng.uniform(0.4, 1.6))
        members.append(b.add_sphere(tuple(float(x) for x in c), r, (matte, mirror, glass)[i % 3]))
    if grouped:
        b.add_group(members)
    b.add_sphere((0.0, -3.0, 6.0), 0.7, mirror)          # stays in world.entities
    return b


def cfg2_as_scene(b, loaders):
    """cfg2's world with the three entities under one Scene<T> (bench: cost of the union-box gate)."""
    return main_rs_as_scene(b, loaders, skybox="blue_sunset")


def many_entity_flat(b, loaders):
    return many_entity_scene(b, loaders, grouped=False)
