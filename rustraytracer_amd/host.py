"""Python host-side mirror of the reference's scene API over the C ABI.

Names follow the reference (src/main.rs, src/material.rs, src/tracer.rs):
World ~ `World` + `RayTracer::new_empty_world`; add_material ~
`Material::new_lambert_material`; add_sphere ~ `Sphere::new` + push;
add_model ~ `Model::new` + push; set_sky ~ `Material::new_sky_material`;
render ~ `RayTracer::render` minus the PNG save.  (The compiled-language host
mirror is include/rrt_host.hpp + examples/main.cpp; this module is the thin
driver the tests and bench.py use.)  Everything executes in librrt.so on the
GPU; there is no CPU path here.
"""
import ctypes as C

import numpy as np

from . import binding as B


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_camera():
    cam = B.Camera()
    B.lib.rrt_default_camera(C.byref(cam))
    return cam


def make_settings(width, height, spp=1, max_bounce=2, seed=0, prune_zero_mix=0,
                  tile_rank=0, tile_world=1, batch_spp=0, tile_deal=0):
    return B.Settings(width, height, spp, max_bounce, seed, prune_zero_mix, tile_rank, tile_world, batch_spp, tile_deal)


class loaders:
    """Asset loaders implemented in the product's C++ host layer (host_io.cpp)."""

    @staticmethod
    def load_model(path, position, scale, triangulate=False):
        v, n, cnt = C.c_void_p(), C.c_void_p(), C.c_int64()
        B.check(B.lib.rrt_host_load_model_ex(path.encode(), _d3(position), _d3(scale), 1 if triangulate else 0,
                                             C.byref(v), C.byref(n), C.byref(cnt)))
        try:
            nt = cnt.value
            verts = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(nt, 3, 3)).copy()
            normals = np.ctypeslib.as_array(C.cast(n, C.POINTER(C.c_double)), shape=(nt, 3)).copy()
        finally:
            B.lib.rrt_host_free(v)
            B.lib.rrt_host_free(n)
        return verts, normals

    @staticmethod
    def load_png(path):
        p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
        B.check(B.lib.rrt_host_load_png_rgb8(path.encode(), C.byref(p), C.byref(w), C.byref(h)))
        try:
            img = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
        finally:
            B.lib.rrt_host_free(p)
        return img

    @staticmethod
    def load_cubemap(folder):
        import os
        return [loaders.load_png(os.path.join(folder, n + ".png"))
                for n in ("left", "right", "down", "up", "front", "back")]   # cubemap.rs:47-48

    @staticmethod
    def save_png(path, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        B.check(B.lib.rrt_host_save_png_rgb8(path.encode(), _ptr(rgb), rgb.shape[1], rgb.shape[0]))


class World:
    def __init__(self, device=None):
        if device is not None:
            B.check(B.lib.rrt_set_device(int(device)))
        self.h = C.c_void_p(B.lib.rrt_scene_create())
        if not self.h:
            raise B.RrtError("rrt_scene_create failed")

    def close(self):
        if getattr(self, "h", None):
            B.lib.rrt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene description ---------------------------------------------------
    def add_material(self, color, albedo, lambert, reflective, phong, alpha):
        return B.check(B.lib.rrt_add_material(self.h, _d3(color), albedo, lambert, reflective, phong, int(alpha)))

    def add_material_ext(self, color, albedo, lambert, lobe_mix, phong, alpha, lobe, ior=1.0):
        return B.check(B.lib.rrt_add_material_ext(self.h, _d3(color), albedo, lambert, lobe_mix, phong, int(alpha),
                                                  int(lobe), float(ior)))

    def add_sphere(self, center, radius, mat):
        return B.check(B.lib.rrt_add_sphere(self.h, _d3(center), float(radius), mat))

    def add_mesh(self, verts, normals, mat):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        if v.shape[0] != n.shape[0]:
            raise ValueError("verts / normals triangle counts differ")
        return B.check(B.lib.rrt_add_mesh(self.h, _ptr(v), _ptr(n), v.shape[0], mat))

    def add_triangle(self, v0, v1, v2, normal, mat):
        """Triangle::new + push (triangle.rs:19-23): a bare triangle as an entity of its own."""
        return B.check(B.lib.rrt_add_triangle(self.h, _d3(v0), _d3(v1), _d3(v2), _d3(normal), mat))

    def add_group(self, members):
        """Scene::new over entities already added (scene.rs:39-47); returns the group's entity id."""
        m = np.ascontiguousarray(members, dtype=np.int32)
        return B.check(B.lib.rrt_add_group(self.h, _ptr(m), m.size))

    def add_model(self, path, mat, position, scale):
        return B.check(B.lib.rrt_host_add_model(self.h, path.encode(), mat, _d3(position), _d3(scale)))

    def set_sky(self, faces):
        for i, f in enumerate(faces):
            f = np.ascontiguousarray(f, dtype=np.uint8)
            B.check(B.lib.rrt_set_sky_face(self.h, i, _ptr(f), f.shape[1], f.shape[0]))

    def set_directional_light(self, direction, color, intensity):
        B.check(B.lib.rrt_set_directional_light(self.h, _d3(direction), _d3(color), float(intensity)))

    def add_directional_light(self, direction, color, intensity):
        """DirectionalLight::new + world.light_sources.push (lighting.rs:25-32): appends; returns the light's index."""
        return B.check(B.lib.rrt_add_directional_light(self.h, _d3(direction), _d3(color), float(intensity)))

    def set_ambient(self, a):
        B.check(B.lib.rrt_set_ambient(self.h, float(a)))

    def add_point_light(self, position, color, brightness, attenuation):
        """PointLight{position, color, brightness, attenuation} (lighting.rs:52-57) pushed to world.light_sources."""
        return B.check(B.lib.rrt_add_point_light(self.h, _d3(position), _d3(color), float(brightness), float(attenuation)))

    def set_shadow_mode(self, mode):
        B.check(B.lib.rrt_set_shadow_mode(self.h, int(mode)))

    def translate(self, ent, vec, refit=0):
        """Entity::translate (sphere.rs:68-70, model.rs:80-86, scene.rs:68-74); refit=1 also moves the boxes."""
        B.check(B.lib.rrt_translate_entity(self.h, int(ent), _d3(vec), int(refit)))

    def occluded_rays(self, rays, lim2):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        lim2 = np.ascontiguousarray(np.broadcast_to(np.asarray(lim2, dtype=np.float64), (rays.shape[0],)))
        out = np.zeros(rays.shape[0], np.uint8)
        B.check(B.lib.rrt_occluded_rays(self.h, _ptr(rays), _ptr(lim2), rays.shape[0], _ptr(out)))
        return out

    def set_work_budget(self, nbytes):
        B.check(B.lib.rrt_set_work_budget(self.h, int(nbytes)))

    def set_cull_mode(self, mode):
        B.check(B.lib.rrt_set_cull_mode(self.h, int(mode)))

    def set_timing(self, level):
        B.check(B.lib.rrt_set_timing(self.h, int(level)))

    def set_sky_stream(self, mode):
        """1 (default): the sky-only tiles' kernel on a side stream behind the primary kernel; 0: on the frame's own stream."""
        B.check(B.lib.rrt_set_sky_stream(self.h, int(mode)))

    def commit(self):
        B.check(B.lib.rrt_scene_commit(self.h))

    # -- queries -------------------------------------------------------------
    def tree(self, ent):
        nn, nr = C.c_int32(), C.c_int32()
        B.check(B.lib.rrt_mesh_tree_sizes(self.h, ent, C.byref(nn), C.byref(nr)))
        nn, nr = nn.value, nr.value
        out = dict(boxes=np.zeros((nn, 6)), split=np.zeros(nn), axis=np.zeros(nn, np.int32),
                   child=np.zeros((nn, 2), np.int32), ropes=np.zeros((nn, 6), np.int32),
                   leaf_refs=np.zeros(nr, np.int32))
        B.check(B.lib.rrt_mesh_tree_export(self.h, ent, _ptr(out["boxes"]), _ptr(out["split"]), _ptr(out["axis"]),
                                           _ptr(out["child"]), _ptr(out["ropes"]), _ptr(out["leaf_refs"])))
        return out

    def cover_mask(self, ent, width, height, camera=None):
        """The primary-ray coverage mask of mesh entity `ent` (rrt_cover_mask) as a bool image + the "every pixel" flag."""
        cam = camera or default_camera()
        px = np.zeros((height, width), np.uint8)
        every = C.c_int(0)
        B.check(B.lib.rrt_cover_mask(self.h, C.byref(cam), width, height, ent, _ptr(px), C.byref(every)))
        return px.astype(bool), bool(every.value)

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(ent=np.zeros(n, np.int32), tri=np.zeros(n, np.int32), t=np.zeros(n),
                   pos=np.zeros((n, 3)), normal=np.zeros((n, 3)))
        B.check(B.lib.rrt_trace_rays(self.h, _ptr(rays), n, _ptr(out["ent"]), _ptr(out["tri"]), _ptr(out["t"]),
                                     _ptr(out["pos"]), _ptr(out["normal"])))
        return out

    def cast_rays(self, rays, max_bounce=2, prune=0):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        col = np.zeros((rays.shape[0], 3))
        B.check(B.lib.rrt_cast_rays(self.h, _ptr(rays), rays.shape[0], max_bounce, prune, _ptr(col)))
        return col

    def render(self, width, height, spp=1, max_bounce=2, seed=0, prune_zero_mix=0, batch_spp=0,
               camera=None, want=("rgb", "ids", "t", "color")):
        cam = camera or default_camera()
        st = make_settings(width, height, spp, max_bounce, seed, prune_zero_mix, 0, 1, batch_spp)
        out = {}
        out["rgb"] = np.zeros((height, width, 3), np.uint8) if "rgb" in want else None
        out["ids"] = np.full((height, width, 2), -2, np.int32) if "ids" in want else None
        out["t"] = np.zeros((height, width)) if "t" in want else None
        out["color"] = np.zeros((height, width, 3)) if "color" in want else None
        B.check(B.lib.rrt_render(self.h, C.byref(cam), C.byref(st), _ptr(out["rgb"]), _ptr(out["ids"]),
                                 _ptr(out["t"]), _ptr(out["color"])))
        out["stats"] = self.stats()
        return out

    def render_device(self, settings, d_rgb8, d_ids=None, d_t=None, d_color=None, stream=None, camera=None):
        """Asynchronous render into device pointers (ints, e.g. torch.Tensor.data_ptr())."""
        cam = camera or default_camera()
        B.check(B.lib.rrt_render_device(self.h, C.byref(cam), C.byref(settings), d_rgb8, d_ids, d_t, d_color, stream))

    def render_device_noalloc(self, settings, d_rgb8, d_ids=None, d_t=None, d_color=None, stream=None, camera=None):
        """rrt_render_device_noalloc: same, over the library's static scratch (allocates nothing)."""
        cam = camera or default_camera()
        B.check(B.lib.rrt_render_device_noalloc(self.h, C.byref(cam), C.byref(settings), d_rgb8, d_ids, d_t, d_color, stream))

    def stats(self):
        st = B.Stats()
        B.check(B.lib.rrt_get_stats(self.h, C.byref(st)))
        d = {k: getattr(st, k) for k, _ in B.Stats._fields_}
        d["level_rays"] = [int(x) for x in st.level_rays]
        return d


def tile_slots(width, height, world):
    return int(B.lib.rrt_tile_slots(width, height, world))


def tile_buffer_bytes(width, height, world, dealt=0):
    return int(B.lib.rrt_tile_buffer_bytes(width, height, world, int(dealt)))


def assemble_tiles(d_gathered, width, height, world, d_rgb8, stream=None, dealt=0):
    if dealt:
        B.check(B.lib.rrt_assemble_tiles_dealt(d_gathered, width, height, world, d_rgb8, stream))
    else:
        B.check(B.lib.rrt_assemble_tiles(d_gathered, width, height, world, d_rgb8, stream))
