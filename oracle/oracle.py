"""ctypes front-end of the CPU parity oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under rustraytracer_amd/ imports this.

Also restates, in numpy, the host-side pieces of the reference that sit above
the C ABI on the oracle side:
  * Model::new            (src/geometry/model.rs:24-56, 89-103)  -> load_model()
  * CubemapBehavior::new  (src/behavior/cubemap.rs:16-52)        -> load_cubemap()
  * main.rs scene         (src/main.rs:22-65)                    -> in scenes.py
These use numpy float32 parsing + PIL decoding, independent of the product's
C++ loaders (rustraytracer_amd/csrc/host_io.cpp), so the two cross-check.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Camera(C.Structure):
    _fields_ = [("eye", C.c_double * 3), ("lens_z", C.c_double),
                ("lens_w", C.c_double), ("lens_h", C.c_double)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32),
                ("max_bounce", C.c_uint32), ("seed", C.c_uint64),
                ("prune_zero_mix", C.c_uint32), ("n_threads", C.c_uint32)]


def default_camera():
    """tracer.rs:38-51 + :84-110 : eye (0,0,0), lens plane z=75, 160x90."""
    return Camera((C.c_double * 3)(0.0, 0.0, 0.0), 75.0, 160.0, 90.0)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rrt_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_set_sort_mode.argtypes = [C.c_int]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_add_material.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]
        L.orc_add_material_ext.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double]
        L.orc_add_sphere.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_double, C.c_int]
        L.orc_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_add_group.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_add_group.restype = C.c_int
        L.orc_set_sky_face.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_set_directional_light.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
        L.orc_add_directional_light.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double]
        L.orc_add_triangle.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.c_int]
        L.orc_set_ambient.argtypes = [C.c_void_p, C.c_double]
        L.orc_add_point_light.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double]
        L.orc_set_shadow_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_translate_entity.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.orc_occluded_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_mesh_tree_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_mesh_tree_export.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        L.orc_cast_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Settings), C.c_int64, C.c_int64, C.c_int64] + [C.c_void_p] * 5
        L.orc_render.restype = C.c_int
        L.orc_aabb_collide.argtypes = [C.c_void_p] * 4
        L.orc_triangle_collide.argtypes = [C.c_void_p] * 4 + [C.POINTER(C.c_double)]
        L.orc_sphere_collide.argtypes = [C.c_void_p, C.c_double] + [C.c_void_p] * 4 + [C.POINTER(C.c_double)]
        L.orc_cubemap.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_powi.argtypes = [C.c_double, C.c_int]
        L.orc_powi.restype = C.c_double
        L.orc_philox.argtypes = [C.c_void_p] * 3
        L.orc_primary_rays_rows.argtypes = [C.POINTER(Camera), C.POINTER(Settings), C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]
        L.orc_set_count_box_hits.argtypes = [C.c_int]
        L.orc_set_static_chunks.argtypes = [C.c_int]
        L.orc_primary_ray.argtypes = [C.POINTER(Camera), C.POINTER(Settings), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _LIB = L
    return _LIB


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def color_vec(r, g, b):
    """common.rs:87-93"""
    return (float(r) / 255.0, float(g) / 255.0, float(b) / 255.0)


# ---------------------------------------------------------------------------
# Model::new restated (model.rs:24-56).  obj-rs 0.6.2 semantics: numbers parsed
# as f32; faces must be triangles with position+normal ("v//vn" or "v/vt/vn").
# ---------------------------------------------------------------------------
def parse_obj(path, triangulate=False):
    """triangulate: EXTENSION (obj-rs has no such mode) -- a polygon becomes the fan (c0, ci, ci+1)."""
    pos, nrm, faces = [], [], []
    with open(path, "r") as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                pos.append([np.float32(t) for t in tok[1:4]])
            elif tok[0] == "vn":
                nrm.append([np.float32(t) for t in tok[1:4]])
            elif tok[0] == "f":
                if len(tok) < 4 or (len(tok) != 4 and not triangulate):
                    raise ValueError("obj-rs refuses untriangulated faces: " + line.strip())
                corner = []
                for t in tok[1:]:
                    parts = t.split("/")
                    if len(parts) != 3 or parts[2] == "":
                        raise ValueError("face without normal index: " + line.strip())
                    pi, ni = int(parts[0]), int(parts[2])
                    pi = pi - 1 if pi > 0 else len(pos) + pi
                    ni = ni - 1 if ni > 0 else len(nrm) + ni
                    corner.append((pi, ni))
                for i in range(1, len(corner) - 1):
                    faces.append([corner[0], corner[i], corner[i + 1]])
    P = np.asarray(pos, dtype=np.float32).reshape(-1, 3)
    N = np.asarray(nrm, dtype=np.float32).reshape(-1, 3)
    F = np.asarray(faces, dtype=np.int64).reshape(-1, 3, 2)
    return P, N, F


def _normalize_rows(v):
    # cgmath normalize: v * (1/sqrt((x*x+y*y)+z*z))
    m2 = (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]
    return v * (1.0 / np.sqrt(m2))[..., None]


def load_model(path, position, scale, triangulate=False):
    """Returns (verts[N,3,3] f64, normals[N,3] f64) with the last face dropped
    (model.rs:31), T*S transform on positions (model.rs:29-41) and averaged,
    untransformed flat normals (model.rs:35-42)."""
    P, Nn, F = parse_obj(path, triangulate)
    nfaces = F.shape[0]
    n_idx = 3 * nfaces
    # for i in (0..indices.len()-4).step_by(3)
    keep = len(range(0, n_idx - 4, 3))
    F = F[:keep]
    p = P[F[:, :, 0]].astype(np.float64)           # vertex2point (model.rs:89-95)
    n = _normalize_rows(Nn[F[:, :, 1]].astype(np.float64))  # vertex2normal (model.rs:97-103)
    s = np.asarray(scale, dtype=np.float64)
    t = np.asarray(position, dtype=np.float64)
    verts = p * s + t                               # fl(fl(s*x)+t)
    normals = _normalize_rows((n[:, 0] + n[:, 1]) + n[:, 2])
    return np.ascontiguousarray(verts), np.ascontiguousarray(normals)


FACE_FILES = ["left", "right", "down", "up", "front", "back"]   # cubemap.rs:47-48


def load_cubemap(folder):
    """image::open(..).to_rgb8() for the six faces, in maps[] order."""
    from PIL import Image
    faces = []
    for name in FACE_FILES:
        im = Image.open(os.path.join(folder, name + ".png")).convert("RGB")
        faces.append(np.ascontiguousarray(np.asarray(im, dtype=np.uint8)))
    return faces


class OracleScene:
    """Mirror of World + entity list, backed by liboracle.so."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_create())
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_material(self, color, albedo, lambert, reflective, phong, alpha):
        return self.L.orc_add_material(self.h, _d3(color), albedo, lambert, reflective, phong, int(alpha))

    def add_material_ext(self, color, albedo, lambert, lobe_mix, phong, alpha, lobe, ior=1.0):
        return self.L.orc_add_material_ext(self.h, _d3(color), albedo, lambert, lobe_mix, phong, int(alpha), int(lobe), float(ior))

    def add_sphere(self, center, radius, mat):
        return self.L.orc_add_sphere(self.h, _d3(center), float(radius), mat)

    def add_mesh(self, verts, normals, mat):
        v = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 9)
        n = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
        assert v.shape[0] == n.shape[0]
        return self.L.orc_add_mesh(self.h, _ptr(v), _ptr(n), v.shape[0], mat)

    def add_triangle(self, v0, v1, v2, normal, mat):
        """Triangle::new + push (triangle.rs:19-23): a bare triangle as an entity of its own."""
        return self.L.orc_add_triangle(self.h, _d3(v0), _d3(v1), _d3(v2), _d3(normal), mat)

    def add_group(self, members):
        """Scene::new (scene.rs:39-47) over entities already added; returns the group's entity id."""
        m = np.ascontiguousarray(members, dtype=np.int32)
        e = self.L.orc_add_group(self.h, _ptr(m), m.size)
        if e < 0:
            raise ValueError("bad member list")
        return e

    def set_sky(self, faces):
        for i, f in enumerate(faces):
            f = np.ascontiguousarray(f, dtype=np.uint8)
            self.L.orc_set_sky_face(self.h, i, _ptr(f), f.shape[1], f.shape[0])

    def set_directional_light(self, direction, color, intensity):
        self.L.orc_set_directional_light(self.h, _d3(direction), _d3(color), float(intensity))

    def add_directional_light(self, direction, color, intensity):
        """DirectionalLight::new + world.light_sources.push (lighting.rs:25-32): appends."""
        i = self.L.orc_add_directional_light(self.h, _d3(direction), _d3(color), float(intensity))
        if i < 0:
            raise ValueError("too many lights")
        return i

    def set_ambient(self, a):
        self.L.orc_set_ambient(self.h, float(a))

    def add_point_light(self, position, color, brightness, attenuation):
        """PointLight{position, color, brightness, attenuation} (lighting.rs:52-57) pushed to world.light_sources."""
        i = self.L.orc_add_point_light(self.h, _d3(position), _d3(color), float(brightness), float(attenuation))
        if i < 0:
            raise ValueError("too many point lights")
        return i

    def set_shadow_mode(self, mode):
        self.L.orc_set_shadow_mode(self.h, int(mode))

    def translate(self, ent, vec, refit=0):
        """Entity::translate (sphere.rs:68-70, model.rs:80-86, scene.rs:68-74); refit = the extension that
        also moves the boxes the reference leaves behind."""
        if self.L.orc_translate_entity(self.h, int(ent), _d3(vec), int(refit)) != 0:
            raise ValueError("unknown entity")

    def occluded_rays(self, rays, lim2):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        lim2 = np.ascontiguousarray(np.broadcast_to(np.asarray(lim2, dtype=np.float64), (rays.shape[0],)))
        out = np.zeros(rays.shape[0], np.uint8)
        self.L.orc_occluded_rays(self.h, _ptr(rays), _ptr(lim2), rays.shape[0], _ptr(out))
        return out

    def tree(self, ent):
        nn, nr = C.c_int(), C.c_int()
        self.L.orc_mesh_tree_sizes(self.h, ent, C.byref(nn), C.byref(nr))
        nn, nr = nn.value, nr.value
        out = dict(boxes=np.zeros((nn, 6)), split=np.zeros(nn), axis=np.zeros(nn, np.int32),
                   child=np.zeros((nn, 2), np.int32), ropes=np.zeros((nn, 6), np.int32),
                   leaf_refs=np.zeros(nr, np.int32))
        self.L.orc_mesh_tree_export(self.h, ent, _ptr(out["boxes"]), _ptr(out["split"]), _ptr(out["axis"]),
                                    _ptr(out["child"]), _ptr(out["ropes"]), _ptr(out["leaf_refs"]))
        return out

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        n = rays.shape[0]
        out = dict(ent=np.zeros(n, np.int32), tri=np.zeros(n, np.int32), t=np.zeros(n),
                   pos=np.zeros((n, 3)), normal=np.zeros((n, 3)), counters=np.zeros(8, np.uint64))
        self.L.orc_trace_rays(self.h, _ptr(rays), n, _ptr(out["ent"]), _ptr(out["tri"]), _ptr(out["t"]),
                              _ptr(out["pos"]), _ptr(out["normal"]), _ptr(out["counters"]))
        return out

    def cast_rays(self, rays, max_bounce=2, prune=0):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        col = np.zeros((rays.shape[0], 3))
        self.L.orc_cast_rays(self.h, _ptr(rays), rays.shape[0], max_bounce, prune, _ptr(col))
        return col

    def render(self, width, height, spp=1, max_bounce=2, seed=0, prune_zero_mix=0, n_threads=8,
               camera=None, y0=0, y1=-1, row_stride=1, want=("rgb", "ids", "t", "color")):
        cam = camera or default_camera()
        st = Settings(width, height, spp, max_bounce, seed, prune_zero_mix, n_threads)
        npx = width * height
        out = dict(counters=np.zeros(8, np.uint64))
        out["rgb"] = np.zeros((height, width, 3), np.uint8) if "rgb" in want else None
        out["ids"] = np.full((height, width, 2), -2, np.int32) if "ids" in want else None
        out["t"] = np.zeros((height, width)) if "t" in want else None
        out["color"] = np.zeros((height, width, 3)) if "color" in want else None
        rc = self.L.orc_render(self.h, C.byref(cam), C.byref(st), y0, y1, row_stride, _ptr(out["rgb"]), _ptr(out["ids"]),
                               _ptr(out["t"]), _ptr(out["color"]), _ptr(out["counters"]))
        assert rc == 0 and npx > 0
        return out


COUNTER_NAMES = ["rays", "node_steps", "leaf_visits", "tri_tests", "sphere_tests", "box_tests", "overflow", "tri_box_hits"]
