"""Pins the CPU oracle against everything the reference holds for this path.

The reference has no tests; its one pinned result is bruh.png, the
3840x2160 output of src/main.rs at HEAD.  The oracle must reproduce it byte for
byte -- that covers camera, cubemap, sphere, kd-tree build (including the
tie order of Rust's sort_unstable_by), rope traversal, shading and
quantisation at once.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

BRUH_SHA = "3bb433432f6e869c022d5ce00794e859bead3051959320e8061572c6056c14e8"

# SURVEY §4 known-answer pixels of bruh.png: (x, y) -> RGB
KAT_PIXELS = {
    (0, 0): (76, 135, 203), (3839, 2159): (34, 61, 112), (840, 1080): (26, 26, 51),
    (723, 718): (16, 16, 31), (1045, 835): (15, 16, 31), (1920, 1080): (93, 93, 59),
    (1920, 1300): (71, 71, 37), (2000, 1500): (86, 86, 47), (2380, 1080): (62, 92, 146),
}


@pytest.fixture(scope="module")
def bruh(golden_dir):
    from PIL import Image
    g = np.asarray(Image.open(os.path.join(golden_dir, "bruh.png")).convert("RGB"))
    assert g.shape == (2160, 3840, 3)
    assert hashlib.sha256(g.tobytes()).hexdigest() == BRUH_SHA
    return g


@pytest.fixture(scope="module")
def main_oracle(O, scenes):
    return scenes.main_rs_scene(O.OracleScene(), O)


def test_oracle_reproduces_bruh_png(main_oracle, bruh):
    out = main_oracle.render(3840, 2160, n_threads=8, want=("rgb", "ids"))
    assert out["counters"][6] == 0          # no traversal overflow
    bad = np.any(out["rgb"] != bruh, axis=2)
    assert bad.sum() == 0, "%d pixels differ from bruh.png" % bad.sum()
    assert hashlib.sha256(out["rgb"].tobytes()).hexdigest() == BRUH_SHA
    ents = out["ids"][..., 0]
    # all three entities and the sky are on screen, so every code path is pinned
    assert (ents == 0).sum() > 800000 and (ents == 1).sum() > 400000 and (ents == 2).sum() > 100000
    for (x, y), rgb in KAT_PIXELS.items():
        assert tuple(out["rgb"][y, x]) == rgb


def test_plain_tie_order_does_not_reproduce_bruh(O, scenes, bruh):
    """Documents why the pdqsort restatement exists: with the plain
    (key, index) order the kd-tree differs and UFO pixels change."""
    O.lib().orc_set_sort_mode(1)
    try:
        s = scenes.main_rs_scene(O.OracleScene(), O)
        out = s.render(3840, 2160, n_threads=8, y0=900, y1=1300, want=("rgb",))
    finally:
        O.lib().orc_set_sort_mode(0)
    bad = np.any(out["rgb"][900:1300] != bruh[900:1300], axis=2)
    assert bad.sum() > 1000


def test_philox_known_answers(O):
    """Random123 known-answer vectors for Philox4x32-10."""
    kats = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kats:
        c = np.array(ctr, np.uint32)
        k = np.array(key, np.uint32)
        o = np.zeros(4, np.uint32)
        O.lib().orc_philox(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        assert tuple(int(x) for x in o) == want


def test_powi_matches_compiler_rt_order(O):
    # alpha = 20 = 0b10100 : r = a^4 * a^16, squarings a->a^2->a^4->a^8->a^16
    for a in (0.3, 0.9371, 1.0000001, 0.0, 1e-200):
        a2 = a * a; a4 = a2 * a2; a8 = a4 * a4; a16 = a8 * a8
        assert O.lib().orc_powi(a, 20) == a4 * a16
    assert O.lib().orc_powi(3.0, 0) == 1.0
    assert O.lib().orc_powi(2.0, -2) == 0.25


def _aabb(O, box, o, d):
    pos = np.zeros(3)
    b, o, d = np.array(box, float), np.array(o, float), np.array(d, float)   # keep the buffers alive over the call
    hit = O.lib().orc_aabb_collide(b.ctypes.data, o.ctypes.data, d.ctypes.data, pos.ctypes.data)
    return hit, pos


def test_aabb_collide_quirks(O):
    box = (-1, -1, -1, 1, 1, 1)
    # from outside, head on: entry point on the near face
    hit, p = _aabb(O, box, (0, 0, -5), (0, 0, 1))
    # x,y axes are inside their slabs while the origin is outside the box: dist=-dir -> -0 -> time -1 (aabb.rs:69-80)
    assert hit == 1 and tuple(p) == (0, 0, -1)
    # pointing away: negative with position (0,0,0) (common.rs:57-72)
    hit, p = _aabb(O, box, (0, 0, -5), (0, 0, -1))
    assert hit == 0 and tuple(p) == (0, 0, 0)
    # from inside: exit point
    hit, p = _aabb(O, box, (0, 0, 0), (1, 0, 0))
    assert hit == 1 and tuple(p) == (1, 0, 0)
    # outside entry that grazes past the +0.001 containment nudge (aabb.rs:89)
    hit, p = _aabb(O, box, (0.9995, 0, -5), (0, 0, 1))
    assert hit == 0
    # diagonal from outside
    d = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    hit, p = _aabb(O, box, (-3, -3, -3), d)
    assert hit == 1 and np.allclose(p, (-1, -1, -1))


def test_triangle_collide(O):
    v = np.array([0, 0, 5, 1, 0, 5, 0, 1, 5], float)
    pos = np.zeros(3); t = C.c_double()
    L = O.lib()
    o = np.array([0.25, 0.25, 0.0]); d = np.array([0.0, 0.0, 1.0])
    assert L.orc_triangle_collide(v.ctypes.data, o.ctypes.data, d.ctypes.data, pos.ctypes.data, C.byref(t)) == 1
    assert t.value == 5.0 and tuple(pos) == (0.25, 0.25, 5.0)
    # double sided (triangle.rs:34 tests |a| only)
    o2 = np.array([0.25, 0.25, 10.0]); d2 = np.array([0.0, 0.0, -1.0])
    assert L.orc_triangle_collide(v.ctypes.data, o2.ctypes.data, d2.ctypes.data, pos.ctypes.data, C.byref(t)) == 1
    # behind the origin: t <= EPS rejected
    assert L.orc_triangle_collide(v.ctypes.data, o2.ctypes.data, d.ctypes.data, pos.ctypes.data, C.byref(t)) == 0
    # u + v > 1
    o3 = np.array([0.75, 0.75, 0.0])
    assert L.orc_triangle_collide(v.ctypes.data, o3.ctypes.data, d.ctypes.data, pos.ctypes.data, C.byref(t)) == 0
    # parallel
    d4 = np.array([1.0, 0.0, 0.0])
    assert L.orc_triangle_collide(v.ctypes.data, o.ctypes.data, d4.ctypes.data, pos.ctypes.data, C.byref(t)) == 0


def test_sphere_returns_near_root_even_if_negative(O):
    """sphere.rs:28-47 rejects only tca<0 or d2>r2."""
    L = O.lib()
    c = np.array([0.0, 0.0, 0.5]); pos = np.zeros(3); n = np.zeros(3); t = C.c_double()
    o = np.array([0.0, 0.0, 0.0]); d = np.array([0.0, 0.0, 1.0])
    # origin inside the sphere: near root is behind the origin, still a "hit"
    assert L.orc_sphere_collide(c.ctypes.data, 1.0, o.ctypes.data, d.ctypes.data, pos.ctypes.data, n.ctypes.data, C.byref(t)) == 1
    assert t.value == -0.5 and tuple(pos) == (0, 0, -0.5) and tuple(n) == (0, 0, -1)
    # centre behind the origin
    c2 = np.array([0.0, 0.0, -3.0])
    assert L.orc_sphere_collide(c2.ctypes.data, 1.0, o.ctypes.data, d.ctypes.data, pos.ctypes.data, n.ctypes.data, C.byref(t)) == 0


def test_cubemap_face_table(O):
    """cubemap.rs:99-157: 0=+X 1=-X 2=+Y 3=-Y 4=+Z 5=-Z, later ifs win ties."""
    L = O.lib()
    def cm(x, y, z):
        i = C.c_uint32(); u = C.c_double(); v = C.c_double()
        L.orc_cubemap(x, y, z, C.byref(i), C.byref(u), C.byref(v))
        return i.value, u.value, v.value
    assert cm(1, 0, 0) == (0, 0.5, 0.5)
    assert cm(-1, 0, 0)[0] == 1 and cm(0, 1, 0)[0] == 2 and cm(0, -1, 0)[0] == 3
    assert cm(0, 0, 1)[0] == 4 and cm(0, 0, -1)[0] == 5
    assert cm(1, 1, 1)[0] == 4            # three-way tie: the +Z test runs last among the positives
    assert cm(1, 1, 0)[0] == 2
    assert cm(0.2, -0.1, 1.0) == (4, 0.5 * (0.2 / 1.0 + 1.0), 0.5 * (-0.1 / 1.0 + 1.0))
    i, u, v = cm(0, 0, 0)                 # not-positive on every axis -> -X,-Y,-Z all match, -Z last; 0/0 = NaN
    assert i == 5 and np.isnan(u) and np.isnan(v)


def test_model_new_quirks(O, scenes):
    """model.rs:31 drops the last face; positions f32->f64 then T*S; normals untransformed."""
    v, n = O.load_model(os.path.join(scenes.ASSETS, "obj", "cube.obj"), (0, 0, 12), (2, 2, 2))
    assert v.shape == (11, 3, 3) and n.shape == (11, 3)
    assert set(np.unique(v[..., 2])) == {10.0, 14.0}
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0)
    v2, n2 = O.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0, 30, 70), (1, -1, 1))
    assert v2.shape[0] == 1705
    # bbox after the main.rs transform
    assert np.allclose(v2.reshape(-1, 3).min(0), (-33.92, -4.82, 36.04), atol=0.01)
    assert np.allclose(v2.reshape(-1, 3).max(0), (33.95, 22.21, 103.91), atol=0.01)


def test_spp1_is_reference_and_jitter_stays_in_pixel(O):
    cam = O.default_camera()
    st = O.Settings(640, 480, 4, 2, 5, 0, 1)
    r = np.zeros(6)
    O.lib().orc_primary_ray(C.byref(cam), C.byref(st), 10, 20, 0, r.ctypes.data)
    lp = np.array([-80 + (10 / 640) * 160.0, -45 + (20 / 480) * 90.0, 75.0])
    want = lp * (1.0 / np.sqrt((lp[0] * lp[0] + lp[1] * lp[1]) + lp[2] * lp[2]))
    assert tuple(r[:3]) == (0, 0, 0) and tuple(r[3:]) == tuple(want)
    for s in (1, 2, 3):
        O.lib().orc_primary_ray(C.byref(cam), C.byref(st), 10, 20, s, r.ctypes.data)
        p = r[3:] * (75.0 / r[5])
        fx = (p[0] + 80) / 160 * 640; fy = (p[1] + 45) / 90 * 480
        assert 10 <= fx < 11 and 20 <= fy < 21


def test_point_light_known_answers(O, scenes):
    """PointLight (lighting.rs:52-89) has no golden in the reference (dead code): the oracle's restatement is
    pinned to the impl block's own formulas evaluated by hand in numpy -- illuminate's power and direction,
    Lambert's and Phong's sums (lambert.rs:38-41, phong.rs:41-43) -- and to the two visible() definitions:
    as written (visible as soon as one entity is missed) and corrected (blocked by the sphere in between)."""
    import numpy as np
    from oracle import oracle as Om

    def build(blocker):
        c = Om.OracleScene()
        c.set_sky(Om.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", "sky")))
        c.set_ambient(0.15)                                   # no directional light: only the point light shines
        m = c.add_material((0.5, 0.25, 0.125), 0.8, 1.0, 0.0, 0.5, 20)
        c.add_sphere((0.0, 0.0, 10.0), 2.0, m)
        if blocker:
            c.add_sphere((0.0, -2.5, 5.5), 1.0, m)
        c.add_point_light((0.0, -5.0, 3.0), (1.0, 0.5, 0.25), 300.0, 2.0)
        return c
    ray = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    pos, n, d = np.array([0.0, 0.0, 8.0]), np.array([0.0, 0.0, -1.0]), np.array([0.0, 0.0, 1.0])
    color, lcol, L = np.array([0.5, 0.25, 0.125]), np.array([1.0, 0.5, 0.25]), np.array([0.0, -5.0, 3.0])
    dot = lambda a, b: (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
    norm = lambda v: v * (1.0 / np.sqrt(dot(v, v)))
    dir0 = pos - L
    power = 300.0 / (2.0 * dot(dir0, dir0))
    dirn = norm(dir0)
    lam = color * max(power * (0.8 / np.pi) * -dot(n, dirn), 0.0)
    bis = norm(-dirn - d)
    x = max(dot(bis, n), 0.0)
    x2 = x * x; x4 = x2 * x2; x8 = x4 * x4; x16 = x8 * x8
    ph = lcol * (power * (x4 * x16))
    lit = ((color * 0.15 + (np.zeros(3) + lam) * 1.0) + (np.zeros(3) + ph) * 0.5)
    unlit = ((color * 0.15 + np.zeros(3) * 1.0) + np.zeros(3) * 0.5)
    for mode in (0, 1):
        c = build(False)
        c.set_shadow_mode(mode)
        assert np.array_equal(c.cast_rays(ray, 0, 1)[0], lit), mode          # nothing in the way: both definitions say lit
    c = build(True)
    c.set_shadow_mode(1)
    assert np.array_equal(c.cast_rays(ray, 0, 1)[0], unlit)                  # corrected: the blocker casts a shadow
    c.set_shadow_mode(0)
    assert np.array_equal(c.cast_rays(ray, 0, 1)[0], lit)                    # as written: the first entity is missed -> "visible"
    # as written, a world whose every entity is hit by the (un-normalised) shadow ray is dark
    c = Om.OracleScene()
    c.set_sky(Om.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", "sky")))
    m = c.add_material((0.5, 0.25, 0.125), 0.8, 1.0, 0.0, 0.5, 20)
    c.add_sphere((0.0, 0.0, 10.0), 2.0, m)
    c.add_point_light((0.0, 0.0, 20.0), (1.0, 1.0, 1.0), 300.0, 2.0)         # behind the sphere: the ray from pos runs through it
    assert np.array_equal(c.cast_rays(ray, 0, 1)[0], unlit)
    # any-hit query: limits before / behind the hit
    c = build(False)
    assert list(c.occluded_rays(np.repeat(ray, 3, axis=0), [63.9, 64.0, 64.1])) == [0, 0, 1]


def test_light_list_and_bare_triangle_known_answers(O, scenes):
    """world.light_sources is summed in push order (lambert.rs:34-43, phong.rs:37-45), and a bare Triangle entity
    answers with Triangle::collide and its STORED normal (triangle.rs:25-58).  Neither has a golden in the reference
    (main.rs holds one light and no bare triangle), so the oracle's restatement is pinned to the formulas evaluated by
    hand in numpy: two directional lights with a point light between them shining on an axis-aligned triangle."""
    import numpy as np
    from oracle import oracle as Om
    dot = lambda a, b: (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]
    norm = lambda v: v * (1.0 / np.sqrt(dot(v, v)))
    color, albedo, mixL, mixP, alpha = np.array([0.5, 0.25, 0.125]), 0.8, 0.9, 0.5, 20
    lights = [("d", np.array([1.0, -1.0, 1.0]), np.array([0.9, 0.9, 0.9]), 2.0, 0.0),
              ("p", np.array([3.0, -5.0, 3.0]), np.array([1.0, 0.5, 0.25]), 300.0, 2.0),
              ("d", np.array([-0.3, 0.2, 1.0]), np.array([0.2, 0.4, 1.0]), 0.7, 0.0)]

    def build(order):
        c = Om.OracleScene()
        c.set_sky(Om.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", "sky")))
        m = c.add_material(color, albedo, mixL, 0.0, mixP, alpha)
        # axis-aligned (z = 10), stored normal deliberately not unit length: returned as stored
        e = c.add_triangle((-4.0, -4.0, 10.0), (4.0, -4.0, 10.0), (0.0, 6.0, 10.0), (0.0, 0.0, -1.25), m)
        assert e == 0
        for k in order:
            kind, v, col, a, b = lights[k]
            if kind == "d":
                c.add_directional_light(v, col, a)
            else:
                c.add_point_light(v, col, a, b)
        c.set_shadow_mode(1)
        return c

    ray = np.array([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    pos, n, d = np.array([0.0, 0.0, 10.0]), np.array([0.0, 0.0, -1.25]), np.array([0.0, 0.0, 1.0])

    def expect(order):
        lam, ph = np.zeros(3), np.zeros(3)
        for k in order:
            kind, v, col, a, b = lights[k]
            if kind == "d":
                dirn, power = norm(v), a
                if not dot(n, dirn) < 0.0:
                    continue
            else:
                dir0 = pos - v
                assert dot(n, dir0) < 0.0                    # faces the light, nothing in between: visible
                dirn, power = norm(dir0), a / (b * dot(dir0, dir0))
            lam = lam + color * max(power * (albedo / np.pi) * -dot(n, dirn), 0.0)
            x = max(dot(norm(-dirn - d), n), 0.0)
            x2 = x * x; x4 = x2 * x2; x8 = x4 * x4; x16 = x8 * x8
            ph = ph + col * (power * (x4 * x16))
        return (color * 0.15 + lam * mixL) + ph * mixP

    tr = build((0,)).trace_rays(ray)
    assert tr["ent"][0] == 0 and tr["tri"][0] == -1 and tr["t"][0] == 10.0
    assert np.array_equal(tr["normal"][0], n) and np.array_equal(tr["pos"][0], pos)
    results = {}
    for order in ((0,), (0, 1), (0, 1, 2), (2, 1, 0), (1, 2, 0)):
        got = build(order).cast_rays(ray, 0, 1)[0]
        assert np.array_equal(got, expect(order)), order
        results[order] = got
    assert not np.array_equal(results[(0, 1, 2)], results[(0, 1)])
    # the triangle pushed as a one-triangle Model instead is never seen: its kd root box is flat (aabb.rs:89)
    c = Om.OracleScene()
    c.set_sky(Om.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", "sky")))
    m = c.add_material(color, albedo, mixL, 0.0, mixP, alpha)
    c.add_mesh(np.array([[[-4.0, -4.0, 10.0], [4.0, -4.0, 10.0], [0.0, 6.0, 10.0]]]), np.array([[0.0, 0.0, -1.0]]), m)
    assert c.trace_rays(ray)["ent"][0] == -1
