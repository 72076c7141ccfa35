"""world.light_sources as the ordered list it is (common.rs:12; summed in push order by lambert.rs:34-43 and
phong.rs:37-45) and a bare Triangle as a world entity (triangle.rs:19-58).

The reference's own main.rs holds one DirectionalLight and no bare triangle, so neither has a golden: PARITY
UNPINNED against the reference; the oracle restates the loops literally (tests/test_oracle_golden.py pins it
with numpy known answers) and the HIP path must equal it bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(g, c, keys=("ids", "t", "color", "rgb")):
    for k in keys:
        assert np.array_equal(g[k], c[k]), "%s differs in %d places" % (k, int((g[k] != c[k]).sum()))


def _rays(rng, n, target, spread, origin_box=120.0):
    o = rng.uniform(-origin_box, origin_box, (n, 3))
    d = np.array(target) + rng.uniform(-spread, spread, (n, 3)) - o
    d = d * (1.0 / np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]))[:, None]
    return np.concatenate([o, d], axis=1)


def _many_lights(b, L, scenes, order):
    """main.rs world (its DirectionalLight is light 0) + a floor; then `order` pushes more lights."""
    scenes.main_rs_scene(b, L, skybox="sky")
    matte = b.add_material(scenes.color_vec(120, 130, 110), 1.0, 1.0, 0.0, 0.2, 20)
    b.add_sphere((0.0, 1030.0, 70.0), 1000.0, matte)
    for kind in order:
        if kind == "d2":
            b.add_directional_light((-0.3, -1.0, 0.45), scenes.color_vec(255, 180, 120), 0.7)
        elif kind == "d3":
            b.add_directional_light((0.2, 0.1, 1.0), scenes.color_vec(90, 120, 255), 1.3)
        else:
            b.add_point_light((25.0, -40.0, 40.0), scenes.color_vec(255, 240, 200), 9000.0, 1.5)
    return b


@pytest.mark.parametrize("order", [("d2",), ("p", "d2"), ("d2", "p", "d3")])
def test_light_list_in_push_order_bit_exact(H, O, scenes, order):
    """Two and three directional lights, a point light pushed between them: every frame equals the oracle's bit for
    bit, and the ORDER matters (the same lights pushed in another order give another f64 image)."""
    g, c = _many_lights(H.World(device=0), H.loaders, scenes, order), _many_lights(O.OracleScene(), O, scenes, order)
    g.set_shadow_mode(1); c.set_shadow_mode(1)
    co = c.render(480, 270, spp=2, seed=3)
    for mode in (1, 2):
        g.set_cull_mode(mode)
        go = g.render(480, 270, spp=2, seed=3)
        _same(go, co)
        assert go["stats"]["cull_violations"] == 0 and go["stats"]["rays_traced"] == int(co["counters"][0])
    rays = _rays(np.random.default_rng(5), 30000, (0.0, 10.0, 60.0), 40.0)
    assert np.array_equal(g.cast_rays(rays, 2, 0), c.cast_rays(rays, 2, 0))
    if len(order) > 1:
        rev = _many_lights(O.OracleScene(), O, scenes, tuple(reversed(order)))
        rev.set_shadow_mode(1)
        other = rev.render(480, 270, spp=2, seed=3, want=("color",))["color"]
        assert not np.array_equal(other, co["color"])                       # a sum's order is part of an f64 result
        assert np.abs(other - co["color"]).max() < 1e-9                     # ... and only that


def test_set_directional_light_replaces_the_list(H, O, scenes):
    """rrt_set_directional_light = light_sources becomes [this light] (new_empty_world, tracer.rs:55-65): lights
    pushed before are gone, and the frame equals the plain main.rs frame."""
    g = scenes.main_rs_scene(H.World(device=0), H.loaders, skybox="sky")
    g.add_directional_light((0.0, -1.0, 0.0), (1.0, 1.0, 1.0), 3.0)
    g.add_point_light((0.0, -12.0, 20.0), (1.0, 0.9, 0.8), 400.0, 1.0)
    lit = g.render(320, 180, want=("color",))["color"]
    g.set_directional_light((1.0, -1.0, 1.0), scenes.color_vec(230, 230, 230), 2.0)
    plain = scenes.main_rs_scene(O.OracleScene(), O, skybox="sky").render(320, 180)
    go = g.render(320, 180)
    _same(go, plain)
    assert not np.array_equal(lit, go["color"])
    # a world without any light: ambient only
    dark_g, dark_c = H.World(device=0), O.OracleScene()
    for b, L in ((dark_g, H.loaders), (dark_c, O)):
        b.set_sky(L.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", "sky")))
        b.add_sphere((0.0, 0.0, 8.0), 2.0, b.add_material((0.8, 0.4, 0.2), 1.0, 1.0, 0.2, 0.3, 20))
    _same(dark_g.render(160, 90), dark_c.render(160, 90))


def _tri_world(b, L, scenes, grouped=False):
    """Bare triangles: an axis-aligned one (z = 20: its box is flat), a tilted mirror, one behind a sphere, one whose
    stored normal is neither unit nor geometric (triangle.rs:54 returns it as stored) -- next to a sphere."""
    scenes.empty_world(b, L, "sky")
    red = b.add_material(scenes.color_vec(200, 60, 40), 1.0, 1.0, 0.0, 0.3, 20)
    mirror = b.add_material(scenes.color_vec(10, 10, 10), 0.8, 0.1, 0.9, 0.2, 20)
    blue = b.add_material(scenes.color_vec(40, 80, 220), 0.9, 1.0, 0.05, 0.2, 20)
    ids = [
        b.add_triangle((-12.0, -7.0, 20.0), (12.0, -7.0, 20.0), (0.0, 9.0, 20.0), (0.0, 0.0, -1.0), red),
        b.add_triangle((6.0, -9.0, 14.0), (16.0, -4.0, 19.0), (8.0, 6.0, 12.0), (-0.6, 0.1, -0.79), mirror),
        b.add_sphere((-4.0, 1.0, 12.0), 2.5, blue),
        b.add_triangle((-9.0, -3.0, 16.0), (1.0, -3.0, 15.0), (-4.0, 5.0, 17.0), (0.0, 0.0, -2.5), blue),
        b.add_triangle((-15.0, 6.0, 9.0), (-6.0, 8.0, 11.0), (-11.0, 2.0, 30.0), (0.3, -1.0, -0.2), red),
    ]
    if grouped:
        b.add_group(ids)
    return b


@pytest.mark.parametrize("grouped", [False, True])
def test_bare_triangle_entities_bit_exact(H, O, scenes, grouped):
    g, c = _tri_world(H.World(device=0), H.loaders, scenes, grouped), _tri_world(O.OracleScene(), O, scenes, grouped)
    co = c.render(480, 270, spp=2, seed=11)
    for mode in (1, 2, 0):
        g.set_cull_mode(mode)
        go = g.render(480, 270, spp=2, seed=11)
        _same(go, co)
        assert go["stats"]["cull_violations"] == 0
    hit = go["ids"][..., 0]
    assert set(np.unique(hit)) >= {-1, 0, 1, 2, 3, 4}
    assert (go["ids"][..., 1] == -1).all()                 # no triangle id: they are not part of a mesh
    # a frame large enough for the tile bins (>= 2^20 path samples): the bins must never hide a triangle
    co = c.render(960, 540, spp=4, seed=2)
    g.set_cull_mode(2)
    go = g.render(960, 540, spp=4, seed=2)
    _same(go, co)
    assert go["stats"]["cull_violations"] == 0
    g.set_cull_mode(1)
    _same(g.render(960, 540, spp=4, seed=2), co)
    rays = _rays(np.random.default_rng(3), 60000, (0.0, 0.0, 16.0), 14.0, 40.0)
    gt, ct = g.trace_rays(rays), c.trace_rays(rays)
    for k in ("ent", "tri", "t", "pos", "normal"):
        assert np.array_equal(gt[k], ct[k]), k
    assert np.array_equal(g.cast_rays(rays, 2, 0), c.cast_rays(rays, 2, 0))
    lim = np.random.default_rng(4).uniform(0, 900.0, len(rays))
    assert np.array_equal(g.occluded_rays(rays, lim), c.occluded_rays(rays, lim))
    # Triangle::translate (triangle.rs:79-83)
    for b in (g, c):
        b.translate(0, (1.5, -0.25, 2.0))
        b.translate(5 if grouped else 3, (0.5, 0.5, -1.0))
    _same(g.render(480, 270), c.render(480, 270))


def test_axis_aligned_triangle_needs_the_bare_entity(H, O, scenes):
    """What rrt_add_mesh(n = 1) cannot express: the kd root box of an axis-aligned triangle is flat, and
    AABB::collide's +0.001 nudge (aabb.rs:89) then rejects every ray, so Model never reaches Triangle::collide;
    a Triangle pushed as an entity is collided directly (triangle.rs:25-58) and is seen."""
    v = np.array([[[-12.0, -7.0, 20.0], [12.0, -7.0, 20.0], [0.0, 9.0, 20.0]]])
    n = np.array([[0.0, 0.0, -1.0]])
    for bare in (False, True):
        g, c = H.World(device=0), O.OracleScene()
        for b, L in ((g, H.loaders), (c, O)):
            scenes.empty_world(b, L, "sky")
            m = b.add_material(scenes.color_vec(200, 60, 40), 1.0, 1.0, 0.0, 0.3, 20)
            if bare:
                b.add_triangle(v[0, 0], v[0, 1], v[0, 2], n[0], m)
            else:
                b.add_mesh(v, n, m)
        go, co = g.render(320, 180), c.render(320, 180)
        _same(go, co)
        seen = int((go["ids"][..., 0] == 0).sum())
        assert (seen > 2000) if bare else (seen == 0)
