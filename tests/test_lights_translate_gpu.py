"""PointLight with real shadow rays (the any-hit walk) and Entity::translate.

The reference ships both as dead code (PointLight has no constructor, lighting.rs:52-89; translate is
never called by render and forgets the boxes, kdtree.rs:35-42, model.rs:80-86), so there is no golden:
PARITY UNPINNED against the reference.  The oracle restates the impl blocks literally (inverted
visible() and stale boxes included) plus the two documented corrections (rrt_set_shadow_mode 1, refit 1);
the HIP path must equal it bit for bit in all four combinations."""
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


def _lit_scene(b, L, scenes, sky="sky"):
    """main.rs world + a floor sphere to catch shadows + two point lights around the UFO."""
    scenes.main_rs_scene(b, L, skybox=sky)
    matte = b.add_material(scenes.color_vec(120, 130, 110), 1.0, 1.0, 0.0, 0.2, 20)
    b.add_sphere((0.0, 1030.0, 70.0), 1000.0, matte)
    b.add_point_light((25.0, -40.0, 40.0), scenes.color_vec(255, 240, 200), 9000.0, 1.5)
    b.add_point_light((-6.0, -3.0, 2.0), scenes.color_vec(120, 160, 255), 60.0, 0.7)
    return b


def test_occluded_rays_any_hit_bit_exact(H, O, scenes):
    """rrt_occluded_rays = the any-hit walk on its own: incoherent rays against UFO + spheres, with
    limits that fall before, inside and behind the geometry, DBL_MAX, zero and tiny limits; all cull modes;
    flat world, a Scene<T> group, and a mesh too large for the LDS descent array."""
    rng = np.random.default_rng(7)
    rays = np.concatenate([_rays(rng, 150000, (0.0, 10.0, 70.0), 45.0), _rays(rng, 30000, (0.0, 0.0, 6.0), 6.0, 20.0)])
    tr = scenes.main_rs_scene(O.OracleScene(), O).trace_rays(rays)
    d2 = np.where(tr["ent"] >= 0, ((tr["pos"] - rays[:, :3]) ** 2).sum(axis=1), 1e4)
    lims = [np.full(len(rays), np.finfo(np.float64).max), d2 * rng.uniform(0.5, 1.5, len(rays)), d2, np.nextafter(d2, np.inf),
            np.zeros(len(rays)), rng.uniform(0, 3e4, len(rays))]
    for build in (scenes.main_rs_scene, scenes.main_rs_as_scene):
        c = build(O.OracleScene(), O)
        for mode in (1, 2, 0):
            g = build(H.World(device=0), H.loaders)
            g.set_cull_mode(mode)
            for lim in lims:
                a, b = g.occluded_rays(rays, lim), c.occluded_rays(rays, lim)
                assert np.array_equal(a, b), (build.__name__, mode, int((a != b).sum()))
        assert 20000 < int(c.occluded_rays(rays, lims[0]).sum()) < len(rays) - 20000
    # the large-mesh code path (descent array in global memory) on a height field
    g, c = scenes.cfg5_scene(H.World(device=0), H.loaders, n=150), scenes.cfg5_scene(O.OracleScene(), O, n=150)
    rays = _rays(rng, 100000, (0.0, 0.0, 90.0), 50.0)
    for lim in (np.finfo(np.float64).max, rng.uniform(0, 4e4, len(rays))):
        assert np.array_equal(g.occluded_rays(rays, lim), c.occluded_rays(rays, lim))


@pytest.mark.parametrize("shadow_mode", [0, 1])
def test_point_lights_with_shadow_rays_bit_exact(H, O, scenes, shadow_mode):
    g, c = _lit_scene(H.World(device=0), H.loaders, scenes), _lit_scene(O.OracleScene(), O, scenes)
    g.set_shadow_mode(shadow_mode); c.set_shadow_mode(shadow_mode)
    co = c.render(480, 270, spp=2, seed=3)
    for mode in (1, 2):
        g.set_cull_mode(mode)
        go = g.render(480, 270, spp=2, seed=3)
        _same(go, co)
        assert np.array_equal(go["color"].view(np.uint64), co["color"].view(np.uint64))
        assert go["stats"]["cull_violations"] == 0 and go["stats"]["traversal_overflow"] == 0
        assert go["stats"]["rays_traced"] == int(co["counters"][0])        # shadow rays are not cast()s
    # the lights matter, and the two visible() definitions differ
    dark = scenes.main_rs_scene(O.OracleScene(), O, skybox="sky")
    m = dark.add_material(scenes.color_vec(120, 130, 110), 1.0, 1.0, 0.0, 0.2, 20)
    dark.add_sphere((0.0, 1030.0, 70.0), 1000.0, m)
    assert not np.array_equal(dark.render(480, 270, spp=2, seed=3, want=("color",))["color"], co["color"])
    c.set_shadow_mode(1 - shadow_mode)
    assert not np.array_equal(c.render(480, 270, spp=2, seed=3, want=("color",))["color"], co["color"])
    # cast() on caller rays goes through the same shading
    rng = np.random.default_rng(5)
    rays = _rays(rng, 40000, (0.0, 10.0, 60.0), 40.0)
    c.set_shadow_mode(shadow_mode)
    assert np.array_equal(g.cast_rays(rays, 2, 0), c.cast_rays(rays, 2, 0))


def test_point_lights_in_grouped_and_large_worlds(H, O, scenes):
    """Shadow rays through a Scene<T> group (62 members, refraction materials around) and in a world with more
    than 64 entities (entity table read from global memory), corrected and literal visible()."""
    for n_spheres, grouped in ((60, True), (90, False)):
        def build(b, L):
            scenes.many_entity_scene(b, L, n_spheres=n_spheres, seed=3, grouped=grouped)
            b.add_point_light((0.0, -12.0, 20.0), (1.0, 0.9, 0.8), 400.0, 1.0)
            return b
        g, c = build(H.World(device=0), H.loaders), build(O.OracleScene(), O)
        for sm in (1, 0):
            g.set_shadow_mode(sm); c.set_shadow_mode(sm)
            go, co = g.render(320, 180, max_bounce=3), c.render(320, 180, max_bounce=3)
            _same(go, co)
            assert go["stats"]["traversal_overflow"] == 0


def test_corrected_shadows_are_shadows(H, scenes):
    """Sanity of the extension's meaning (not parity): under shadow_mode 1 a point of a ceiling above the UFO, with
    the light below the UFO, is darker than the same point without the UFO in between.  (The shadow ray travels
    towards +y: a ray that meets a kd root box on one of its max faces is a miss in the reference, aabb.rs:89-92.)"""
    def build(with_ufo):
        b = H.World(device=0)
        scenes.empty_world(b, H.loaders, "sky")
        matte = b.add_material(scenes.color_vec(200, 200, 200), 1.0, 1.0, 0.0, 0.0, 20)
        b.add_sphere((0.0, -1030.0, 70.0), 1000.0, matte)           # ceiling at y = -30 (world +y is image-down)
        if with_ufo:
            v, n = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
            b.add_mesh(v, n, matte)
        b.add_point_light((4.0, 80.0, 72.0), (1.0, 1.0, 1.0), 30000.0, 1.0)
        b.set_shadow_mode(1)
        return b
    ray = np.array([[0.0, 0.0, 0.0, 3.0, -30.0, 66.0]])
    ray[0, 3:] /= np.linalg.norm(ray[0, 3:])
    w = build(True)
    assert w.trace_rays(ray)["ent"][0] == 0                          # the eye sees the ceiling, not the UFO
    lit = build(False).cast_rays(ray, 0, 0)[0]
    shadowed = w.cast_rays(ray, 0, 0)[0]
    assert lit[0] > shadowed[0] + 0.3


@pytest.mark.parametrize("refit", [0, 1])
def test_translate_bit_exact(H, O, scenes, refit):
    """Entity::translate on the UFO, a sphere and a Scene<T> group, twice in a row (the additions accumulate
    in the vertices).  refit 0 = the reference's own (stale boxes), refit 1 = the boxes follow."""
    rng = np.random.default_rng(17)
    for build, ents in ((scenes.main_rs_scene, (0, 1)), (scenes.main_rs_as_scene, (3,))):
        g, c = build(H.World(device=0), H.loaders), build(O.OracleScene(), O)
        for step, vec in enumerate(((3.25, -1.5, 4.0), (-0.1, 0.3, 0.7))):
            for e in ents:
                g.translate(e, vec, refit); c.translate(e, vec, refit)
            for mode in (1, 2):
                g.set_cull_mode(mode)
                go, co = g.render(480, 270, spp=2, seed=step), c.render(480, 270, spp=2, seed=step)
                _same(go, co)
                assert go["stats"]["cull_violations"] == 0
            rays = _rays(rng, 100000, (2.0, 9.0, 72.0), 45.0)
            gt, ct = g.trace_rays(rays), c.trace_rays(rays)
            for k in ("ent", "tri", "t", "pos", "normal"):
                assert np.array_equal(gt[k], ct[k]), (k, refit, step)
            if build is scenes.main_rs_scene:
                tg, tc = g.tree(0), c.tree(0)
                assert all(np.array_equal(tg[k], tc[k]) for k in tg)
        assert (go["ids"][..., 0] >= 0).sum() > 1000


def test_translate_refit_moves_the_picture(H, O, scenes):
    """With refit, translating by a vector whose additions are exact equals building the scene at the new place."""
    vec = (4.0, -2.0, 8.0)          # ufo_fix vertices are f32 values: adding small integers is exact in f64
    a = scenes.main_rs_scene(H.World(device=0), H.loaders)
    a.translate(0, vec, 1)
    b = H.World(device=0)
    scenes.empty_world(b, H.loaders, "hd_blue_sunset")
    m1 = b.add_material(scenes.color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
    m2 = b.add_material(scenes.color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
    m3 = b.add_material(scenes.color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20)
    v, n = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
    b.add_mesh(v + np.array(vec), n, m3)
    b.add_sphere((-3.0, 0.0, 5.0), 1.0, m1)
    b.add_sphere((2.0, 0.0, 8.0), 1.0, m2)
    ra, rb = a.render(640, 360), b.render(640, 360)
    _same(ra, rb)
    still = scenes.main_rs_scene(H.World(device=0), H.loaders).render(640, 360, want=("ids",))
    assert not np.array_equal(still["ids"], ra["ids"])
