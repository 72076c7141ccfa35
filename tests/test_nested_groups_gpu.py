"""Scene<Scene<T>> (scene.rs:32 is generic over any Entity, a Scene included): a group whose members are groups.
The reference never builds one (main.rs:57-59, commented out, is a flat Scene), so there is no golden: PARITY UNPINNED
against the reference; the oracle restates Scene::collide recursively (a member scene's collide is its own tree walk,
its bounding box its own tree's root box) and the HIP path must equal it bit for bit."""
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


def _nested(b, L, scenes, n_spheres=14, seed=5, with_light=False):
    """main.rs world; the UFO and a ring of spheres in one inner scene, the two main.rs spheres, a bare triangle and more
    spheres in a second, both (and one loose sphere) under an outer scene; one sphere stays top-level."""
    scenes.main_rs_scene(b, L, skybox="sky")                       # entities 0 (ufo), 1, 2 (spheres)
    rng = np.random.default_rng(seed)
    mats = [b.add_material(scenes.color_vec(200, 80, 60), 0.9, 1.0, 0.0, 0.2, 20),
            b.add_material(scenes.color_vec(40, 40, 40), 0.8, 0.2, 0.8, 0.3, 30)]
    ring, loose = [], []
    for i in range(n_spheres):
        c = (float(rng.uniform(-30, 30)), float(rng.uniform(-5, 40)), float(rng.uniform(40, 100)))
        e = b.add_sphere(c, float(rng.uniform(1.0, 4.0)), mats[i % 2])
        (ring if i % 2 == 0 else loose).append(e)
    tri = b.add_triangle((-12.0, 4.0, 30.0), (12.0, 4.0, 30.0), (0.0, -14.0, 34.0), (0.0, 0.0, -1.0), mats[0])
    inner_a = b.add_group([0] + ring)
    inner_b = b.add_group([1, 2, tri] + loose[:-2])
    outer = b.add_group([inner_a, inner_b, loose[-2]])
    if with_light:
        b.add_point_light((0.0, -30.0, 40.0), (1.0, 0.95, 0.8), 2500.0, 1.2)
    return b, outer, inner_a, loose[-1]


def test_nested_scene_bit_exact(H, O, scenes):
    g, outer, _, _ = _nested(H.World(device=0), H.loaders, scenes)
    c, _, _, _ = _nested(O.OracleScene(), O, scenes)
    co = c.render(480, 270, spp=2, seed=1)
    assert len(np.unique(co["ids"][..., 0])) > 8           # many different members are on screen
    for mode in (1, 2, 0):
        g.set_cull_mode(mode)
        go = g.render(480, 270, spp=2, seed=1)
        _same(go, co)
        assert go["stats"]["cull_violations"] == 0 and go["stats"]["traversal_overflow"] == 0
        assert go["stats"]["rays_traced"] == int(co["counters"][0])
    g.set_cull_mode(1)
    rng = np.random.default_rng(3)
    rays = np.concatenate([_rays(rng, 120000, (0.0, 15.0, 70.0), 45.0), _rays(rng, 30000, (0.0, 0.0, 6.0), 6.0, 20.0)])
    gt, ct = g.trace_rays(rays), c.trace_rays(rays)
    for k in ("ent", "tri", "t", "pos", "normal"):
        assert np.array_equal(gt[k], ct[k]), k
    assert np.array_equal(g.cast_rays(rays[:40000], 2, 0), c.cast_rays(rays[:40000], 2, 0))
    lim = rng.uniform(0, 3e4, len(rays))
    assert np.array_equal(g.occluded_rays(rays, lim), c.occluded_rays(rays, lim))
    # the trees: outer over the inner scenes' root boxes
    tg, tc = g.tree(outer), c.tree(outer)
    assert all(np.array_equal(tg[k], tc[k]) for k in tg)


def test_nested_scene_with_more_than_64_entities(H, O, scenes):
    """Entity table read from global memory (more than 64 entities): the per-lane member path of the nested walk."""
    g, _, _, _ = _nested(H.World(device=0), H.loaders, scenes, n_spheres=80, seed=9)
    c, _, _, _ = _nested(O.OracleScene(), O, scenes, n_spheres=80, seed=9)
    for mode in (1, 2):
        g.set_cull_mode(mode)
        go, co = g.render(320, 180, spp=2, max_bounce=3), c.render(320, 180, spp=2, max_bounce=3)
        _same(go, co)
        assert go["stats"]["cull_violations"] == 0 and go["stats"]["traversal_overflow"] == 0
    rng = np.random.default_rng(4)
    rays = _rays(rng, 60000, (0.0, 15.0, 70.0), 45.0)
    gt, ct = g.trace_rays(rays), c.trace_rays(rays)
    for k in ("ent", "tri", "t"):
        assert np.array_equal(gt[k], ct[k]), k


def test_nested_scene_is_not_the_flat_scene(O, scenes):
    """Sanity of the restatement: nesting changes which leaf answers first, so some pixels differ from the same
    entities in one flat Scene<T> -- otherwise the test above would not exercise the nested walk's semantics."""
    c, _, _, _ = _nested(O.OracleScene(), O, scenes)
    f = O.OracleScene()
    scenes.main_rs_scene(f, O, skybox="sky")
    rng = np.random.default_rng(5)
    mats = [f.add_material(scenes.color_vec(200, 80, 60), 0.9, 1.0, 0.0, 0.2, 20),
            f.add_material(scenes.color_vec(40, 40, 40), 0.8, 0.2, 0.8, 0.3, 30)]
    ids = []
    for i in range(14):
        cc = (float(rng.uniform(-30, 30)), float(rng.uniform(-5, 40)), float(rng.uniform(40, 100)))
        ids.append(f.add_sphere(cc, float(rng.uniform(1.0, 4.0)), mats[i % 2]))
    tri = f.add_triangle((-12.0, 4.0, 30.0), (12.0, 4.0, 30.0), (0.0, -14.0, 34.0), (0.0, 0.0, -1.0), mats[0])
    loose = [e for i, e in enumerate(ids) if i % 2 == 1]
    f.add_group([0, 1, 2, tri] + [e for e in ids if e != loose[-1]])
    a = c.render(320, 180, want=("ids",))["ids"]
    b = f.render(320, 180, want=("ids",))["ids"]
    assert (a[..., 0] >= 0).sum() > 2000 and (b[..., 0] >= 0).sum() > 2000


@pytest.mark.parametrize("refit", [0, 1])
def test_nested_scene_translate_and_shadows(H, O, scenes, refit):
    g, outer, inner_a, _ = _nested(H.World(device=0), H.loaders, scenes, with_light=True)
    c, _, _, _ = _nested(O.OracleScene(), O, scenes, with_light=True)
    for sm in (1, 0):
        g.set_shadow_mode(sm); c.set_shadow_mode(sm)
        _same(g.render(320, 180, spp=2), c.render(320, 180, spp=2))
    for step, (ent, vec) in enumerate(((outer, (2.5, -1.0, 3.0)), (inner_a, (-1.25, 0.5, 0.75)))):
        g.translate(ent, vec, refit); c.translate(ent, vec, refit)
        for mode in (1, 2):
            g.set_cull_mode(mode)
            go, co = g.render(320, 180, spp=2, seed=step), c.render(320, 180, spp=2, seed=step)
            _same(go, co)
            assert go["stats"]["cull_violations"] == 0


def test_nesting_limits(H, scenes):
    b = H.World(device=0)
    scenes.empty_world(b, H.loaders, "sky")
    m = b.add_material(scenes.color_vec(200, 80, 60), 0.9, 1.0, 0.0, 0.2, 20)
    s = [b.add_sphere((float(i), 0.0, 20.0), 0.4, m) for i in range(6)]
    g1 = b.add_group(s[:2])
    g2 = b.add_group([g1, s[2]])
    with pytest.raises(Exception):
        b.add_group([g2, s[3]])            # two levels of nesting: refused
    with pytest.raises(Exception):
        b.add_group([g1, s[4]])            # g1 is a member of g2 already
    b.add_group([s[3], s[4]])              # the world is still usable
    out = b.render(64, 36)
    assert out["rgb"].shape == (36, 64, 3)
