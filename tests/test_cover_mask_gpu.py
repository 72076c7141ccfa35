"""Primary-ray coverage masks (include/rrt.h: rrt_cover_mask) must never change a result: frames rendered
with the masks (cull mode 1), with the masks verified against the walk (mode 2: a masked ray that the walk
answers counts as a cull violation) and without them (mode 0 = every test as written) are compared with each
other and with the oracle, for cameras chosen to sit on the mask's own edge cases -- eye near, inside and
past the mesh (triangles at and behind the eye plane), a long lens on the silhouette (pixel-sized triangles
become many pixels wide), image widths that are not a multiple of 32, sub-pixel jitter, tiles of several
ranks -- and the mask itself is checked against the oracle's primary hits."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cam(H, eye=(0.0, 0.0, 0.0), lens_z=75.0, lens_w=160.0, lens_h=90.0):
    c = H.default_camera()
    for a in range(3):
        c.eye[a] = eye[a]
    c.lens_z, c.lens_w, c.lens_h = lens_z, lens_w, lens_h
    return c


def _ocam(O, cam):
    c = O.default_camera()
    for a in range(3):
        c.eye[a] = cam.eye[a]
    c.lens_z, c.lens_w, c.lens_h = cam.lens_z, cam.lens_w, cam.lens_h
    return c


def _same(g, c, keys=("ids", "t", "color", "rgb")):
    for k in keys:
        assert np.array_equal(g[k], c[k]), "%s differs in %d places" % (k, int((g[k] != c[k]).sum()))


CAMERAS = [
    dict(),                                                          # main.rs
    dict(eye=(0.0, 22.0, 52.0)),                                     # close below the UFO: triangles tens of pixels wide
    dict(eye=(0.0, 30.0, 70.0)),                                     # inside the mesh: triangles behind the eye plane
    dict(eye=(1.5, 30.25, 64.0)),                                    # inside its root box, off centre
    dict(eye=(0.0, 30.0, 120.0)),                                    # past it, looking away: everything behind
    dict(eye=(-40.0, 10.0, 20.0), lens_z=30.0),                      # off axis, wide lens: the UFO near the image edge
    dict(lens_z=75.0, lens_w=16.0, lens_h=9.0, eye=(-4.0, 26.0, 0.0)),   # long lens on the rim: silhouette at pixel scale
    dict(lens_z=75.0, lens_w=1.6, lens_h=0.9, eye=(-5.5, 29.0, 0.0)),    # 100x zoom: single triangles fill the image
]


@pytest.mark.parametrize("k", range(len(CAMERAS)))
def test_masked_frames_equal_oracle_and_unmasked(H, O, scenes, k):
    cam = _cam(H, **CAMERAS[k])
    w, h, spp = (333, 207, 3) if k % 2 else (320, 184, 4)            # 333: the mask's last word of a row is partial
    c = scenes.main_rs_scene(O.OracleScene(), O)
    want = c.render(w, h, spp=spp, seed=11 + k, camera=_ocam(O, cam))
    hits = None
    for mode in (1, 2, 0):
        g = scenes.main_rs_scene(H.World(device=0), H.loaders)
        g.set_cull_mode(mode)
        got = g.render(w, h, spp=spp, seed=11 + k, camera=cam)
        _same(got, want)
        assert got["stats"]["cull_violations"] == 0, (k, mode)
        assert got["stats"]["rays_traced"] == int(want["counters"][0])
        if mode == 2:
            big = g.render(1920, 1080, spp=2, seed=3, camera=cam, want=("rgb", "ids"))
            assert big["stats"]["cull_violations"] == 0, k
            hits = big["ids"][..., 0] == 0
            bits, every = g.cover_mask(0, 1920, 1080, camera=cam)
            # sample 0 is the un-jittered ray: wherever it hits the UFO the pixel must be set
            assert every or not (hits & ~bits).any(), k
            if k == 0:
                # and the mask is worth having: the default view has the UFO on ~1/7 of the image, the mask
                # (bounding rectangles, one pixel of slack each side) must stay within 1.5x of that
                assert not every
                assert hits.mean() > 0.05 and bits.mean() < 1.5 * hits.mean(), (hits.mean(), bits.mean())
            if k in (2, 3, 4):
                assert every                                      # triangles at / behind the eye plane
    assert hits is not None


def test_masks_with_big_triangles_and_several_meshes(H, O, scenes):
    """A coarse grid mesh seen from close by (its triangles cover thousands of pixels each: the wave-cooperative
    fill), the UFO and the cube (12 triangles: below the mask threshold) in one world; and that world under a
    Scene<T> group, where a member mesh is reached through the group walk."""
    def grid(n, origin, scale):
        z = lambda i, j: 0.5 * ((i + j) & 1)
        v = []
        for j in range(n):
            for i in range(n):
                p00, p10 = (i, j, z(i, j)), (i + 1, j, z(i + 1, j))
                p01, p11 = (i, j + 1, z(i, j + 1)), (i + 1, j + 1, z(i + 1, j + 1))
                v += [p00, p10, p11, p00, p11, p01]
        v = np.array(v, np.float64) * scale + np.array(origin, np.float64)
        tri = v.reshape(-1, 3, 3)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        nrm /= np.linalg.norm(nrm, axis=1)[:, None]
        return tri.reshape(-1, 9), nrm

    def build(b, loaders, grouped):
        scenes.main_rs_scene(b, loaders)
        m = b.add_material((0.4, 0.3, 0.2), 0.9, 1.0, 0.2, 0.2, 10)
        verts, normals = grid(12, (-30.0, -12.0, 40.0), 5.0)          # 288 triangles, 5 units each, from 40 units away
        b.add_mesh(verts, normals, m)
        if grouped:
            b.add_group([0, 1, 2, 3])
        return b

    for grouped in (False, True):
        c = build(O.OracleScene(), O, grouped)
        for cam in (_cam(H), _cam(H, eye=(0.0, -8.0, 30.0), lens_z=40.0)):
            want = c.render(352, 198, spp=2, seed=5, camera=_ocam(O, cam))
            for mode in (1, 2):
                g = build(H.World(device=0), H.loaders, grouped)
                g.set_cull_mode(mode)
                got = g.render(352, 198, spp=2, seed=5, camera=cam)
                _same(got, want)
                assert got["stats"]["cull_violations"] == 0
                if mode == 2:
                    big = g.render(1920, 1080, spp=1, seed=1, camera=cam, want=("rgb", "ids"))
                    assert big["stats"]["cull_violations"] == 0
                    for ent in (0, 3):
                        bits, every = g.cover_mask(ent, 1920, 1080, camera=cam)
                        assert every or not ((big["ids"][..., 0] == ent) & ~bits).any()


def test_masked_tiles_of_several_ranks_assemble_to_the_frame(H, O, scenes):
    """Every rank fills the mask of the whole image and looks its own tiles up in it."""
    import torch
    from rustraytracer_amd import host as HH
    w, h, spp = 480, 270, 4
    c = scenes.main_rs_scene(O.OracleScene(), O)
    want = c.render(w, h, spp=spp, seed=9)
    g = scenes.main_rs_scene(H.World(device=0), H.loaders)
    g.commit()
    for world in (2, 3):
        slots = HH.tile_slots(w, h, world)
        gathered = torch.zeros((world, slots * 256 * 3), dtype=torch.uint8, device="cuda:0")
        for r in range(world):
            st = HH.make_settings(w, h, spp, 2, 9, 0, r, world, 0)
            g.render_device(st, gathered[r].data_ptr())
        out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        HH.assemble_tiles(gathered.data_ptr(), w, h, world, out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want["rgb"]), world


def test_random_cameras_verify_masks_and_bins(H, scenes):
    """Conservativeness under fire: 48 random cameras around the 63-entity world (UFO, cube, 60 spheres) and the
    main.rs world -- eye anywhere from inside the scene to 400 units out, lens distances from fish-eye to tele,
    lens sizes that put the scene on and off the image -- each rendered with the masks and bins verified
    (cull mode 2: every masked mesh and every binned entity is collided anyway and must not answer) and compared
    with the frame rendered with everything off (mode 0) and on (mode 1).  GPU against GPU: the oracle pins the
    unmasked path elsewhere."""
    rng = np.random.default_rng(20261111)
    builders = [lambda b: scenes.many_entity_scene(b, H.loaders, grouped=False),
                lambda b: scenes.many_entity_scene(b, H.loaders, grouped=True),
                lambda b: scenes.main_rs_scene(b, H.loaders)]
    worlds = []
    for build in builders:
        trio = []
        for mode in (0, 1, 2):
            g = build(H.World(device=0))
            g.set_cull_mode(mode)
            trio.append(g)
        worlds.append(trio)
    for n in range(48):
        trio = worlds[n % len(worlds)]
        r = 10.0 ** rng.uniform(0.3, 2.6)
        v = rng.standard_normal(3)
        v /= np.linalg.norm(v)
        eye = np.array([0.0, 10.0, 40.0]) + r * v
        eye[2] = min(eye[2], 60.0 if n % 3 else 200.0)          # mostly in front of the scene, sometimes past it
        lens_z = 10.0 ** rng.uniform(0.5, 2.5)
        scale = 10.0 ** rng.uniform(-0.5, 0.5)
        cam = _cam(H, eye=tuple(eye), lens_z=lens_z, lens_w=160.0 * scale, lens_h=90.0 * scale)
        w, h = (1920, 1080) if n % 4 == 0 else (999, 563)       # bins need 2^20 path samples; 999: ragged rows
        spp = 1 if n % 4 == 0 else 2
        base = trio[0].render(w, h, spp=spp, seed=n, camera=cam, want=("rgb", "ids", "t"))
        for g in trio[1:]:
            got = g.render(w, h, spp=spp, seed=n, camera=cam, want=("rgb", "ids", "t"))
            _same(got, base, keys=("rgb", "ids", "t"))
            assert got["stats"]["cull_violations"] == 0, n
