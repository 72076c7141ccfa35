"""CPU-side checks of the product's host layer (no GPU compute calls):
the C-ABI library loads and exports every declared symbol, its C++ asset
loaders agree with the oracle-side numpy/PIL loaders, and its kd-tree builder
emits the same flat tree as the oracle's restatement of kdtree.rs."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(H):
    from rustraytracer_amd import binding as B
    declared = set()
    for hdr in ("rrt.h", "rrt_host.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(B.EXPORTS)
    for name in declared:
        assert hasattr(B.lib, name), name
    assert b"gfx950" in B.lib.rrt_version()


def test_rust_binding_source_matches_the_c_abi(H):
    """bindings/rust/rrt-sys cannot be compiled here (no rustc); what can be checked is checked: every
    function it declares is exported by librrt.so with the same number of parameters as the C header, and
    its #[repr(C)] structs list the same fields, in order, with the same widths as the C structs."""
    from rustraytracer_amd import binding as B
    rs = open(os.path.join(ROOT, "bindings", "rust", "rrt-sys", "src", "lib.rs")).read()
    rs = re.sub(r"//[^\n]*", "", rs)
    hdr = ""
    for h in ("rrt.h", "rrt_host.h"):
        hdr += re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
    c_fns = {m.group(1): m.group(2) for m in re.finditer(r"\b(rrt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr)}
    rs_fns = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rrt_[a-z0-9_]+)\s*\(([^)]*)\)", rs)}
    assert len(rs_fns) >= 30
    for name, args in rs_fns.items():
        assert name in c_fns and hasattr(B.lib, name), name
        n_rs = 0 if not args.strip() else len([a for a in args.split(",") if a.strip()])
        c_args = c_fns[name].strip()
        n_c = 0 if c_args in ("", "void") else len(c_args.split(","))
        assert n_rs == n_c, (name, n_rs, n_c)
    width = {"u32": 4, "u64": 8, "f64": 8, "i32": 4, "uint32_t": 4, "uint64_t": 8, "double": 8, "int32_t": 4}
    for st in ("rrt_camera", "rrt_settings", "rrt_stats"):
        cbody = re.search(r"typedef struct %s \{(.*?)\}" % st, hdr, re.S).group(1)
        cf = [(m.group(2), width[m.group(1)] * int(m.group(3) or 1))
              for m in re.finditer(r"(\w+)\s+(\w+)(?:\[(\d+)\])?;", cbody)]
        rbody = re.search(r"pub struct %s \{(.*?)\}" % st, rs, re.S).group(1)
        rf = [(m.group(1), width[m.group(3) or m.group(2)] * int(m.group(4) or 1))
              for m in re.finditer(r"pub (\w+): (?:(\w+)|\[(\w+); (\d+)\]),", rbody)]
        assert cf == rf and cf, (st, cf, rf)
        assert sum(w for _, w in cf) == C.sizeof({"rrt_camera": B.Camera, "rrt_settings": B.Settings, "rrt_stats": B.Stats}[st])


def test_loaders_agree_with_oracle_side(H, O, scenes):
    for obj, pos, sc in (("ufo_fix.obj", (0, 30, 70), (1, -1, 1)), ("cube.obj", (0, 0, 12), (2, 2, 2))):
        p = os.path.join(scenes.ASSETS, "obj", obj)
        vo, no = O.load_model(p, pos, sc)
        vp, np_ = H.loaders.load_model(p, pos, sc)
        assert np.array_equal(vo, vp) and np.array_equal(no, np_)
    for cm in ("sky", "hd_blue_sunset"):   # RGB, palette ('sky/up.png') and RGBA PNGs
        a = O.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", cm))
        b = H.loaders.load_cubemap(os.path.join(scenes.ASSETS, "cubemaps", cm))
        assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_png_round_trip(H, tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p = str(tmp_path / "x.png")
    H.loaders.save_png(p, img)
    assert np.array_equal(H.loaders.load_png(p), img)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(p)), img)


def test_loader_errors_are_codes_not_aborts(H, tmp_path):
    from rustraytracer_amd import binding as B
    with pytest.raises(B.RrtError, match="cannot open"):
        H.loaders.load_model(str(tmp_path / "missing.obj"), (0, 0, 0), (1, 1, 1))
    quad = tmp_path / "quad.obj"
    quad.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n")
    with pytest.raises(B.RrtError, match="not a triangle"):     # obj-rs refuses untriangulated models
        H.loaders.load_model(str(quad), (0, 0, 0), (1, 1, 1))
    nonorm = tmp_path / "nonorm.obj"
    nonorm.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3\nf 1 2 3\n")
    with pytest.raises(B.RrtError, match="normal"):
        H.loaders.load_model(str(nonorm), (0, 0, 0), (1, 1, 1))
    with pytest.raises(B.RrtError):
        H.loaders.load_png(os.path.join(ROOT, "assets", "obj", "cube.obj"))


def test_scene_argument_errors(H):
    from rustraytracer_amd import binding as B
    w = H.World()
    with pytest.raises(B.RrtError, match="unknown material"):
        w.add_sphere((0, 0, 0), 1.0, 3)
    m = w.add_material((1, 1, 1), 1, 1, 0, 0, 20)
    assert m == 0 and w.add_sphere((0, 0, 5), 1.0, m) == 0
    with pytest.raises(B.RrtError, match="not a mesh"):
        w.tree(0)
    with pytest.raises(B.RrtError, match="bad face"):
        B.check(B.lib.rrt_set_sky_face(w.h, 7, np.zeros(3, np.uint8).ctypes.data, 1, 1))
    # rrt_cover_mask checks its arguments before it touches the device (no GPU here)
    with pytest.raises(B.RrtError, match="not a mesh"):
        w.cover_mask(0, 64, 32)
    with pytest.raises(B.RrtError, match="not a mesh"):
        w.cover_mask(5, 64, 32)
    with pytest.raises(B.RrtError, match="empty image"):
        w.cover_mask(0, 0, 32)
    assert H.tile_slots(1920, 1080, 1) == 60 * 135
    assert H.tile_slots(1920, 1080, 8) == (60 * 135 + 7) // 8
    assert H.tile_slots(33, 9, 2) == 2


def _trees_equal(a, b):
    return all(np.array_equal(a[k], b[k]) for k in a)


def test_kd_tree_identical_to_oracle_ufo_and_cube(H, O, scenes):
    for fn in (scenes.main_rs_scene, scenes.cfg1_scene):
        so = fn(O.OracleScene(), O)
        sp = fn(H.World(), H.loaders)
        to, tp = so.tree(0), sp.tree(0)
        assert _trees_equal(to, tp)
    # structural facts of the UFO tree
    leaf = to["axis"] < 0 if fn is scenes.main_rs_scene else None


def _random_mesh(rng, n, quant=None, spread=20.0, size=1.5):
    c = rng.uniform(-spread, spread, (n, 1, 3)) + np.array([0, 0, 60.0])
    v = c + rng.uniform(-size, size, (n, 3, 3))
    if quant:                      # many tied bbox-min keys, like a modelled mesh
        v = np.round(v / quant) * quant
    nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    nrm += (np.linalg.norm(nrm, axis=1, keepdims=True) == 0) * np.array([0, 0, 1.0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return v, nrm


@pytest.mark.parametrize("n,quant,seed", [(5, None, 0), (6, 1.0, 1), (64, None, 2), (300, 0.5, 3),
                                          (5000, 1.0, 4), (5000, None, 5), (40000, 2.0, 6)])
def test_kd_tree_identical_to_oracle_random(H, O, n, quant, seed):
    """The sort's tie order, block partition (>256 items), pattern breaking and
    the no-progress leaf rule all get exercised by quantised meshes."""
    rng = np.random.default_rng(seed)
    v, nrm = _random_mesh(rng, n, quant)
    so, sp = O.OracleScene(), H.World()
    for s in (so, sp):
        m = s.add_material((0.5, 0.5, 0.5), 1.0, 1.0, 0.0, 0.3, 20)
        s.add_mesh(v, nrm, m)
    to, tp = so.tree(0), sp.tree(0)
    assert _trees_equal(to, tp)
    leaf = to["axis"] < 0
    # every triangle is referenced at least once; rope targets are valid nodes
    assert set(np.unique(to["leaf_refs"])) == set(range(n))
    r = to["ropes"][leaf]
    assert r.min() >= -1 and r.max() < len(to["axis"])
    assert (to["ropes"][~leaf] == -1).all()


def test_group_tree_identical_to_oracle(H, O, scenes):
    """Scene<T>'s kd-tree over entity bounding boxes (scene.rs:39-47): 3 members = one leaf behind
    the union box; 62 members = a real tree.  Same generic builder as the meshes', same oracle check."""
    from rustraytracer_amd import binding as B
    so, sp = scenes.main_rs_as_scene(O.OracleScene(), O), scenes.main_rs_as_scene(H.World(), H.loaders)
    to, tp = so.tree(3), sp.tree(3)
    assert _trees_equal(to, tp)
    assert list(to["axis"]) == [-1] and list(to["leaf_refs"]) == [0, 1, 2]
    so, sp = scenes.many_entity_scene(O.OracleScene(), O), scenes.many_entity_scene(H.World(), H.loaders)
    gid = 62
    to, tp = so.tree(gid), sp.tree(gid)
    assert _trees_equal(to, tp)
    assert (to["axis"] >= 0).sum() > 10 and set(np.unique(to["leaf_refs"])) == set(range(62))
    # argument errors: unknown id, duplicate, nesting, member reuse
    w = scenes.main_rs_scene(H.World(), H.loaders)
    for bad in ([0, 7], [1, 1], []):
        with pytest.raises(B.RrtError):
            w.add_group(bad)
    g = w.add_group([1, 2])
    with pytest.raises(B.RrtError, match="top-level"):
        w.add_group([1])                      # a member of g already
    # Scene<Scene<T>>: a group may hold groups, one level deep; the outer tree is built over the inner root boxes
    so = scenes.main_rs_scene(O.OracleScene(), O)
    go = so.add_group([1, 2])
    outer_p, outer_o = w.add_group([0, g]), so.add_group([0, go])
    assert _trees_equal(so.tree(outer_o), w.tree(outer_p))
    extra = w.add_sphere((0.0, 0.0, 9.0), 1.0, 0)
    with pytest.raises(B.RrtError, match="one level"):
        w.add_group([outer_p, extra])
    with pytest.raises(B.RrtError, match="top-level"):
        w.add_group([g, extra])               # g is a member of the outer group now


def test_rope_faces_touch(H):
    """A leaf's rope through face i leads to a node whose box touches that face."""
    rng = np.random.default_rng(9)
    v, nrm = _random_mesh(rng, 2000)
    w = H.World()
    w.add_mesh(v, nrm, w.add_material((1, 1, 1), 1, 1, 0, 0, 20))
    t = w.tree(0)
    root = t["boxes"][0]
    for i in np.nonzero(t["axis"] < 0)[0]:
        for k in range(6):
            nb = t["ropes"][i, k]
            a = k % 3
            mine = t["boxes"][i, a] if k < 3 else t["boxes"][i, 3 + a]
            if nb < 0:
                assert mine == (root[a] if k < 3 else root[3 + a])
            else:
                theirs = t["boxes"][nb, 3 + a] if k < 3 else t["boxes"][nb, a]
                assert theirs == mine


def test_max_bounce_is_range_checked_before_any_launch(H):
    """max_bounce + 2 levels must fit kMaxLevels = 18; the check may not wrap in u32 (0xFFFFFFFF is the
    usual 'unlimited' idiom).  Reported as RRT_E_INVALID before the device is touched (this test has no GPU)."""
    import ctypes as C
    from rustraytracer_amd import binding as B
    w = H.World()
    cam = H.default_camera()
    rgb = np.zeros((4, 4, 3), np.uint8)
    rays = np.zeros((2, 6))
    col = np.zeros((2, 3))
    for mb in (17, 0xFFFFFFFE, 0xFFFFFFFF):
        st = H.make_settings(4, 4, max_bounce=mb)
        assert B.lib.rrt_render(w.h, C.byref(cam), C.byref(st), rgb.ctypes.data, None, None, None) == -1
        assert b"max_bounce" in B.lib.rrt_last_error()
        assert B.lib.rrt_render_device(w.h, C.byref(cam), C.byref(st), None, None, None, None, None) == -1
        assert B.lib.rrt_cast_rays(w.h, rays.ctypes.data, 2, mb, 0, col.ctypes.data) == -1
        assert b"max_bounce" in B.lib.rrt_last_error()


def test_translate_host_side_matches_oracle_and_rebuild(H, O, scenes):
    """rrt_translate_entity on the host (no GPU): literal = split planes move and boxes stay
    (kdtree.rs:35-42), refit = boxes follow; both equal the oracle's restatement array for array, and a
    refit translation by an exactly representable vector equals the tree built at the new place."""
    import os
    p = os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj")
    v, n = H.loaders.load_model(p, (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
    vec = (4.0, -2.0, 8.0)
    for refit in (0, 1):
        g, c = H.World(), O.OracleScene()
        for s in (g, c):
            s.add_mesh(v, n, s.add_material((0.5, 0.5, 0.5), 1.0, 1.0, 0.0, 0.3, 20))
        before = g.tree(0)
        g.translate(0, vec, refit); c.translate(0, vec, refit)
        tg, tc = g.tree(0), c.tree(0)
        assert all(np.array_equal(tg[k], tc[k]) for k in tg)
        interior = before["axis"] >= 0
        assert np.array_equal(tg["split"][interior], before["split"][interior] + np.array(vec)[before["axis"][interior]])
        if refit:
            fresh = H.World()
            fresh.add_mesh(v + np.array(vec), n, fresh.add_material((0.5, 0.5, 0.5), 1.0, 1.0, 0.0, 0.3, 20))
            tf = fresh.tree(0)
            assert all(np.array_equal(tg[k], tf[k]) for k in tg)
        else:
            assert np.array_equal(tg["boxes"], before["boxes"])          # the reference forgets them
    from rustraytracer_amd import binding as B
    w = H.World()
    with pytest.raises(B.RrtError, match="unknown entity"):
        w.translate(0, vec)
    for i in range(16):
        assert w.add_point_light((0, 0, 0), (1, 1, 1), 1.0, 1.0) == i
    with pytest.raises(B.RrtError, match="too many"):
        w.add_point_light((0, 0, 0), (1, 1, 1), 1.0, 1.0)
    with pytest.raises(B.RrtError, match="shadow mode"):
        w.set_shadow_mode(2)


def test_png_decoder_fixtures(H, golden_dir):
    """to_rgb8 over the whole PNG format: Adam7, 16-bit, packed grey and palettes, grey+alpha,
    all five filters, two IDAT chunks, sizes with empty Adam7 passes.  Files and expected images come from
    tests/golden/make_png_fixtures.py; PIL decodes them too, as an independent reader (where its own to-RGB
    conversion means the same thing: 8-bit and palette files)."""
    from PIL import Image
    d = os.path.join(golden_dir, "png")
    exp = np.load(os.path.join(d, "expected.npz"))
    assert len(exp.files) == 17
    for name in exp.files:
        got = H.loaders.load_png(os.path.join(d, name + ".png"))
        assert got.shape == exp[name].shape and np.array_equal(got, exp[name]), name
        if "16" not in name and not name.startswith("grey1") and not name.startswith("grey2") and not name.startswith("grey4"):
            pil = np.asarray(Image.open(os.path.join(d, name + ".png")).convert("RGB"))
            assert np.array_equal(pil, exp[name]), name
    from rustraytracer_amd import binding as B
    bad = os.path.join(d, "expected.npz")
    with pytest.raises(B.RrtError, match="not a PNG"):
        H.loaders.load_png(bad)


def test_obj_triangulation_loads_the_reference_ufo(H, O, scenes, tmp_path):
    """RRT_OBJ_TRIANGULATE (extension): quads and n-gons become fans, so the file BASELINE.json names, obj/ufo.obj
    (789 quads, 38 triangles, 12 n-gons with v/vt/vn corners), loads -- same triangles as the oracle-side numpy
    loader produces, and its kd-tree equals the oracle's.  Without the flag it is refused, as obj-rs refuses it."""
    from rustraytracer_amd import binding as B
    p = os.path.join(scenes.ASSETS, "obj", "ufo.obj")
    with pytest.raises(B.RrtError, match="not a triangle"):
        H.loaders.load_model(p, (0, 30, 70), (1, -1, 1))
    v, n = H.loaders.load_model(p, (0, 30, 70), (1, -1, 1), triangulate=True)
    vo, no = O.load_model(p, (0, 30, 70), (1, -1, 1), triangulate=True)
    assert np.array_equal(v, vo) and np.array_equal(n, no)
    assert len(v) == 38 + 789 * 2 + 8 * 3 + 2 * 15 + 2 * 18 - 1          # fans, minus the face model.rs:31 drops
    g, c = H.World(), O.OracleScene()
    for s in (g, c):
        s.add_mesh(v, n, s.add_material((0.4, 0.4, 0.2), 1.0, 1.0, 0.0, 0.3, 20))
    tg, tc = g.tree(0), c.tree(0)
    assert all(np.array_equal(tg[k], tc[k]) for k in tg)
    # a triangulated file is unchanged by the flag; a pentagon fans in corner order
    a = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0, 30, 70), (1, -1, 1), triangulate=True)
    b = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0, 30, 70), (1, -1, 1))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    pent = tmp_path / "pent.obj"
    pent.write_text("v 0 0 0\nv 2 0 0\nv 3 1 0\nv 1 3 0\nv -1 1 0\nvn 0 0 1\nf 1//1 2//1 3//1 4//1 5//1\nf 1//1 2//1 3//1\n")
    pv, _ = H.loaders.load_model(str(pent), (0, 0, 0), (1, 1, 1), triangulate=True)
    assert pv.shape == (3, 3, 3)
    assert np.array_equal(pv[:, 0, :2], [[0, 0]] * 3) and np.array_equal(pv[:, 1, :2], [[2, 0], [3, 1], [1, 3]]) \
        and np.array_equal(pv[:, 2, :2], [[3, 1], [1, 3], [-1, 1]])


def test_absent_rccl_is_a_status_code_not_a_crash():
    """A host without librccl: rrt_comm_unique_id / rrt_comm_init answer RRT_E_STATE with a message (the loader's
    dlerror() text is read ONCE -- a second call returns NULL).  RRT_RCCL_LIB points the loader at a library that
    does not exist; a process of its own, because the library is resolved once per process."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, os\n"
        "os.environ['RRT_NO_TORCH'] = '1'\n"
        "from rustraytracer_amd import binding as B\n"
        "buf = (C.c_ubyte * 128)()\n"
        "rc = B.lib.rrt_comm_unique_id(buf)\n"
        "msg = B.lib.rrt_last_error().decode()\n"
        "assert rc == -3 and 'RCCL not found' in msg and 'no_such_rccl' in msg, (rc, msg)\n"
        "out = C.c_void_p()\n"
        "rc = B.lib.rrt_comm_init(0, 1, buf, C.byref(out))\n"
        "assert rc == -3 and not out.value, rc\n"
        "assert B.lib.rrt_gather_tiles(None, None, None, 64, 64, 0, None) == -1\n"
        "print('ok')\n")
    env = dict(os.environ, RRT_RCCL_LIB="/nonexistent/libno_such_rccl.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_png_with_an_oversized_header_is_refused(H, tmp_path):
    """IHDR sizes near 2^31 / 2^32 would wrap the scanline and image byte counts: refused right after the header
    (RRT_E_FORMAT), before anything is allocated."""
    import struct
    import zlib
    from rustraytracer_amd import binding as B

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    for w, h in ((0xfffffff0, 0xfffffff0), (0x80000001, 1), (70000, 70000)):
        p = tmp_path / ("big_%x_%x.png" % (w, h))
        p.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 6, 0, 0, 0))
                      + chunk(b"IDAT", zlib.compress(b"\0")) + chunk(b"IEND", b""))
        with pytest.raises(B.RrtError, match="too large"):
            H.loaders.load_png(str(p))
