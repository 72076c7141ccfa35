"""rrt_settings.tile_deal: the image's tiles dealt to the ranks by cost class (a mesh can be met / only spheres can /
only sky, from the frame's tile bins) instead of by position; every rank's buffer carries its slot -> tile table, the
root rebuilds the frame from the gathered bytes alone (rrt_assemble_tiles_dealt).  No reference counterpart (the
reference splits the image into 12 contiguous pixel ranges, tracer.rs:117-158); the image must not depend on it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,Hh,spp", [(500, 270, 3), (960, 540, 5)])       # without / with tile bins (>= 2^20 path samples)
def test_dealt_tiles_assemble_to_full_frame(H, O, scenes, W, Hh, spp):
    import torch
    from rustraytracer_amd import tiles, distributed as D
    g = scenes.main_rs_scene(H.World(device=0), H.loaders, skybox="sky")
    full = g.render(W, Hh, spp=spp, seed=9, want=("rgb",))
    full_paths = full["stats"]["paths"]
    full = full["rgb"]
    if W * Hh <= 500 * 270:
        c = scenes.main_rs_scene(O.OracleScene(), O, skybox="sky").render(W, Hh, spp=spp, seed=9, want=("rgb",))["rgb"]
        assert np.array_equal(full, c)
    tx, ty = tiles.tile_grid(W, Hh)
    for world in (2, 3, 8):
        for mode in (1, 2):
            g.set_cull_mode(mode)
            assert H.tile_buffer_bytes(W, Hh, world, 1) == tiles.dealt_buffer_bytes(W, Hh, world)
            _, gathered = D.alloc_tile_buffers(W, Hh, world, torch.device("cuda", 0), dealt=True)
            paths, heavy, rays = 0, [], 0
            for r in range(world):
                st = H.make_settings(W, Hh, spp=spp, seed=9, tile_rank=r, tile_world=world, tile_deal=1)
                g.render_device(st, gathered[r].data_ptr())
                torch.cuda.synchronize()
                s = g.stats()
                paths += s["paths"]
                rays += s["rays_traced"]
                heavy.append(s["heavy_tiles"])
                assert s["cull_violations"] == 0
            out = torch.zeros((Hh, W, 3), dtype=torch.uint8, device="cuda:0")
            H.assemble_tiles(gathered.data_ptr(), W, Hh, world, out.data_ptr(), dealt=1)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), full), (world, mode)
            assert paths == full_paths == W * Hh * spp
            gh = gathered.cpu().numpy()
            assert np.array_equal(tiles.assemble_dealt_numpy(gh, W, Hh, world), full)
            tables = np.stack([tiles.split_dealt(gh[r], W, Hh, world)[1] for r in range(world)])
            assert np.array_equal(np.sort(tables[tables >= 0]), np.arange(tx * ty))         # a partition of the image
            if mode == 1:
                assert max(heavy) - min(heavy) <= 1, heavy                                   # each rank its share of the heavy tiles
    g.set_cull_mode(1)
    # the positional map is unchanged by all this
    slots = H.tile_slots(W, Hh, 3)
    gathered = torch.zeros((3, slots, 8, 32, 3), dtype=torch.uint8, device="cuda:0")
    for r in range(3):
        g.render_device(H.make_settings(W, Hh, spp=spp, seed=9, tile_rank=r, tile_world=3), gathered[r].data_ptr())
    out = torch.zeros((Hh, W, 3), dtype=torch.uint8, device="cuda:0")
    H.assemble_tiles(gathered.data_ptr(), W, Hh, 3, out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full)


def test_dealt_ranks_cost_the_same(H, scenes):
    """What the dealing is for: on the bench frame the eight ranks' shares hold the same number of heavy tiles
    (to within one), where the positional map leaves them a few per cent apart."""
    import torch
    from rustraytracer_amd import distributed as D
    g = scenes.cfg2_scene(H.World(device=0), H.loaders)
    W, Hh = 1920, 1080
    dev = torch.device("cuda", 0)
    spread = {}
    for dealt in (1, 0):
        local, _ = D.alloc_tile_buffers(W, Hh, 8, dev, dealt=bool(dealt))
        heavy, lvl1 = [], []
        for r in range(8):
            g.render_device(H.make_settings(W, Hh, spp=16, tile_rank=r, tile_world=8, tile_deal=dealt), local.data_ptr())
            torch.cuda.synchronize()
            s = g.stats()
            heavy.append(s["heavy_tiles"])
            lvl1.append(s["level_rays"][1])
        spread[dealt] = max(lvl1) / float(min(lvl1))
        if dealt:
            assert max(heavy) - min(heavy) <= 1, heavy
        else:
            assert max(heavy) - min(heavy) > 1
    # ... and numbers of bounce rays within a few per cent of each other (the cost estimate is per tile, not per pixel;
    # the positional map's lattice happens to do as well on this frame -- what the deal evens out is the heavy TILES,
    # i.e. the primary kernel's work)
    assert spread[1] < 1.08 and spread[0] < 1.08, spread
