"""Per-rank sums of the true per-tile hit-pixel counts under the dealt and the positional tile map (cfg2, 1080p)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rustraytracer_amd import host as H, scenes, distributed as D, tiles
W, Hh, world = 1920, 1080, 8
g = scenes.cfg2_scene(H.World(device=0), H.loaders)
ids = g.render(W, Hh, spp=1, want=("ids",))["ids"][..., 0]
hit = (ids >= 0).reshape(Hh // 8, 8, W // 32, 32).sum(axis=(1, 3)).reshape(-1)          # per tile
ufo = (ids == 0).reshape(Hh // 8, 8, W // 32, 32).sum(axis=(1, 3)).reshape(-1)
dev = torch.device("cuda", 0)
local, _ = D.alloc_tile_buffers(W, Hh, world, dev, dealt=True)
tabs = []
for r in range(world):
    g.render_device(H.make_settings(W, Hh, spp=16, tile_rank=r, tile_world=world, tile_deal=1), local.data_ptr())
    torch.cuda.synchronize()
    s = g.stats()
    _, table = tiles.split_dealt(local.cpu().numpy(), W, Hh, world)
    t = table[table >= 0]
    tabs.append(table.copy())
    nh = s["heavy_tiles"]
    print("rank %d: heavy %d  hit px %d (ufo %d)  first heavy tiles %s  hits of first 12 slots %s  lvl1 %d" % (r, nh, hit[t].sum(), ufo[t].sum(), t[:6], hit[t[:12]], s["level_rays"][1]))
for r in range(world):
    own = np.array([tiles.tile_origin(t, W // 32, world) for t in range(r, (W // 32) * (Hh // 8), world)])
    idx = (own[:, 1] // 8) * (W // 32) + own[:, 0] // 32
    print("positional rank %d: hit px %d" % (r, hit[idx].sum()))
np.savez("gpurun_out/deal_check.npz", hit=hit, ufo=ufo, tabs=np.stack(tabs))
