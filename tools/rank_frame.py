"""One rank's share of the strong-scaling target frame (1080p, 64 spp, tile_world 8) rendered N times on this GPU:
   python tools/rank_frame.py [rank] [frames] [tile_world]     (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rustraytracer_amd import host as H, scenes, distributed as D  # noqa: E402

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
w = scenes.cfg2_scene(H.World(device=0), H.loaders)
w.commit()
w.set_timing(int(os.environ.get('TIMING', '2')))
dev = torch.device("cuda", 0)
local, _ = D.alloc_tile_buffers(1920, 1080, world, dev)
deal = int(os.environ.get("DEAL", "1")) if world > 1 else 0
local, _ = D.alloc_tile_buffers(1920, 1080, world, dev, dealt=bool(deal))
st = H.make_settings(1920, 1080, spp=64, tile_rank=rank, tile_world=world, tile_deal=deal)
for _ in range(3):
    w.render_device(st, local.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(frames):
    w.render_device(st, local.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / frames * 1e3
s = w.stats()
print("rank %d of %d: %.3f ms/frame  (gpu %.3f: primary %.3f, bounce %.3f; %d batches; level rays %s)"
      % (rank, world, dt, s["gpu_ms"], s["primary_kernel_ms"], s["bounce_kernel_ms"], s["n_batches"], s["level_rays"]))
