#!/usr/bin/env python3
"""ms per cfg2 frame over N back-to-back rrt_render_device calls (host clock around a synchronised run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustraytracer_amd import host as H, scenes
w = scenes.cfg2_scene(H.World(device=0), H.loaders); w.commit()
st = H.make_settings(1920, 1080, spp=16)
buf = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
for _ in range(5): w.render_device(st, buf.data_ptr())
torch.cuda.synchronize()
n = int(os.environ.get("N", "100"))
t0 = time.perf_counter()
for _ in range(n): w.render_device(st, buf.data_ptr())
torch.cuda.synchronize()
print("%.4f ms/frame" % ((time.perf_counter() - t0) / n * 1e3))
