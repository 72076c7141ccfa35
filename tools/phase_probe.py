#!/usr/bin/env python3
"""Per-phase wave-cycle shares of the trace kernels (needs the -DRRT_PROBE=1 library:
RRT_LIB=rustraytracer_amd/librrt_probe.so python tools/phase_probe.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustraytracer_amd import host as H, scenes, binding as B
names = ["-", "entity loop excl. mesh (spheres, ray setup)", "mesh: setup/root box/find_point + leaf pointer",
         "leaf header load (first,count)", "cull loop (boxes, loop overhead)", "exact f64 triangle tests (load+test)",
         "leaf advance (box load, exit, rope, find_point)", "finish_hit + shading + sky + records + stores"]
w = scenes.cfg2_scene(H.World(device=0), H.loaders); w.commit()
st = H.make_settings(1920, 1080, spp=16)
buf = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
out = (C.c_ulonglong * 16)()
w.render_device(st, buf.data_ptr()); torch.cuda.synchronize()
B.lib.rrt_debug_phases(out)
w.render_device(st, buf.data_ptr()); torch.cuda.synchronize()
B.lib.rrt_debug_phases(out)
tot = sum(out[:8])
s = w.stats()
print("gpu %.3f ms primary %.3f bounce %.3f (probe build; read shares, not lengths)" % (s["gpu_ms"], s["primary_kernel_ms"], s["bounce_kernel_ms"]))
for i in range(1, 8):
    print("  %-62s %6.2f %%   (lane max %d cyc)" % (names[i], 100.0 * out[i] / tot, out[8 + i]))
