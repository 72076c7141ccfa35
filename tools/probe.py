#!/usr/bin/env python3
"""Kernel-time probes: where does the primary kernel's time go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustraytracer_amd import host as H, scenes

W, Hh, SPP = 1920, 1080, 16
def run(name, w):
    w.commit()
    st = H.make_settings(W, Hh, spp=SPP)
    buf = torch.empty((Hh, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        w.render_device(st, buf.data_ptr())
    torch.cuda.synchronize()
    s = w.stats()
    print("%-28s gpu %.3f ms  primary %.3f  bounce %.3f  rays %d levels %s" % (name, s["gpu_ms"], s["primary_kernel_ms"], s["bounce_kernel_ms"], s["rays_traced"], s["level_rays"]))

a = H.World(device=0); scenes.empty_world(a, H.loaders, "blue_sunset"); run("sky only", a)
b = H.World(device=0); scenes.empty_world(b, H.loaders, "blue_sunset")
m1 = b.add_material(scenes.color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20); m2 = b.add_material(scenes.color_vec(0, 0, 0), 0.8, 0.0, 1.0, 0.1, 20)
b.add_sphere((-3.0, 0.0, 5.0), 1.0, m1); b.add_sphere((2.0, 0.0, 8.0), 1.0, m2); run("sky + 2 spheres", b)
c = H.World(device=0); scenes.empty_world(c, H.loaders, "blue_sunset")
m3 = c.add_material(scenes.color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20)
v, n = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, 70.0), (1.0, -1.0, 1.0))
c.add_mesh(v, n, m3); run("sky + ufo", c)
d = scenes.cfg2_scene(H.World(device=0), H.loaders); run("cfg2", d)
e = H.World(device=0); scenes.empty_world(e, H.loaders, "blue_sunset")
m1 = e.add_material(scenes.color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
e.add_sphere((-3.0, 0.0, -50.0), 1.0, m1); e.add_sphere((2.0, 0.0, -80.0), 1.0, m1); run("2 spheres behind the camera", e)
f = H.World(device=0); scenes.empty_world(f, H.loaders, "blue_sunset")
m1 = f.add_material(scenes.color_vec(100, 100, 200), 0.8, 1.0, 0.01, 0.1, 20)
f.add_sphere((-3.0, 0.0, 5.0), 1.0, m1); run("1 matte sphere (mixR 0.01)", f)
g = H.World(device=0); scenes.empty_world(g, H.loaders, "blue_sunset")
v, n = H.loaders.load_model(os.path.join(scenes.ASSETS, "obj", "ufo_fix.obj"), (0.0, 30.0, -170.0), (1.0, -1.0, 1.0))
m3 = g.add_material(scenes.color_vec(100, 100, 50), 1.0, 1.0, 0.0, 0.3, 20)
g.add_mesh(v, n, m3); run("ufo behind the camera", g)
