"""Cost of the sky-only arm: the cfg2 frame with the entities replaced by two tiny spheres off screen (every tile bin empty),
against the real cfg2 frame.   python tools/sky_only_ms.py [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rustraytracer_amd import host as H, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
buf = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
def run(w, tag):
    w.commit(); w.set_timing(2)
    st = H.make_settings(1920, 1080, spp=spp)
    for _ in range(4): w.render_device(st, buf.data_ptr())
    torch.cuda.synchronize()
    s = w.stats()
    print("%s spp %d: gpu %.3f ms primary %.3f bounce %.3f  rays %s" % (tag, spp, s["gpu_ms"], s["primary_kernel_ms"], s["bounce_kernel_ms"], s["level_rays"]))
w = H.World(device=0)
scenes.empty_world(w, H.loaders, "blue_sunset")
m = w.add_material((0.5, 0.5, 0.5), 1.0, 1.0, 0.0, 0.3, 20)
w.add_sphere((1e5, 0.0, 10.0), 1.0, m); w.add_sphere((-1e5, 0.0, 10.0), 1.0, m)
run(w, "sky-only")
run(scenes.cfg2_scene(H.World(device=0), H.loaders), "cfg2")
