"""ms per rrt_render call (blocking, host buffer) on the cfg2 frame: pageable numpy array against pinned memory.
   python tools/host_render_ms.py [spp]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from rustraytracer_amd import host as H, scenes, binding as B
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
w = scenes.cfg2_scene(H.World(device=0), H.loaders); w.commit(); w.set_timing(0)
cam = H.default_camera(); st = H.make_settings(1920, 1080, spp=spp)
page = np.zeros((1080, 1920, 3), np.uint8)
n = 1920 * 1080 * 3
p = B.lib.rrt_alloc_host_pinned(n)
pin = np.frombuffer((C.c_ubyte * n).from_address(p), dtype=np.uint8).reshape(1080, 1920, 3)
dev = torch.empty((1080, 1920, 3), dtype=torch.uint8, device="cuda")
def run(buf, reps=20):
    for _ in range(3): B.check(B.lib.rrt_render(w.h, C.byref(cam), C.byref(st), buf.ctypes.data, None, None, None))
    t0 = time.perf_counter()
    for _ in range(reps): B.check(B.lib.rrt_render(w.h, C.byref(cam), C.byref(st), buf.ctypes.data, None, None, None))
    return (time.perf_counter() - t0) / reps * 1e3
def run_dev(reps=20):
    for _ in range(3): w.render_device(st, dev.data_ptr()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): w.render_device(st, dev.data_ptr()); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for rep in range(2):
    print("spp %d: rrt_render pageable %.3f ms, pinned %.3f ms; rrt_render_device + sync %.3f ms" % (spp, run(page), run(pin), run_dev()))
assert np.array_equal(page, pin)
