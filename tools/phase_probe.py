"""Wave-level event counts of one cfg2 frame (diagnostic build: make -C rustraytracer_amd/csrc
probe; run with RRT_LIB=rustraytracer_amd/librrt_probe.so).  For each site: waves that reached it,
active lanes, and the per-ray averages -- multiply by the block's static instruction count in
rrt_kernels.s for the issue budget."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustraytracer_amd import binding as B, host as H, scenes  # noqa: E402

NAMES = ["cast (process_ray)", "mesh_collide entered", "root box passed", "descent steps",
         "leaf visits", "group-cull trips (4 groups)", "exact tests", "leaf advances", "sky lookups", "shade_direct",
         "path folds", "entry-cull trips (1 group)"]
cfg = os.environ.get("CFG", "cfg2")
w = getattr(scenes, cfg + "_scene")(H.World(device=0), H.loaders)
spp = int(os.environ.get("SPP", "4"))
out = (ctypes.c_ulonglong * 32)()
w.render(1920, 1080, spp=spp, seed=1, want=("rgb",))
B.lib.rrt_debug_phases(out)          # read + clear (warm-up frame)
r = w.render(1920, 1080, spp=spp, seed=1, want=("rgb",))
B.lib.rrt_debug_phases(out)
rays = r["stats"]["rays_traced"]
print("rays %d  (wave-rays if fully packed: %d)" % (rays, rays // 64))
print("%-24s %14s %14s %10s %12s %12s" % ("site", "waves", "lanes", "lanes/wave", "waves/wray", "lanes/ray"))
wr = max(out[0], 1)
for i, n in enumerate(NAMES):
    wv, ln = out[2 * i], out[2 * i + 1]
    print("%-24s %14d %14d %10.1f %12.2f %12.2f" % (n, wv, ln, ln / max(wv, 1), wv / wr, ln / max(out[1], 1)))

t = [out[24 + i] for i in range(8)]
if t[2]:
    print("bounce level 1, per 64-ray unit: grab + queue read %.0f cycles, process_ray %.0f cycles (%d units); per wave: %.0f cycles in the unit loop (%d waves)"
          % (t[0] / t[2], t[1] / t[2], t[2], t[3] / max(t[4], 1), t[4]))
    print("   units whose rays are all dead (mix 0.0 parents): %d, process_ray %.0f cycles each; the others: %d, %.0f cycles each"
          % (t[6], t[5] / max(t[6], 1), t[2] - t[6], (t[1] - t[5]) / max(t[2] - t[6], 1)))
    print("   of which closest_hit (lane 0's clock): %.0f cycles per unit with live rays" % (t[7] / max(t[2] - t[6], 1)))
