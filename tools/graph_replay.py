"""The entry point that allocates nothing under HIP graph capture: a frame's launch chain captured once (torch.cuda.CUDAGraph
on the capture stream) and replayed -- one graph launch instead of the frame's separate launches.  Prints ms per frame
for direct calls and for replays, and checks that a replay leaves the same bytes.
   python tools/graph_replay.py [cfg1|cfg2] [spp]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rustraytracer_amd import host as H, scenes  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
w, h = (640, 480) if cfg == "cfg1" else (960, 540)
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = getattr(scenes, cfg + "_scene")(H.World(device=0), H.loaders)
g.commit()
st = H.make_settings(w, h, spp=spp, seed=3)
out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
ref = torch.zeros_like(out)
g.render_device(st, ref.data_ptr())
torch.cuda.synchronize()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.render_device_noalloc(st, out.data_ptr(), stream=s.cuda_stream)      # warm: symbol lookup, lazy loads
s.synchronize()
assert torch.equal(out, ref)
graph = torch.cuda.CUDAGraph()
out.zero_()
with torch.cuda.graph(graph, stream=s):
    g.render_device_noalloc(st, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(out, ref), "replay differs"
n = 200


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.cuda.stream(s):
    direct = timed(lambda: g.render_device_noalloc(st, out.data_ptr(), stream=s.cuda_stream))
replay = timed(graph.replay)
out.zero_()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(out, ref)
print("%s %dx%d %d spp: direct calls %.4f ms/frame, graph replay %.4f ms/frame, same bytes" % (cfg, w, h, spp, direct, replay))
