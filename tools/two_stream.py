"""Frames in flight: the same frame rendered back to back on one stream against round-robin over 2, 3, 4 streams (the
scene keeps one set of work buffers per stream).   python tools/two_stream.py [world] [spp] [frames] [cfg]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from rustraytracer_amd import host as H, scenes, distributed as D  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device("cuda", 0)
if os.environ.get("NCCL1"):      # an RCCL communicator (world of one) before anything renders: its streams exist first, as in bench.py --gpus N
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.ones(1024, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
bufs, streams = [], []
_extra = []
for _p in [int(x) for x in os.environ.get("EXTRA_STREAMS", "").split(",") if x]:      # extra streams of these priorities, each used once, before anything renders
    _s = torch.cuda.Stream(device=dev, priority=_p)
    with torch.cuda.stream(_s):
        torch.ones(16, device=dev).add_(1)
    _extra.append(_s)
torch.cuda.synchronize()
w = getattr(scenes, cfg + "_scene")(H.World(device=0), H.loaders)
w.commit()
w.set_timing(0)
if os.environ.get("SKY_STREAM"):
    w.set_sky_stream(int(os.environ["SKY_STREAM"]))
NS = [int(x) for x in os.environ.get('NSTREAMS', '1,2,3,4').split(',')]
for i in range(max(NS)):
    if world > 1:
        bufs.append(D.alloc_tile_buffers(1920, 1080, world, dev, dealt=int(os.environ.get("DEAL", "1")) != 0)[0])
    else:
        bufs.append(torch.empty(1080 * 1920 * 3, dtype=torch.uint8, device=dev))
    streams.append(torch.cuda.Stream(device=dev))
    for _ in range(int(os.environ.get("SPACING", "1")) - 1):      # skip streams of torch's pool: which hardware queues the frames' streams share
        torch.cuda.Stream(device=dev)
deal = int(os.environ.get("DEAL", "1")) if world > 1 else 0
st = H.make_settings(1920, 1080, spp=spp, tile_rank=int(os.environ.get("RANK_", "2")) % world, tile_world=world, tile_deal=deal)


def run(n_streams):
    for f in range(2 * n_streams):
        w.render_device(st, bufs[f % n_streams].data_ptr(), stream=streams[f % n_streams].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):
        w.render_device(st, bufs[f % n_streams].data_ptr(), stream=streams[f % n_streams].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3


for rep in range(2):
    t = [run(n) for n in NS]
    print("world %d spp %d %s: ms/frame with %s frames in flight: %s" % (world, spp, cfg, NS, " ".join("%.3f" % x for x in t)))
for b in bufs[1:]:
    assert torch.equal(bufs[0], b)
