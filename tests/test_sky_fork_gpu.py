"""Where the sky-only kernel runs (rrt_api.cpp: side stream forked off behind the primary kernel; RRT_SKY_AFTER = 0 / 2 fork
it in front of the primary kernel / behind the first bounce level, RRT_SKY_FORK = 0 keeps it on the frame's own stream)
is scheduling only: every placement must give the same bytes.  The switches are read once per process, so each
placement renders in a process of its own; the frame is the main.rs world at 1080p (tile lists need 2^20 path samples),
three samples in two batches, so that the second batch's sky kernel meets the first batch's accumulate pass.  The
default placement is pinned against the oracle by the other tests of the suite."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import torch  # noqa: F401  (brings the process's HIP runtime)
from rustraytracer_amd import host as H, scenes
g = scenes.main_rs_scene(H.World(device=0), H.loaders)
h = hashlib.sha256()
for kw in (dict(spp=3, seed=7, batch_spp=2), dict(spp=1, seed=8)):
    out = g.render(1920, 1080, want=("rgb", "ids", "t"), **kw)
    for k in ("rgb", "ids", "t"):
        h.update(np.ascontiguousarray(out[k]).tobytes())
    assert g.stats()["n_batches"] == (2 if kw["spp"] == 3 else 1)
print("DIGEST", h.hexdigest())
""" % ROOT


def _digest(env_extra):
    env = dict(os.environ)
    for k in ("RRT_SKY_AFTER", "RRT_SKY_FORK"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST ")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0].split()[1]


def test_every_placement_of_the_sky_kernel_gives_the_same_frame():
    want = _digest({})
    assert len(want) == len(hashlib.sha256().hexdigest())
    for extra in ({"RRT_SKY_AFTER": "0"}, {"RRT_SKY_AFTER": "2"}, {"RRT_SKY_FORK": "0"}):
        assert _digest(extra) == want, extra


def test_sky_stream_setter_changes_nothing_but_the_schedule():
    """rrt_set_sky_stream (what a pipelining caller switches at run time, TilePipeline.autotune): the same frames with
    the sky-only kernel on the side stream and on the frame's own stream, alternating on one scene and with frames of
    both kinds in flight on two streams."""
    import numpy as np
    import torch
    from rustraytracer_amd import host as H, scenes
    g = scenes.main_rs_scene(H.World(device=0), H.loaders)
    want = g.render(1920, 1080, spp=2, seed=11, want=("rgb", "ids", "t"))
    for mode in (0, 1, 0):
        g.set_sky_stream(mode)
        got = g.render(1920, 1080, spp=2, seed=11, want=("rgb", "ids", "t"))
        for k in ("rgb", "ids", "t"):
            assert np.array_equal(got[k], want[k]), (mode, k)
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
    st = H.make_settings(1920, 1080, spp=2, seed=11)
    torch.cuda.synchronize()
    for i in range(4):
        g.set_sky_stream(i & 1)
        g.render_device(st, outs[i].data_ptr(), stream=streams[i % 2].cuda_stream)
    torch.cuda.synchronize()
    for i in range(4):
        assert np.array_equal(outs[i].cpu().numpy(), want["rgb"]), i
    with pytest.raises(Exception, match="sky stream"):
        g.set_sky_stream(2)
    g.set_sky_stream(1)


def test_tile_pipeline_autotune_keeps_one_placement_and_the_frame():
    """TilePipeline.autotune times both placements over the pipeline's own frames in flight and leaves the scene on one
    of them.  One process stands for rank 1 of 2 (no process group: the gather copies this rank's tiles into slot 0 of
    the gathered buffer, the other rank's stay zero), so what the pipeline assembles must be the frame's tiles of rank 1
    where the de-interleave puts rank 0's -- after the tuning and for the next frame."""
    import numpy as np
    import torch
    from rustraytracer_amd import host as H, scenes, distributed as D, tiles
    g = scenes.main_rs_scene(H.World(device=0), H.loaders)
    W, Hh, world, rank = 1920, 1080, 2, 1
    full = g.render(W, Hh, spp=1, seed=2, want=("rgb",))["rgb"]
    gathered = np.zeros((world,) + tiles.extract_rank_tiles(full, rank, world).shape, np.uint8)
    gathered[0] = tiles.extract_rank_tiles(full, rank, world)
    want = tiles.assemble_numpy(gathered, W, Hh, world)
    assert want.any()
    dev = torch.device("cuda", 0)
    frame = torch.zeros((Hh, W, 3), dtype=torch.uint8, device=dev)
    pipe = D.TilePipeline(g, W, Hh, world, dev, frame, depth=3)
    st = H.make_settings(W, Hh, spp=1, seed=2, tile_rank=rank, tile_world=world)
    mode = pipe.autotune(st, reps=1)
    assert mode in (0, 1)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), want)
    frame.zero_()
    pipe.step(st)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), want)
    g.set_sky_stream(1)
