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
