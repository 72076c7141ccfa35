"""rrt_render (the blocking call the reference's `render` makes, src/tracer.rs:77) into pageable and into page-locked
host memory (banded resolve + copies under it), and rrt_release_stream."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pinned_array(B, shape, dtype=np.uint8):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = B.lib.rrt_alloc_host_pinned(n)
    assert p
    buf = (C.c_ubyte * n).from_address(p)
    return np.frombuffer(buf, dtype=dtype).reshape(shape), p


@pytest.mark.parametrize("w,h,spp", [(640, 360, 2), (1920, 1080, 16), (333, 77, 3), (64, 8, 1)])
def test_render_into_pinned_memory_equals_pageable(H, O, scenes, w, h, spp):
    from rustraytracer_amd import binding as B
    g = scenes.main_rs_scene(H.World(device=0), H.loaders, skybox="sky")
    cam = H.default_camera()
    st = H.make_settings(w, h, spp=spp, seed=5)
    page = np.zeros((h, w, 3), np.uint8)
    B.check(B.lib.rrt_render(g.h, C.byref(cam), C.byref(st), page.ctypes.data, None, None, None))
    pin, p = _pinned_array(B, (h, w, 3))
    ids = np.full((h, w, 2), -2, np.int32)
    col = np.zeros((h, w, 3))
    try:
        for _ in range(3):                                  # repeated calls reuse streams, events and buffers
            pin[:] = 0
            B.check(B.lib.rrt_render(g.h, C.byref(cam), C.byref(st), pin.ctypes.data, None, None, None))
            assert np.array_equal(pin, page)
        # AOVs ride along (pageable) with a pinned RGB8 frame
        pin[:] = 0
        B.check(B.lib.rrt_render(g.h, C.byref(cam), C.byref(st), pin.ctypes.data, ids.ctypes.data, None, col.ctypes.data))
        assert np.array_equal(pin, page)
    finally:
        del pin
        B.lib.rrt_free_host_pinned(p)
    if w * h <= 640 * 360:
        c = scenes.main_rs_scene(O.OracleScene(), O, skybox="sky").render(w, h, spp=spp, seed=5)
        assert np.array_equal(page, c["rgb"]) and np.array_equal(ids, c["ids"]) and np.array_equal(col, c["color"])


def test_release_stream_frees_a_work_set(H, scenes):
    """Nine streams in a row would exhaust the eight work sets; releasing each before the next does not, and the
    frames stay identical.  An unknown stream is RRT_E_INVALID."""
    import torch
    from rustraytracer_amd import binding as B
    g = scenes.main_rs_scene(H.World(device=0), H.loaders, skybox="sky")
    g.commit()
    st = H.make_settings(320, 180, spp=2)
    dev = torch.device("cuda", 0)
    ref = torch.empty((180, 320, 3), dtype=torch.uint8, device=dev)
    g.render_device(st, ref.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(12):
        s = torch.cuda.Stream(device=dev)
        out = torch.zeros_like(ref)
        g.render_device(st, out.data_ptr(), stream=s.cuda_stream)
        B.check(B.lib.rrt_release_stream(g.h, s.cuda_stream))       # waits for the frame, then frees the set
        assert torch.equal(out, ref)
    s = torch.cuda.Stream(device=dev)
    assert B.lib.rrt_release_stream(g.h, s.cuda_stream) == -1
