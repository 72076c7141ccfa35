"""Multi-GPU path: one process per GPU, scene replicated, image tiles dealt
round-robin (tiles.py), ONE exchange step per frame -- an all-gather of the
ranks' compact RGB8 tile buffers (RCCL over xGMI with backend "nccl"; "gloo"
in the CPU tests) -- then a de-interleave to row-major.  The reference has no
counterpart (single process, 12 threads over contiguous pixel chunks,
src/tracer.rs:117-158); pixels are independent there too (tracer.rs:150).

Message size per rank = slots * 768 B: 1080p on 8 GPUs -> 1013 tiles -> 0.78 MB,
latency-bound over xGMI, so no bucketing or ring tuning is needed.
"""
import torch
import torch.distributed as dist

from . import tiles


def alloc_tile_buffers(width, height, world, device, dealt=False):
    """A rank's compact tile buffer and the gathered one.  dealt (rrt_settings.tile_deal): flat bytes -- the slots'
    pixels followed by the rank's int32 slot -> tile table (tiles.dealt_buffer_bytes)."""
    slots = tiles.tile_slots(width, height, world)
    if dealt and world > 1:
        local = torch.zeros((tiles.dealt_buffer_bytes(width, height, world),), dtype=torch.uint8, device=device)
    else:
        local = torch.zeros((slots, tiles.TILE_H, tiles.TILE_W, 3), dtype=torch.uint8, device=device)
    gathered = torch.zeros((world,) + tuple(local.shape), dtype=torch.uint8, device=device)
    return local, gathered


def _host_staged():
    """gloo moves host memory: device tensors go through a host copy (rehearsals of the N-rank path
    on one GPU, RRT_BENCH_BACKEND=gloo; the production backend is "nccl" = RCCL, device to device)."""
    return dist.get_backend() == "gloo"


class NativeComm:
    """The library's own RCCL communicator (include/rrt.h: rrt_comm_*), the path a host without Python uses.
    The 128-byte id travels over the already initialised torch.distributed group here; a C++ / Rust host would
    hand it over by file, environment or MPI."""

    def __init__(self, rank, world, device):
        import ctypes as C
        from . import binding as B
        self._B, self._C = B, C
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            B.check(B.lib.rrt_comm_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            staged = ident if _host_staged() else ident.to(device)
            dist.broadcast(staged, src=0)
            ident = staged.cpu()
        raw = (C.c_uint8 * 128)(*ident.tolist())
        self.h = C.c_void_p()
        B.check(B.lib.rrt_set_device(device.index if device.index is not None else 0))
        B.check(B.lib.rrt_comm_init(rank, world, raw, C.byref(self.h)))

    def gather(self, local, gathered, width, height, root=-1, stream=None, dealt=False):
        fn = self._B.lib.rrt_gather_tiles_dealt if dealt else self._B.lib.rrt_gather_tiles
        self._B.check(fn(self.h, local.data_ptr(), gathered.data_ptr(), width, height, root, stream))

    def close(self):
        if getattr(self, "h", None):
            self._B.lib.rrt_comm_destroy(self.h)
            self.h = None


def gather_tiles(local, gathered):
    """The frame's only collective.  Every rank ends with every rank's tiles."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # output viewed as the concatenation along dim 0: the one layout both RCCL and gloo accept
        flat = gathered.view((-1,) + tuple(local.shape[1:]))
        if local.is_cuda and _host_staged():
            h_out = torch.empty(flat.shape, dtype=flat.dtype)
            dist.all_gather_into_tensor(h_out, local.cpu())
            flat.copy_(h_out)
        else:
            dist.all_gather_into_tensor(flat, local)
    else:
        gathered[0].copy_(local)
    return gathered


class TilePipeline:
    """One frame per step() on this rank: render its tiles, all-gather, de-interleave -- with `depth` frames in
    flight.  Frame i renders on stream i % depth into tile buffer i % depth (the scene keeps one set of work buffers
    per stream, so renders on different streams overlap: the thin tail of one frame's launch chain runs under the
    next frame's primary kernel); the gather and the de-interleave of the frames run, in frame order, on one
    communication stream behind an event; render i waits for gather i - depth to have read its tile buffer."""

    def __init__(self, scene, width, height, world, device, frame, native=None, depth=3, dealt=False):
        from . import host as H
        self.native = native          # a NativeComm: gather to rank 0 through librrt.so's own RCCL calls
        self._H = H
        self.scene, self.w, self.h, self.world, self.frame = scene, width, height, world, frame
        self.depth = depth
        self.dealt = bool(dealt) and world > 1      # the settings passed to step() must carry tile_deal = 1 then
        self.local, self.gathered = [], []
        for _ in range(depth):
            lo, ga = alloc_tile_buffers(width, height, world, device, self.dealt)
            self.local.append(lo)
            self.gathered.append(ga)
        self.s_render = [torch.cuda.Stream(device=device) for _ in range(depth)]
        self.s_comm = torch.cuda.Stream(device=device)
        self.ev_render = [torch.cuda.Event() for _ in range(depth)]
        self.ev_comm = [torch.cuda.Event() for _ in range(depth)]
        self.i = 0

    def step(self, settings):
        k = self.i % self.depth
        sr = self.s_render[k]
        if self.i >= self.depth:
            sr.wait_event(self.ev_comm[k])                   # frame i-depth's gather has read local[k]
        self.scene.render_device(settings, self.local[k].data_ptr(), stream=sr.cuda_stream)
        self.ev_render[k].record(sr)
        with torch.cuda.stream(self.s_comm):
            self.s_comm.wait_event(self.ev_render[k])
            if self.native is not None:
                self.native.gather(self.local[k], self.gathered[k], self.w, self.h, root=-1, stream=self.s_comm.cuda_stream, dealt=self.dealt)
            else:
                gather_tiles(self.local[k], self.gathered[k])
            self._H.assemble_tiles(self.gathered[k].data_ptr(), self.w, self.h, self.world, self.frame.data_ptr(),
                                   stream=self.s_comm.cuda_stream, dealt=self.dealt)
            self.ev_comm[k].record(self.s_comm)
        self.i += 1

    def autotune(self, settings, reps=3):
        """Picks where the sky-only tiles' kernel runs (World.set_sky_stream) for this pipeline: both placements are
        timed over 3 * depth frames, `reps` times, and the faster one stays (the side stream unless the frame's own is
        more than 1 % ahead).  With frames in flight the better one depends on which of the process's streams share a
        hardware queue -- an RCCL communicator created before the first frame is enough to turn it round -- so it is
        measured, once, before the frames that count.  Every rank takes the same decision (slowest rank's times).
        Returns the mode chosen."""
        import time
        best = {}
        for mode in (1, 0):
            self.scene.set_sky_stream(mode)
            for _ in range(2 * self.depth):
                self.step(settings)
            ts = []
            for _ in range(reps):
                if dist.is_initialized() and dist.get_world_size() > 1:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3 * self.depth):
                    self.step(settings)
                torch.cuda.synchronize()
                ts.append(max_over_ranks(time.perf_counter() - t0, self.frame.device))
            best[mode] = min(ts)
        mode = 1 if best[1] <= best[0] * 1.01 else 0
        self.scene.set_sky_stream(mode)
        return mode


def max_over_ranks(seconds, device):
    multi = dist.is_initialized() and dist.get_world_size() > 1
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if multi and _host_staged() else device)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    multi = dist.is_initialized() and dist.get_world_size() > 1
    t = torch.tensor(list(values), dtype=torch.int64, device="cpu" if multi and _host_staged() else device)
    if multi:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]
