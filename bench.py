#!/usr/bin/env python3
"""Benchmark of the tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path (primary rays -> closest hit ->
shading -> reflection bounces -> sky lookup -> accumulate -> quantise, and for
N > 1 the framebuffer gather over RCCL + de-interleave) over one frame.  The
scene is resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

N = 1 (the contract's bench line): the frame BASELINE.json's metric and north_star's
target are quoted on -- the main.rs world (ufo_fix.obj kd-tree + 2 spheres) under the
blue_sunset cubemap at 1920x1080, 64 spp -- THE SAME FRAME every `--gpus N` line
renders, so value(N) / value(1) is a true strong-scaling ratio.  BASELINE configs[1]'s
own sampling (16 spp) is measured in the same run and rides along as `configs1_16spp`.
`--config cfg5` runs configs[4] (2M-triangle mesh) with its own roofline and
cpu_baseline; the other configs print the same line without them.

N > 1: without WORLD_SIZE in the environment the script starts its N ranks
itself (a child `python -m torch.distributed.run`, before anything touches the
GPU); under torch.distributed.run it is one rank.  Default = STRONG scaling on
north_star's target frame: the same scene at 1920x1080, 64 spp IN TOTAL whatever
N is; the image's 32x8 tiles are dealt round-robin over the ranks, each renders
its tiles at 64 spp, one all-gather of the compact RGB8 tile buffers (6.2 MB in
total) and a de-interleave end the step; the gather of frame i overlaps the
render of the next frames (own stream, one tile buffer per frame in flight).  The weak
figure (16*N spp, one 1080p/16spp frame of samples per GPU) is measured in the
same run and printed beside it (`weak`); `--scaling weak` swaps the two.

Metric: Mrays/s, whole job.  A ray = one world-level closest-hit query (one
`RayTracer::cast`, src/tracer.rs:177) actually executed on the GPU, with
reference semantics (prune_zero_mix = 0: the reflection ray is cast even when
its mix is 0.0, as src/tracer.rs:195-201 does).

Frames in flight: steps are enqueued round-robin on `--in-flight` streams (default 4, 3 for frames under a million paths; the scene keeps one set of work
buffers per stream), so the thin tail of one frame's launch chain (bounce levels, accumulate) runs under the next
frame's primary kernel.  Every step is a complete frame; `value` is the throughput of K of them.  The figure with one
frame at a time (the latency of a frame) is printed beside it (`one_frame_at_a_time`), and the per-kernel durations
of the roofline come from that sequential pass.

Timing: W warm-up steps, then blocks of exactly K steps, each bracketed by a
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over the
ranks; blocks repeat until 0.5 s of them have been timed.  `value` and
`ms_per_step` come from the MEDIAN block; `timing` carries min / median / max.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP = 1920, 1080, 16
STRONG_SPP = 64                      # north_star: "ufo.obj kd-tree at 1080p/64spp"

# --config: name -> (scene function, W, H, spp, max_bounce, what)
CONFIGS = {
    "cfg1": ("cfg1_scene", 640, 480, 1, 2, "configs[0]: cube.obj + 3 spheres, sky cubemap, 640x480, 1 spp"),
    "cfg2": ("cfg2_scene", W, H, SPP, 2, "configs[1]"),
    "cfg3": ("cfg3_scene", 1920, 1080, 1, 8, "configs[2]: reflect/refract glass-sphere scene, depth 8, 1920x1080, 1 spp (extension lobes)"),
    "cfg4": ("cfg4_scene", 3840, 2160, 256, 2, "configs[3]: ufo_fix.obj with stochastic diffuse bounces, 3840x2160, 256 spp (extension lobes)"),
    "cfg2scene": ("cfg2_as_scene", W, H, SPP, 2, "configs[1] with main.rs:57-59 enabled: the three entities under one Scene<T>"),
    "many": ("many_entity_scene", 1920, 1080, 4, 4, "63 entities (ufo, cube, 60 matte/mirror/glass spheres) under one Scene<T> kd-tree + 1 flat sphere, 1920x1080, 4 spp"),
    "manyflat": ("many_entity_flat", 1920, 1080, 4, 4, "the same 63 entities in the flat entity list (no Scene<T>), 1920x1080, 4 spp"),
    "cfg5": ("cfg5_scene", 1920, 1080, 64, 2, "configs[4]: synthetic 2M-triangle height field, diffuse bounces, 1920x1080, 64 spp (extension lobes)"),
}

# Algorithmic work per ray for the rooflines -- the constants frozen in BASELINE.md section 3.
# Bytes are sizes of the device layout (csrc/rrt_device.h); flops are SURVEY 8(d)'s.
S_NODE, S_LEAF, S_BOX32, S_TRI, S_WIN = 16, 128, 24, 72, 32   # DNodeI / DLeaf / DBox32 / DTri / winner's DRef + normal
F_TRI, F_BOX, F_NODE, F_SHADE, F_SPHERE = 53, 30, 1, 120, 25
B_SKY, B_CONT, B_DEAD = 7, 116, 24   # HBM I/O of a primary ray: ends in the sky / continues / extra when its reflection is dead
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F64_TFLOPS = 78.6      # vector FP64, FMA-counted (this path may not contract FMAs: its own ceiling is half of it)
MIN_TIMED_SECONDS = 0.5


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: which figure is `value` (the other is printed beside it)")
    ap.add_argument("--prune", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rank-share", action="store_true", help="skip the N = 1 strong-scaling rehearsal (tile_world 8 on one GPU)")
    ap.add_argument("--once", action="store_true", help="time ONE block of --steps (profiler runs)")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--batch-spp", type=int, default=0, help="samples per pixel per wavefront batch (0 = from the work budget)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2")
    ap.add_argument("--spp", type=int, default=0, help="override the config's samples per pixel")
    ap.add_argument("--budget-gib", type=float, default=0.0, help="work-buffer budget in GiB (0 = library default)")
    ap.add_argument("--cull-mode", type=int, default=-1, help="rrt_set_cull_mode (0 off, 1 on, 2 verify; -1 = library default)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("RRT_FRAMES_IN_FLIGHT", "0")),
                    help="frames in flight (one stream and one set of work buffers each); 1 = one frame at a time")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` on its own: start the N ranks as a fresh child process (never exec: this
    process may not replace itself once a GPU runtime is loaded, and nothing has touched the GPU yet)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def median(v):
    s = sorted(v)
    n = len(s)
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


# ---------------------------------------------------------------------------
# CPU leg: times the oracle (a port of the reference, oracle/) on a bounded sample of the same
# workload and counts the per-primary-ray traversal work that prices the rooflines.
# ---------------------------------------------------------------------------
def cpu_leg(config, n_threads, w, h, spp, max_bounce):
    import ctypes as C
    import numpy as np
    from oracle import oracle as O
    from rustraytracer_amd import scenes
    L = O.lib()
    t0 = time.perf_counter()
    cpu = getattr(scenes, CONFIGS[config][0])(O.OracleScene(), O)
    build_s = time.perf_counter() - t0
    # a strided-row sample sized for ~2-4 s of wall time on n_threads cores (30-60 core-seconds)
    cpu.render(w, h, spp=1, max_bounce=max_bounce, n_threads=n_threads, y0=0, y1=32, want=())            # warm
    t0 = time.perf_counter()
    probe = cpu.render(w, h, spp=1, max_bounce=max_bounce, n_threads=n_threads, y0=5, row_stride=54, want=())   # 20 rows
    rate = int(probe["counters"][0]) / (time.perf_counter() - t0)        # rays/s, rough
    rays_frame = int(probe["counters"][0]) / 20.0 * h * spp
    # cfg2: every (spp/16)-th row of the frame = the rays of one 16-spp frame; else ~3 s of work
    stride = max(1, spp // 16) if config == "cfg2" else max(1, int(rays_frame / (rate * 3.0)))
    dt, frames, c = 0.0, 0, None
    while dt < 2.0 and frames < 16:
        t0 = time.perf_counter()
        out = cpu.render(w, h, spp=spp, max_bounce=max_bounce, seed=0, n_threads=n_threads, y0=3, row_stride=stride, want=())
        dt += time.perf_counter() - t0
        frames += 1
        c = dict(zip(O.COUNTER_NAMES, (int(x) for x in out["counters"])))
    rows = len(range(3, h, stride))
    base = {
        "value": c["rays"] * frames / dt / 1e6, "unit": "Mrays/s", "cores": n_threads, "kind": "port",
        "sample": "rows 3..%d step %d of the %dx%d frame (%d rows), %d spp, rendered %d times: %d rays in %.2f s; "
                  "C++ f64 restatement of the reference, rows handed to threads dynamically (no per-hit Material "
                  "clone: optimistic for the Rust binary); scene + kd-tree build %.1f s, not timed"
                  % (h - 1, stride, w, h, rows, spp, frames, c["rays"] * frames, dt, build_s),
    }
    if config == "cfg2":
        # the reference's own work split (SURVEY 8d "faithful" mode): 12 threads over 12 contiguous row-major
        # chunks of W*H/12 pixels (tracer.rs:117-158) -- the slowest chunk sets the time
        L.orc_set_static_chunks(1)
        spp_f = min(spp, 16)             # bounded: a quarter of the 64 samples (the split, not the sampling, is the point)
        t0 = time.perf_counter()
        outf = cpu.render(w, h, spp=spp_f, seed=0, n_threads=12, want=())
        dtf = time.perf_counter() - t0
        L.orc_set_static_chunks(0)
        base["reference_work_split"] = {
            "value": int(outf["counters"][0]) / dtf / 1e6, "unit": "Mrays/s", "threads": 12,
            "sample": "one full %dx%d frame at %d spp in 12 static contiguous pixel chunks as tracer.rs:117-158: "
                      "%d rays in %.2f s" % (w, h, spp_f, int(outf["counters"][0]), dtf)}
    # per-primary-ray traversal work: closest hit of the sample-0 primary rays of every 32nd row
    st = O.Settings(w, h, 1, max_bounce, 0, 0, 1)
    cam = O.default_camera()
    ys = np.ascontiguousarray(list(range(3, h, 32)), dtype=np.int32)          # 34 rows of a 1080-row frame
    rays = np.zeros((len(ys) * w, 6))
    L.orc_primary_rays_rows(C.byref(cam), C.byref(st), ys.ctypes.data, len(ys), 0, rays.ctypes.data)
    L.orc_set_count_box_hits(1)
    tr = cpu.trace_rays(rays)
    L.orc_set_count_box_hits(0)
    pc = dict(zip(O.COUNTER_NAMES, (int(x) for x in tr["counters"])))
    n = float(len(rays))
    per = {k: pc[k] / n for k in ("node_steps", "leaf_visits", "tri_tests", "sphere_tests", "box_tests", "tri_box_hits")}
    per["hit_fraction"] = float((tr["ent"] >= 0).sum()) / n
    per["mesh_hit_fraction"] = float((tr["tri"] >= 0).sum()) / n
    per["sampled_primary_rays"] = int(n)
    return base, per


def flops_per_ray(per):
    """N_tri*53 + N_box*30 + N_node*1 + N_sphere*25 + 120 (the reference's arithmetic:
    every triangle of a visited leaf is tested; what the cull saves is not subtracted)."""
    return (per["tri_tests"] * F_TRI + per["box_tests"] * F_BOX + per["node_steps"] * F_NODE
            + per["sphere_tests"] * F_SPHERE + F_SHADE)


def io_bytes_per_ray(stats, n_rays=None):
    """HBM I/O a primary ray cannot avoid: 3 B texel in + 4 B compact sample out when
    it ends in the sky; 56 B level record + 56 B queue entry + 4 B sample tag when the path continues, + 24 B
    advance sample when the continuation ray is dead (mix 0.0).  n_rays: the primary rays of the launch that is priced
    (with tile lists the primary kernel holds the rays of the heavy tiles only; every continuing ray is among them)."""
    lv = stats["level_rays"]
    n = max(n_rays if n_rays else lv[0], 1)
    cont = lv[1] / n
    dead = stats["zero_mix_paths"] / n
    return (1.0 - cont) * B_SKY + cont * B_CONT + dead * B_DEAD, cont, dead


def traversal_bytes_per_ray(per):
    """descent steps x 16 B + leaves x 128 B + listed triangles x 24 B (the cull box the
    kernel reads in place of the triangle) + triangles whose own box the ray crosses x 72 B (the exact test no
    box cull can avoid) + 32 B for a mesh hit (triangle id, rank + normal)."""
    return (per["node_steps"] * S_NODE + per["leaf_visits"] * S_LEAF + per["tri_tests"] * S_BOX32
            + per["tri_box_hits"] * S_TRI + per["mesh_hit_fraction"] * S_WIN)


def attach_profiles(roof, config):
    """PMC traffic / SQ issue figures of the previous collection (tools/collect_all.sh -> profiles/)."""
    sfx = "" if config == "cfg2" else "_" + config
    tf = os.path.join(ROOT, "profiles", "pmc_traffic%s.json" % sfx)
    traffic = None
    if os.path.exists(tf):
        try:
            t = json.load(open(tf)).get("k_wave_primary")
            if t:
                traffic = t["hbm_bytes_per_launch"]
                roof["traffic_source"] = "profiles/pmc_traffic%s.json: %s" % (sfx, t.get("source"))
        except (ValueError, KeyError):
            pass
    vf = os.path.join(ROOT, "profiles", "valu_busy%s.json" % sfx)
    if os.path.exists(vf):
        try:
            v = json.load(open(vf)).get("k_wave_primary")
            if v:
                roof["executed"] = {"valu_busy": v["valu_busy"], "lane_utilisation": v["lane_utilisation"],
                                    "issue_slot_frac": v["valu_busy"] * (v["lane_utilisation"] or 0.0),
                                    "waves_per_simd": v["waves_per_simd"], "avg_launch_ns": v["avg_launch_ns"],
                                    "source": "profiles/valu_busy%s.json: %s" % (sfx, v.get("source"))}
                # what the kernel EXECUTES, beside the delivered-work `frac`: the time its VALU wave-instructions
                # need at one issue per SIMD every 4 cycles, against the launch they were counted in
                if v.get("valu_wave_insts"):
                    roof["issue_floor_ms"] = v["valu_wave_insts"] * 4.0 / (1024 * 2.4e9) * 1e3
                    roof["launch_ms"] = v["avg_launch_ns"] * 1e-6
                    roof["issue_floor_frac"] = roof["issue_floor_ms"] / roof["launch_ms"]
        except (ValueError, KeyError):
            pass
    gf = os.path.join(ROOT, "profiles", "ta_busy%s.json" % sfx)
    if os.path.exists(gf):
        try:
            g = json.load(open(gf)).get("k_wave_primary")
            if g:
                roof["gather"] = g
        except (ValueError, KeyError):
            pass
    return traffic


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    import torch                          # first: brings the process's HIP runtime
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    # Rehearsal of the N-rank path on a box with fewer GPUs (never used by the driver):
    # RRT_BENCH_DEVICE pins every rank to one device, RRT_BENCH_BACKEND=gloo stages the gather through the host.
    backend = os.environ.get("RRT_BENCH_BACKEND", "nccl")
    if "RRT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["RRT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from rustraytracer_amd import host as Hst, scenes, distributed as D
    scene_fn, w, h, spp0, max_bounce, what = CONFIGS[args.config]
    side16_spp = 0
    if args.spp > 0:
        spp0 = args.spp
    elif args.config == "cfg2":
        # the bench frame = north_star's target (1080p, 64 spp), the same at every N; configs[1]'s 16 spp rides along
        side16_spp, spp0 = spp0, STRONG_SPP
    scene = getattr(scenes, scene_fn)(Hst.World(device=local_rank), Hst.loaders)
    if args.budget_gib > 0:
        scene.set_work_budget(int(args.budget_gib * (1 << 30)))
    if args.cull_mode >= 0:
        scene.set_cull_mode(args.cull_mode)
    scene.commit()                                   # scene resident in HBM before timing

    frame = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    if args.in_flight <= 0:
        # 4 frames in flight; 3 for frames of under a million paths, whose launches the host cannot feed to four
        # streams any faster (cube + 3 spheres, 640x480: 0.038 ms a frame with 3 in flight, 0.050 with 4)
        args.in_flight = 4 if w * h * spp0 >= (1 << 20) else 3
    depth = max(1, min(args.in_flight, 4))
    frames = [frame] + [torch.empty_like(frame) for _ in range(depth - 1)] if world == 1 else [frame]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
    # RRT_BENCH_GATHER=native: the gather goes through librrt.so's own RCCL communicator (rrt_comm_*, the path of
    # a host without Python) instead of torch.distributed's; torch's group still carries the id and the barriers
    native = D.NativeComm(rank, world, dev) if world > 1 and os.environ.get("RRT_BENCH_GATHER", "torch") == "native" else None
    # N > 1: RRT_BENCH_DEAL=1 deals the tiles to the ranks by cost class (rrt_settings.tile_deal) instead of by position:
    # measured on this frame it evens out the ranks (0.553-0.580 ms against 0.499-0.560) without making the slowest
    # one faster, so the positional map stays the default
    deal = 1 if os.environ.get("RRT_BENCH_DEAL", "0") != "0" else 0
    pipe = D.TilePipeline(scene, w, h, world, dev, frame, native=native, depth=depth, dealt=bool(deal)) if world > 1 else None
    # no hipEvents between the launches of the timed frames (the sequential pass for the roofline turns them on; with
    # --in-flight 1 the timed pass is that sequential pass)
    scene.set_timing(2 if world == 1 and depth == 1 else 0)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Where the sky-only tiles' kernel runs (rrt_set_sky_stream): on a side stream behind the primary kernel, or on the
    # frame's own stream.  One frame at a time the side stream always wins.  With frames in flight it depends on which
    # of the process's streams share a hardware queue (the runtime deals every stream ever created to four of them):
    # measured on one MI355X, the side stream is 4-18 % ahead in a process of its own and as much behind once an RCCL
    # communicator exists.  So a pipelining caller tries both, untimed, before its warm-up -- every rank takes the same
    # decision from the slowest rank's times.
    sky_mode = [1]
    sky_picks = []

    def pick_sky_stream(step, in_flight, reps=3):
        if in_flight <= 1 or os.environ.get("RRT_BENCH_SKY_STREAM", "") in ("0", "1"):
            mode = int(os.environ.get("RRT_BENCH_SKY_STREAM", "1") or 1) if in_flight > 1 else 1
            scene.set_sky_stream(mode)
            return mode
        best = {}
        for mode in (1, 0):
            scene.set_sky_stream(mode)
            for _ in range(2 * in_flight):
                step()
            ts = []
            for _ in range(reps):
                sync()
                t0 = time.perf_counter()
                for _ in range(3 * in_flight):
                    step()
                torch.cuda.synchronize()
                ts.append(D.max_over_ranks(time.perf_counter() - t0, dev))
            best[mode] = min(ts)
        mode = 1 if best[1] <= best[0] * 1.01 else 0       # the default unless the other is clearly ahead
        sky_picks.append({"in_flight": in_flight, "ms_side_stream": best[1] / (3 * in_flight) * 1e3, "ms_own_stream": best[0] / (3 * in_flight) * 1e3, "chosen": mode})
        scene.set_sky_stream(mode)
        return mode

    def measure(spp_total, in_flight):
        """-> (list of block seconds [max over ranks], stats of this rank's last frame)"""
        st = Hst.make_settings(w, h, spp=spp_total, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune,
                               tile_rank=rank, tile_world=world, batch_spp=args.batch_spp, tile_deal=deal if world > 1 else 0)

        count = [0]

        def step():
            if world == 1:
                k = count[0] % in_flight
                count[0] += 1
                scene.render_device(st, frames[k].data_ptr(), stream=streams[k].cuda_stream)
            else:
                pipe.step(st)

        sky_mode[0] = pick_sky_stream(step, in_flight)
        for _ in range(max(args.warmup, in_flight)):
            step()
        blocks, total = [], 0.0
        while True:
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()
            dt = D.max_over_ranks(time.perf_counter() - t0, dev)     # the same value on every rank: the loop ends together
            blocks.append(dt)
            total += dt
            if args.once or total >= MIN_TIMED_SECONDS or len(blocks) >= 500:
                break
        return blocks, scene.stats()

    def figure(spp_total, in_flight=depth):
        blocks, stats = measure(spp_total, in_flight)
        rays_step, paths_step, overflow = D.sum_over_ranks(
            [stats["rays_traced"], stats["paths"], stats["traversal_overflow"]], dev)
        med = median(blocks)
        return {"value": rays_step * args.steps / med / 1e6, "ms_per_step": med / args.steps * 1e3, "spp_total": spp_total,
                "sky_stream": sky_mode[0],
                "rays_per_step": rays_step, "samples_per_step": paths_step, "traversal_overflow": overflow,
                "timing": {"blocks": len(blocks), "steps_per_block": args.steps,
                           "ms_per_step_min": min(blocks) / args.steps * 1e3, "ms_per_step_median": med / args.steps * 1e3,
                           "ms_per_step_max": max(blocks) / args.steps * 1e3}}, stats

    other = None
    sequential = None
    side16 = None
    if world == 1:
        if side16_spp and not args.once:
            # BASELINE configs[1] as written (16 spp): in flight and one frame at a time, with its kernel times
            f16, _ = figure(side16_spp)
            scene.set_timing(2)
            s16, st16 = figure(side16_spp, 1)
            scene.set_timing(0)
            side16 = {"what": "BASELINE configs[1] at its own sampling: the same scene and frame at %d spp" % side16_spp,
                      "value": f16["value"], "unit": "Mrays/s", "ms_per_step": f16["ms_per_step"], "spp_total": side16_spp,
                      "rays_per_step": f16["rays_per_step"], "timing": f16["timing"],
                      "one_frame_at_a_time": {"value": s16["value"], "ms_per_step": s16["ms_per_step"]},
                      "kernel_ms": {"k_wave_primary": st16["primary_kernel_ms"], "k_wave_bounce": st16["bounce_kernel_ms"],
                                    "gpu": st16["gpu_ms"]}}
        main_fig, stats = figure(spp0)
        scaling = "strong"                    # one GPU: the distinction is empty; N > 1 defaults to strong
        if depth > 1:
            scene.set_timing(2)               # per-kernel events: this pass feeds the roofline
            sequential, stats = figure(spp0, 1)
            scene.set_timing(0)
    else:
        spp_strong, spp_weak = spp0, (side16_spp or spp0) * world
        strong, stats_s = figure(spp_strong)
        weak, stats_w = figure(spp_weak)
        scaling = args.scaling
        main_fig, stats, other = (strong, stats_s, weak) if scaling == "strong" else (weak, stats_w, strong)

    # ---- N = 1 extras -----------------------------------------------------------------------------------
    host_ms = None
    if world == 1:
        # PCIe-inclusive variant (never `value`): rrt_render into host memory = the same frame + one D2H copy
        import numpy as np
        import ctypes as C
        from rustraytracer_amd import binding as B
        rgb = np.zeros((h, w, 3), np.uint8)
        cam = Hst.default_camera()
        n_host = max(2, min(args.steps, 5))
        nbytes = h * w * 3
        pin_p = B.lib.rrt_alloc_host_pinned(nbytes)
        pin = np.frombuffer((C.c_ubyte * nbytes).from_address(pin_p), dtype=np.uint8).reshape(h, w, 3)

        def host_ms_of(spp_h, buf):
            st_h = Hst.make_settings(w, h, spp=spp_h, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune, batch_spp=args.batch_spp)
            for it in range(n_host + 1):
                if it == 1:
                    th = time.perf_counter()
                B.check(B.lib.rrt_render(scene.h, C.byref(cam), C.byref(st_h), buf.ctypes.data_as(C.c_void_p), None, None, None))
            return (time.perf_counter() - th) / n_host * 1e3
        host_ms = host_ms_of(spp0, rgb)                 # pageable destination (a Vec, a numpy array)
        host_pinned_ms = host_ms_of(spp0, pin)          # page-locked destination (rrt_alloc_host_pinned): banded copies under the resolve
        if side16 is not None:
            side16["host_buffer_ms_per_step"] = host_ms_of(side16_spp, rgb)
            side16["host_pinned_ms_per_step"] = host_ms_of(side16_spp, pin)
        del pin
        B.lib.rrt_free_host_pinned(pin_p)
        st = Hst.make_settings(w, h, spp=spp0, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune, batch_spp=args.batch_spp)
        # restore the stats of a device frame (rrt_render overwrote them with identical counters; timings differ)
        scene.set_timing(2)
        scene.render_device(st, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        stats = scene.stats()

    rank_share = None
    if world == 1 and args.config == "cfg2" and not args.no_rank_share and not args.once:
        # Strong-scaling rehearsal on one GPU (no 8-GPU node is guaranteed): the target frame (1080p, 64 spp) as the
        # 8 ranks of tile_world = 8 would each render it, one after the other on this GPU.  max_r T(r) against
        # T(full) / 8 = load balance + fixed per-frame cost, everything but the gather itself.
        def frame_ms(tile_rank, tile_world, outs):
            st = Hst.make_settings(w, h, spp=STRONG_SPP, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune,
                                   tile_rank=tile_rank, tile_world=tile_world, tile_deal=deal if tile_world > 1 else 0)
            count = [0]

            def one():
                scene.render_device(st, outs[count[0] % depth].data_ptr(), stream=streams[count[0] % depth].cuda_stream)
                count[0] += 1
            rank_modes.append(pick_sky_stream(one, depth, reps=2))
            n = 16
            ts = []
            for rep in range(8):      # the first is dropped, the median of seven stays (max over eight ranks of a noisy figure reads high)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    scene.render_device(st, outs[i % depth].data_ptr(), stream=streams[i % depth].cuda_stream)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / n * 1e3)
            return median(ts[1:])
        scene.set_timing(0)      # as the N-rank path runs it
        rank_modes = []
        t_full = frame_ms(0, 1, frames)
        torch.cuda.synchronize()
        full_stats = scene.stats()
        full_rays = int(full_stats["rays_traced"])
        locals_ = [D.alloc_tile_buffers(w, h, 8, dev, dealt=bool(deal))[0] for _ in range(depth)]
        t_rank = [frame_ms(r, 8, locals_) for r in range(8)]
        scene.set_sky_stream(1)
        rank_share = {"tile_world": 8, "tile_deal": deal, "spp_total": STRONG_SPP, "frames_in_flight": depth, "t_full_ms": t_full, "t_rank_ms": t_rank,
                      "sky_stream": rank_modes,      # [whole frame, rank 0 .. 7]: what pick_sky_stream chose for each
                      "rank_share_efficiency": (t_full / 8.0) / max(t_rank),
                      # the same frame as `value` above, re-timed here next to the rank frames (host clock, 16 frames)
                      "strong_frame_one_gpu": {"spp_total": STRONG_SPP, "ms_per_step": t_full, "rays_per_step": full_rays,
                                               "value": full_rays / t_full / 1e3, "unit": "Mrays/s"},
                      "what": "T(full)/8 over max_r T(rank r of 8), each measured alone on this one GPU with the bench's frames in "
                              "flight, host-timed with a device sync around 16 frames (median of 7 such blocks); excludes the RCCL gather (0.78 MB per rank)"}
        # leave the work buffers sized for the bench frame again
        scene.set_timing(2)
        st = Hst.make_settings(w, h, spp=spp0, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune, batch_spp=args.batch_spp)
        scene.render_device(st, frame.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        stats = scene.stats()

    if rank == 0:
        prim_ms, prim_n = stats["primary_kernel_ms"], stats["primary_kernel_launches"]
        bnc_ms, bnc_n = stats["bounce_kernel_ms"], stats["bounce_kernel_launches"]
        spp = main_fig["spp_total"]
        cfg2 = args.config == "cfg2"
        line = {
            "metric": "Mrays/s (node) ufo.obj kd-tree 1080p" if cfg2 else "Mrays/s (node) %s" % args.config,
            "value": main_fig["value"], "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_fig["ms_per_step"],
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic camera rays over the reference's own assets (obj/ufo_fix.obj, cubemaps/blue_sunset)" if cfg2
                    else "synthetic camera rays; scene of rustraytracer_amd/scenes.py:%s" % scene_fn,
            "timing": main_fig["timing"],
            "config": {"workload": ("BASELINE configs[1]: main.rs world (ufo_fix.obj rope kd-tree, 1705 tris, + 2 spheres), "
                                    "blue_sunset cubemap, %dx%d, %d spp in total%s, max_bounce 2, reference ray semantics "
                                    "(prune_zero_mix=%d)" % (w, h, spp, (" (north_star's target frame, the same at every N)" if spp == STRONG_SPP and not args.spp else "")
                                                             if world == 1 or scaling == "strong" else " (16 per GPU)", args.prune))
                       if cfg2 else "BASELINE %s; %d spp in total, max_bounce %d" % (what, spp, max_bounce),
                       "rays_per_step": main_fig["rays_per_step"], "samples_per_step": main_fig["samples_per_step"],
                       "msamples_per_s": main_fig["samples_per_step"] / main_fig["ms_per_step"] / 1e3,
                       "parallelism": "tiles%d" % world, "traversal_overflow": main_fig["traversal_overflow"],
                       "n_batches": stats["n_batches"], "level_rays": stats["level_rays"]},
        }
        line["config"]["frames_in_flight"] = depth
        line["config"]["sky_stream"] = main_fig.get("sky_stream", 1)      # 1 = side stream, 0 = the frame's own (pick_sky_stream)
        line["config"]["sky_stream_picks"] = sky_picks                    # every decision of this run, in order, with both timings
        if sequential is not None:
            line["one_frame_at_a_time"] = {"value": sequential["value"], "ms_per_step": sequential["ms_per_step"],
                                           "timing": sequential["timing"],
                                           "what": "the same K steps with one frame in flight (per-kernel hipEvents on): a frame's latency"}
        if other is not None:
            line["weak" if scaling == "strong" else "strong"] = {
                "value": other["value"], "ms_per_step": other["ms_per_step"], "spp_total": other["spp_total"],
                "rays_per_step": other["rays_per_step"], "timing": other["timing"]}
        if host_ms is not None:
            line["config"]["host_buffer_ms_per_step"] = host_ms      # rrt_render: + D2H of the RGB8 frame, blocking
            line["config"]["host_pinned_ms_per_step"] = host_pinned_ms   # the same into page-locked memory
        if side16 is not None:
            line["configs1_16spp"] = side16
        if rank_share is not None:
            line["rank_share_efficiency"] = rank_share["rank_share_efficiency"]
            line["rank_share"] = rank_share
        if world == 1:
            line["config"]["kernel_ms"] = {"k_wave_primary": prim_ms, "k_wave_bounce": bnc_ms, "k_sky": stats["sky_kernel_ms"], "gpu": stats["gpu_ms"],
                                           "k_wave_primary_launches": prim_n, "k_wave_bounce_launches": bnc_n}
        per = None
        if world == 1 and not args.no_cpu_baseline and args.config in ("cfg2", "cfg5"):
            nthreads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            line["cpu_baseline"], per = cpu_leg(args.config, nthreads, w, h, spp0, max_bounce)
        if world == 1 and args.config in ("cfg2", "cfg5"):
            # Dominant kernel = k_wave_primary (one launch per wavefront batch); live hipEvent timing on the launch stream.
            launch_ms = prim_ms / max(prim_n, 1)
            # With tile lists the primary kernel traces the rays of the heavy tiles (a primary ray can meet an entity
            # there); the sky-only tiles are k_sky's, which runs beside it on a second stream.
            heavy_t, sky_t = stats["heavy_tiles"], stats["sky_tiles"]
            heavy_frac = heavy_t / float(heavy_t + sky_t) if heavy_t + sky_t else 1.0
            rays_per_launch = stats["primary_rays"] * heavy_frac / max(prim_n, 1)
            sky_rays_per_launch = stats["primary_rays"] * (1.0 - heavy_frac) / max(prim_n, 1)
            io_b, cont, dead = io_bytes_per_ray(stats, stats["primary_rays"] * heavy_frac)
            roof = {"kernel": "k_wave_primary", "avg_launch_ms": launch_ms, "launches_per_step": prim_n,
                    "rays_per_launch": rays_per_launch, "continuing_fraction": cont, "zero_mix_fraction": dead,
                    "heavy_tiles": heavy_t, "sky_tiles": sky_t,
                    "k_sky": {"ms_per_step": stats["sky_kernel_ms"], "rays_per_launch": sky_rays_per_launch,
                              "what": "the primary rays of the sky-only tiles (no entity can be met in the tile): texel lookups, "
                                      "on a second stream, forked off behind k_wave_primary (beside the bounce levels); not part of `achieved`"},
                    "other_kernels": {"k_wave_bounce_ms_per_step": bnc_ms, "k_wave_bounce_launches": bnc_n,
                                      "k_wave_primary_ms_per_step": prim_ms, "gpu_ms_per_step": stats["gpu_ms"]}}
            traffic = attach_profiles(roof, args.config)
            secs = launch_ms * 1e-3
            if cfg2:
                # cfg1-4 are NOT HBM-bound (the scene is cache/LDS resident): the bound is the f64 vector
                # ALU.  achieved = the reference's flops per primary ray (oracle counters on the same tree) x rays / time.
                hbm = io_b * rays_per_launch / secs / 1e9
                roof.update({"bound": "valu_f64", "unit": "TFLOP/s", "peak": PEAK_F64_TFLOPS, "traffic": traffic})
                if per:
                    # `per` are means over ALL primary rays of the frame; a ray of a sky-only tile is priced by the same
                    # formula at its two sphere tests, the mesh's root-box test and the shading constant: those rays are
                    # k_sky's, so their flops come off the frame's total before it is put over the primary launch
                    f_sky = 2 * F_SPHERE + 1 * F_BOX + F_SHADE
                    n_all = rays_per_launch + sky_rays_per_launch
                    fl = (flops_per_ray(per) * n_all - f_sky * sky_rays_per_launch) / max(rays_per_launch, 1.0)
                    ach = fl * rays_per_launch / secs / 1e12
                    roof.update({"achieved": ach, "frac": ach / PEAK_F64_TFLOPS, "flops_per_ray": fl, "flops_per_ray_whole_frame": flops_per_ray(per),
                                 "per_primary_ray": per,
                                 "formula": "N_tri*53 + N_box*30 + N_node*1 + N_sphere*25 + 120 (BASELINE.md section 3)",
                                 "note": "work delivered in the reference's flops (the cull skips most of the N_tri exact tests); "
                                         "no-FMA ceiling of this path = peak/2; issue-side figure in `executed`"})
                else:
                    roof.update({"achieved": None, "frac": None})
                roof["hbm"] = {"achieved": hbm, "peak": PEAK_HBM_GBS, "frac": hbm / PEAK_HBM_GBS, "unit": "GB/s",
                               "bytes_per_ray": io_b, "algorithmic_bytes_per_launch": io_b * rays_per_launch, "traffic": traffic}
            else:
                # configs[4] (2 M-triangle height field): SURVEY 8(d) guessed HBM; the counters say otherwise (11.7 GB per
                # launch = 0.03 of the HBM peak, L1 hit 98 %).  A ray's walk is a chain of ~80 dependent per-lane gathers:
                # the bound the counters name is GATHER ISSUE -- busy cycles of the 256 texture-address units over the
                # cycles of the launch (rocprofv3 --pmc TA_TA_BUSY_sum / GRBM_GUI_ACTIVE, profiles/ta_busy_cfg5.json, of
                # the previous collection; live figures: avg_launch_ms).  The HBM formula of BASELINE.md section 3 rides
                # along as a sub-field (bytes if nothing were reused), with the measured traffic beside it.
                g = roof.pop("gather", None)
                roof.update({"bound": "gather_issue", "unit": "TA-busy cycles per launch (256 units)", "traffic": traffic,
                             "achieved": g["ta_busy_cycles"] if g else None, "peak": g["ta_units"] * g["launch_cycles"] if g else None,
                             "frac": g["frac"] if g else None,
                             "gather_wave_instructions": g["gather_wave_instructions"] if g else None,
                             "gather_source": ("profiles/ta_busy_cfg5.json: " + g["source"]) if g else None})
                if per:
                    tb = traversal_bytes_per_ray(per)
                    ach = (io_b + tb) * rays_per_launch / secs / 1e9
                    roof["hbm"] = {"formula_GBps_if_nothing_were_reused": ach, "achieved": (traffic / secs / 1e9) if traffic else None,
                                   "peak": PEAK_HBM_GBS, "frac": (traffic / secs / 1e9 / PEAK_HBM_GBS) if traffic else None, "unit": "GB/s",
                                   "bytes_per_ray": io_b + tb, "io_bytes_per_ray": io_b, "traversal_bytes_per_ray": tb,
                                   "algorithmic_bytes_per_launch": (io_b + tb) * rays_per_launch, "traffic": traffic,
                                   "formula": "B_io + N_node*16 + N_leaf*128 + N_tri*24 + N_tri_box_hit*72 + hit*32 (BASELINE.md section 3)",
                                   "note": "achieved/frac = MEASURED HBM traffic over the launch (the formula's bytes are served by L1/L2/MALL)"}
                    roof["per_primary_ray"] = per
                    fl = flops_per_ray(per)
                    roof["valu_f64"] = {"achieved": fl * rays_per_launch / secs / 1e12, "peak": PEAK_F64_TFLOPS,
                                        "frac": fl * rays_per_launch / secs / 1e12 / PEAK_F64_TFLOPS, "flops_per_ray": fl}
            line["roofline"] = roof
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
