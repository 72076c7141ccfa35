#!/usr/bin/env python3
"""Benchmark of the tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path (primary rays -> closest hit ->
shading -> reflection bounces -> sky lookup -> accumulate -> quantise, and for
N > 1 the framebuffer gather over RCCL + de-interleave) over BASELINE.json's
configs[1]: the main.rs world (ufo_fix.obj kd-tree + 2 spheres) under the
blue_sunset cubemap at 1920x1080, 16 spp.  The scene is resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.

Metric: Mrays/s, whole job.  A ray = one world-level closest-hit query (one
`RayTracer::cast`, src/tracer.rs:177) actually executed on the GPU, with
reference semantics (prune_zero_mix = 0: the reflection ray is cast even when
its mix is 0.0, as src/tracer.rs:195-201 does).

Multi-GPU (weak scaling, the default): the image's 32x8 tiles are dealt
round-robin over the ranks and the frame is rendered at 16*N spp, so each GPU
keeps one 1080p/16spp frame's worth of samples per step; one all-gather of the
compact RGB8 tile buffers (1080p: 6.2 MB in total) and a de-interleave on every
rank end the step.  --scaling strong keeps 16 spp in total instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first: brings the process's HIP runtime)
import torch.distributed as dist  # noqa: E402

W, H, SPP = 1920, 1080, 16

# --config: the other BASELINE.json configs (parity-test cases; measured for the table in DESIGN.md,
# not the contract's bench line, which is cfg2).  name -> (scene function, W, H, spp, max_bounce, what)
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

# Algorithmic work per ray for the roofline (DESIGN.md "Measurement"): sizes of the
# device layout in csrc/rrt_device.h and flop counts of SURVEY §8(d).
S_NODE, S_LEAF, S_TRI, S_HITN = 16, 88, 76, 24     # bytes: interior step / leaf header / triangle test / winner's normal
F_TRI, F_BOX, F_NODE, F_SHADE, F_SPHERE = 53, 30, 1, 120, 25
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F64_TFLOPS = 78.6      # vector FP64 (FMA-counted; this path may not contract FMAs)


def cpu_leg(n_threads, row_stride=1):
    """Times the CPU oracle (a port of the reference, oracle/) on a bounded,
    strided-row sample of the same workload and counts the per-ray traversal
    work that prices the roofline."""
    import numpy as np
    from oracle import oracle as O
    from rustraytracer_amd import scenes
    cpu = scenes.cfg2_scene(O.OracleScene(), O)
    cpu.render(W, H, spp=1, n_threads=n_threads, y0=0, y1=64, want=())            # warm
    dt, frames, c = 0.0, 0, None
    while dt < 2.0 and frames < 16:      # >= 2 s of wall time on n_threads cores (~30 core-seconds)
        t0 = time.perf_counter()
        out = cpu.render(W, H, spp=SPP, seed=0, n_threads=n_threads, y0=3, row_stride=row_stride, want=())
        dt += time.perf_counter() - t0
        frames += 1
        c = dict(zip(O.COUNTER_NAMES, (int(x) for x in out["counters"])))
    # the reference's own work split (SURVEY 8d "faithful" mode): 12 threads over 12 contiguous row-major
    # chunks of W*H/12 pixels (tracer.rs:117-158) -- the slowest chunk sets the time
    O.lib().orc_set_static_chunks(1)
    t0 = time.perf_counter()
    outf = cpu.render(W, H, spp=SPP, seed=0, n_threads=12, want=())
    dtf = time.perf_counter() - t0
    O.lib().orc_set_static_chunks(0)
    faithful = {"value": int(outf["counters"][0]) / dtf / 1e6, "unit": "Mrays/s", "threads": 12,
                "sample": "one full %dx%d frame at %d spp in 12 static contiguous pixel chunks as tracer.rs:117-158: "
                          "%d rays in %.2f s" % (W, H, SPP, int(outf["counters"][0]), dtf)}
    rows = len(range(3, H, row_stride))
    # per-primary-ray traversal work: closest-hit of the sample-0 primary rays of those rows
    import ctypes as C
    st = O.Settings(W, H, 1, 2, 0, 0, 1)
    cam = O.default_camera()
    ys = list(range(3, H, row_stride))
    rays = np.zeros((len(ys) * W, 6))
    r = np.zeros(6)
    L = O.lib()
    k = 0
    for y in ys[::32]:
        for x in range(W):
            L.orc_primary_ray(C.byref(cam), C.byref(st), x, y, 0, r.ctypes.data)
            rays[k] = r
            k += 1
    pc = dict(zip(O.COUNTER_NAMES, (int(x) for x in cpu.trace_rays(rays[:k])["counters"])))
    return {
        "value": c["rays"] * frames / dt / 1e6, "unit": "Mrays/s", "cores": n_threads, "kind": "port",
        "sample": "rows 3..%d step %d of the %dx%d frame (%d rows), %d spp, rendered %d times: %d rays in %.2f s; "
                  "C++ f64 restatement of the reference, rows handed to threads dynamically (no per-hit Material "
                  "clone: optimistic for the Rust binary)" % (H - 1, row_stride, W, H, rows, SPP, frames,
                                                              c["rays"] * frames, dt),
        "reference_work_split": faithful,
    }, c, pc, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--prune", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--batch-spp", type=int, default=0, help="samples per pixel per wavefront batch (0 = from the work budget)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg2",
                    help="BASELINE.json config to run (default cfg2 = the contract's workload; the others "
                         "print the same line without roofline/cpu_baseline)")
    ap.add_argument("--budget-gib", type=float, default=0.0, help="work-buffer budget in GiB (0 = library default)")
    ap.add_argument("--cull-mode", type=int, default=-1, help="rrt_set_cull_mode (0 off, 1 on, 2 verify; -1 = library default)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
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

    from rustraytracer_amd import host as Hst, scenes
    global W, H, SPP
    scene_fn, W, H, SPP, max_bounce, what = CONFIGS[args.config]
    scene = getattr(scenes, scene_fn)(Hst.World(device=local_rank), Hst.loaders)
    if args.budget_gib > 0:
        scene.set_work_budget(int(args.budget_gib * (1 << 30)))
    if args.cull_mode >= 0:
        scene.set_cull_mode(args.cull_mode)
    scene.commit()                                   # scene resident in HBM before timing
    spp = SPP * world if args.scaling == "weak" else SPP
    st = Hst.make_settings(W, H, spp=spp, max_bounce=max_bounce, seed=0, prune_zero_mix=args.prune,
                           tile_rank=rank, tile_world=world, batch_spp=args.batch_spp)
    from rustraytracer_amd import distributed as D
    frame = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    if world > 1:
        local, gathered = D.alloc_tile_buffers(W, H, world, dev)

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        if world == 1:
            scene.render_device(st, frame.data_ptr(), stream=stream)
        else:
            scene.render_device(st, local.data_ptr(), stream=stream)
            D.gather_tiles(local, gathered)
            Hst.assemble_tiles(gathered.data_ptr(), W, H, world, frame.data_ptr(), stream=stream)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    prim_ms = bnc_ms = 0.0
    prim_n = bnc_n = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = scene.stats()                             # per-kernel hipEvent timings of the last timed step
    prim_ms, prim_n = stats["primary_kernel_ms"], stats["primary_kernel_launches"]
    bnc_ms, bnc_n = stats["bounce_kernel_ms"], stats["bounce_kernel_launches"]

    host_ms = None
    if world == 1:
        # PCIe-inclusive variant (never `value`): rrt_render into host memory = the same frame + one D2H copy
        import numpy as np
        import ctypes as C
        from rustraytracer_amd import binding as B
        rgb = np.zeros((H, W, 3), np.uint8)
        cam = Hst.default_camera()
        n_host = max(2, min(args.steps, 5))
        for it in range(n_host + 1):
            if it == 1:
                th = time.perf_counter()
            B.check(B.lib.rrt_render(scene.h, C.byref(cam), C.byref(st), rgb.ctypes.data_as(C.c_void_p), None, None, None))
        host_ms = (time.perf_counter() - th) / n_host * 1e3

    dt = D.max_over_ranks(dt, dev)
    rays_step, paths_step, overflow = D.sum_over_ranks(
        [stats["rays_traced"], stats["paths"], stats["traversal_overflow"]], dev)

    if rank == 0:
        value = rays_step * args.steps / dt / 1e6
        line = {
            "metric": "Mrays/s (node) ufo.obj kd-tree 1080p", "value": value, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic camera rays over the reference's own assets (obj/ufo_fix.obj, cubemaps/blue_sunset)",
            "config": {"workload": "BASELINE configs[1]: main.rs world (ufo_fix.obj rope kd-tree, 1705 tris, + 2 spheres), "
                                   "blue_sunset cubemap, %dx%d, %d spp%s, max_bounce 2, reference ray semantics (prune_zero_mix=%d)"
                                   % (W, H, spp, " (16 per GPU)" if world > 1 and args.scaling == "weak" else "", args.prune)
                       if args.config == "cfg2" else "BASELINE %s; %d spp in total, max_bounce %d" % (what, spp, max_bounce),
                       "rays_per_step": rays_step, "samples_per_step": paths_step,
                       "msamples_per_s": paths_step * args.steps / dt / 1e6,
                       "parallelism": "tiles%d" % world, "traversal_overflow": overflow},
        }
        if host_ms is not None:
            line["config"]["host_buffer_ms_per_step"] = host_ms      # rrt_render: + D2H of the RGB8 frame, blocking
        cpu_c = prim_c = None
        if args.config != "cfg2":
            line["metric"] = "Mrays/s (node) %s" % args.config
            line["config"]["n_batches"] = stats["n_batches"]
            line["config"]["level_rays"] = stats["level_rays"]
            line["config"]["kernel_ms"] = {"k_wave_primary": prim_ms, "k_wave_bounce": bnc_ms, "gpu": stats["gpu_ms"]}
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
            nthreads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            line["cpu_baseline"], cpu_c, prim_c, n_prim = cpu_leg(nthreads)
        if world == 1 and args.config == "cfg2":
            # Dominant kernel = k_wave_primary (one launch per wavefront batch).  Its roofline is priced as
            # DESIGN.md "Measurement" states: the scene of this config (~0.4 MB of nodes/triangles) is
            # cache-resident, so the bytes that MUST cross HBM per primary ray are its outputs only:
            #   path that ends in the sky: 3 B cubemap texel read + 4 B compact sample (the texel itself, samp32)
            #   continuing path: 56 B shading record + 56 B queue entry (o, d, path, medium) + 4 B samp32 tag,
            #                    + 24 B sample colour left in advance when its reflection ray has mix 0.0
            # `achieved` = those bytes x rays per launch / measured launch time (hipEvents on the launch stream).
            # The kernel is f64-VALU/latency bound; that ceiling is reported beside it as valu_f64.
            launch_ms = prim_ms / max(prim_n, 1)
            rays_per_launch = stats["primary_rays"] / max(prim_n, 1)
            lv = stats["level_rays"]
            cont = lv[1] / max(lv[0], 1)
            dead = stats["zero_mix_paths"] / max(lv[0], 1)
            hbm_bytes_ray = (1.0 - cont) * 7.0 + cont * 116.0 + dead * 24.0
            ach = hbm_bytes_ray * rays_per_launch / (launch_ms * 1e-3) / 1e9
            roof = {"kernel": "k_wave_primary", "bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS,
                    "achieved": ach, "frac": ach / PEAK_HBM_GBS, "traffic": None,
                    "avg_launch_ms": launch_ms, "launches_per_step": prim_n, "rays_per_launch": rays_per_launch,
                    "hbm_bytes_per_ray": hbm_bytes_ray, "continuing_fraction": cont, "zero_mix_fraction": dead,
                    "other_kernels": {"k_wave_bounce_ms_per_step": bnc_ms, "k_wave_bounce_launches": bnc_n,
                                      "k_wave_primary_ms_per_step": prim_ms, "gpu_ms_per_step": stats["gpu_ms"],
                                      "level_rays": lv}}
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tf):       # PMC FETCH_SIZE/WRITE_SIZE per launch, collected by tools/collect_pmc.sh
                try:
                    t = json.load(open(tf)).get("k_wave_primary")
                    if t:
                        roof["traffic"] = t["hbm_bytes_per_launch"]
                        roof["traffic_source"] = t.get("source")
                except (ValueError, KeyError):
                    pass
            vf = os.path.join(ROOT, "profiles", "valu_busy.json")
            if os.path.exists(vf):       # SQ_ACTIVE_INST_VALU over the launch's SIMD-cycles (tools/valu_busy.py)
                try:
                    roof["valu_issue"] = json.load(open(vf)).get("k_wave_primary")
                except ValueError:
                    pass
            if prim_c:
                n = float(n_prim)
                per = {k: prim_c[k] / n for k in ("node_steps", "leaf_visits", "tri_tests", "sphere_tests", "box_tests")}
                onchip = per["node_steps"] * S_NODE + per["leaf_visits"] * S_LEAF + per["tri_tests"] * S_TRI
                flops_ray = (per["tri_tests"] * F_TRI + per["box_tests"] * F_BOX + per["node_steps"] * F_NODE
                             + per["sphere_tests"] * F_SPHERE + F_SHADE)
                tf64 = flops_ray * rays_per_launch / (launch_ms * 1e-3) / 1e12
                roof["valu_f64"] = {"achieved": tf64, "peak": PEAK_F64_TFLOPS, "frac": tf64 / PEAK_F64_TFLOPS,
                                    "unit": "TFLOP/s", "flops_per_ray": flops_ray, "per_primary_ray": per,
                                    "cache_served_bytes_per_ray": onchip,
                                    "cache_served_GBs": onchip * rays_per_launch / (launch_ms * 1e-3) / 1e9}
            line["roofline"] = roof
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
