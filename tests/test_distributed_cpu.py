"""The N>1 path on CPU: world_size-2 and -3 gloo groups run the same
tile-ownership / all-gather / de-interleave sequence bench.py runs over RCCL.
Each rank's compact tile buffer is cut from an oracle-rendered frame (the GPU
render itself is covered by tests/test_parity_gpu.py::test_tile_sharding_*)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, frame, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rustraytracer_amd import distributed as D, tiles
        h, w = frame.shape[:2]
        local, gathered = D.alloc_tile_buffers(w, h, world, "cpu")
        local.copy_(torch.from_numpy(tiles.extract_rank_tiles(frame, rank, world)))
        D.gather_tiles(local, gathered)
        out = tiles.assemble_numpy(gathered.numpy(), w, h, world)
        tmax = D.max_over_ranks(float(rank + 1), "cpu")
        tot = D.sum_over_ranks([rank + 1, 10], "cpu")
        q.put((rank, bool(np.array_equal(out, frame)), tmax, tot))
    except Exception as e:      # surface the failure instead of leaving the parent waiting on the queue
        q.put((rank, False, repr(e), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (100, 75)), (3, (33, 9)), (2, (64, 16))])
def test_tile_gather_over_gloo(O, scenes, world, size):
    w, h = size
    frame = scenes.cfg1_scene(O.OracleScene(), O).render(w, h, n_threads=2, want=("rgb",))["rgb"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, frame, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for _, same, tmax, tot in res:
        assert same
        assert tmax == float(world)
        assert tot == [world * (world + 1) // 2, 10 * world]


def test_tile_ownership_is_a_partition():
    from rustraytracer_amd import tiles
    for (w, h, world) in ((1920, 1080, 8), (3840, 2160, 8), (33, 9, 2), (500, 270, 3)):
        tx, ty = tiles.tile_grid(w, h)
        slots = tiles.tile_slots(w, h, world)
        seen = set()
        for t in range(tx * ty):
            r, s = tiles.owner_of_tile(t, world)
            assert 0 <= r < world and 0 <= s < slots
            seen.add((r, s))
        assert len(seen) == tx * ty
        # balanced within one tile
        per = [sum(1 for (r, _) in seen if r == k) for k in range(world)]
        assert max(per) - min(per) <= 1
        # the tiles' places in the image are a permutation of the grid, and the rotation of the rows makes any
        # `world` consecutive tiles of a COLUMN belong to `world` different ranks (as those of a row do)
        skew = tiles.tile_skew(tx, world)
        place = {}
        for t in range(tx * ty):
            x0, y0 = tiles.tile_origin(t, tx, world)
            place[(x0 // tiles.TILE_W, y0 // tiles.TILE_H)] = t % world
        assert len(place) == tx * ty
        if ty >= 2 * world and tx > world:
            col = [place[(tx // 2, y)] for y in range(world)]
            assert len(set(col)) == world, (w, h, world, skew, col)


def test_tile_skew_matches_the_library(H):
    from rustraytracer_amd import binding as B, tiles
    for w in (1920, 3840, 500, 33, 640):
        for world in (1, 2, 3, 4, 6, 8):
            assert B.lib.rrt_tile_skew(w, world) == tiles.tile_skew(tiles.tile_grid(w, 8)[0], world), (w, world)
    assert tiles.tile_skew(60, 8) == 1 and tiles.tile_skew(60, 1) == 0


def _worker_dealt(rank, world, port, frame, classes, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rustraytracer_amd import distributed as D, tiles
        h, w = frame.shape[:2]
        tables = tiles.deal_by_class(classes, world)            # every rank computes the same deal from the same classes
        local, gathered = D.alloc_tile_buffers(w, h, world, "cpu", dealt=True)
        local.copy_(torch.from_numpy(tiles.extract_rank_tiles_dealt(frame, tables[rank], w, h, world)))
        D.gather_tiles(local, gathered)
        out = tiles.assemble_dealt_numpy(gathered.numpy(), w, h, world)
        q.put((rank, bool(np.array_equal(out, frame)), None, None))
    except Exception as e:
        q.put((rank, False, repr(e), None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, (100, 75)), (3, (333, 49))])
def test_dealt_tile_gather_over_gloo(O, scenes, world, size):
    """tile_deal = 1 over gloo: the tiles are dealt by a cost class (here: a synthetic one), every rank's buffer carries
    its own slot -> tile table behind the pixels, and the gathered bytes alone rebuild the frame."""
    from rustraytracer_amd import tiles
    w, h = size
    frame = scenes.cfg1_scene(O.OracleScene(), O).render(w, h, n_threads=2, want=("rgb",))["rgb"]
    tx, ty = tiles.tile_grid(w, h)
    classes = np.random.default_rng(3).integers(0, 3, tx * ty)
    tables = tiles.deal_by_class(classes, world)
    owned = np.sort(tables[tables >= 0])
    assert np.array_equal(owned, np.arange(tx * ty))                      # a partition of the image's tiles
    for c in range(3):                                                   # every rank gets its share of every class
        per_rank = [(classes[t[t >= 0]] == c).sum() for t in tables]
        assert max(per_rank) - min(per_rank) <= 2          # (a class may begin and end in the middle of a round)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_dealt, args=(r, world, port, frame, classes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
