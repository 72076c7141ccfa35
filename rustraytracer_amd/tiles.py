"""Image-tile sharding used by the multi-GPU path (scene replicated, tiles
round-robin over ranks, one framebuffer gather).  Pure index arithmetic that
mirrors k_assemble / local_pixel in csrc/rrt_kernels.hip; the GPU test checks
the two agree."""
import numpy as np

TILE_W, TILE_H = 32, 8


def tile_grid(width, height):
    return (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H


def tile_slots(width, height, world):
    tx, ty = tile_grid(width, height)
    return (tx * ty + world - 1) // world


def tile_skew(tiles_x, world):
    """Mirror of rrt::tile_skew (csrc/rrt_device.h): row ty of the tile grid is rotated by skew * ty columns, so
    that the rank stride down a column of tiles is coprime to the world size (near 0.38 of it)."""
    if world <= 1 or tiles_x == 0:
        return 0
    from math import gcd
    best = min((g for g in range(1, world) if gcd(g, world) == 1), key=lambda g: (abs(g - 0.382 * world), g))
    return (tiles_x % world + world - best) % world


def owner_of_tile(tile, world):
    """(rank, slot) of tile index `tile`."""
    return tile % world, tile // world


def tile_origin(tile, tiles_x, world):
    """Pixel origin (x0, y0) of tile index `tile`: row tile // tiles_x, column rotated by the row's skew."""
    ty = tile // tiles_x
    return ((tile % tiles_x + tile_skew(tiles_x, world) * ty) % tiles_x) * TILE_W, ty * TILE_H


def extract_rank_tiles(frame, rank, world):
    """Compact [slots][8][32][3] buffer rank `rank` produces for a full
    row-major frame (used by the CPU tests to stand in for the GPU render)."""
    h, w = frame.shape[:2]
    tx, ty = tile_grid(w, h)
    slots = tile_slots(w, h, world)
    out = np.zeros((slots, TILE_H, TILE_W, 3), frame.dtype)
    for slot in range(slots):
        tile = slot * world + rank
        if tile >= tx * ty:
            continue
        x0, y0 = tile_origin(tile, tx, world)
        blk = frame[y0:y0 + TILE_H, x0:x0 + TILE_W]
        out[slot, :blk.shape[0], :blk.shape[1]] = blk
    return out


def assemble_numpy(gathered, width, height, world):
    """[world][slots][8][32][3] -> row-major [H][W][3] (what rrt_assemble_tiles does on the GPU)."""
    tx, ty = tile_grid(width, height)
    out = np.zeros((height, width, 3), gathered.dtype)
    for tile in range(tx * ty):
        rank, slot = owner_of_tile(tile, world)
        x0, y0 = tile_origin(tile, tx, world)
        hh, ww = min(TILE_H, height - y0), min(TILE_W, width - x0)
        out[y0:y0 + hh, x0:x0 + ww] = gathered[rank, slot, :hh, :ww]
    return out


# --- dealt mode (rrt_settings.tile_deal): a rank's buffer = its slots' pixels, then its int32 slot -> tile table -------
def dealt_buffer_bytes(width, height, world):
    slots = tile_slots(width, height, world)
    return slots * (TILE_H * TILE_W * 3 + 4)


def split_dealt(buf, width, height, world):
    """One rank's flat dealt buffer -> (pixels [slots][8][32][3], table [slots] int32: image tile ty * tiles_x + tx, -1 = padding)."""
    slots = tile_slots(width, height, world)
    n = slots * TILE_H * TILE_W * 3
    buf = np.ascontiguousarray(buf).reshape(-1)
    return buf[:n].reshape(slots, TILE_H, TILE_W, 3), buf[n:n + 4 * slots].view(np.int32)


def deal_by_class(classes, world):
    """The dealing k_lists does: tiles sorted by class (stable, image order inside a class); position p of that list
    belongs to round p // world, and the rounds are dealt to the ranks to and fro (round 0: ranks 0 .. world-1, round 1:
    world-1 .. 0, ...); slot = round.  classes: one small int per image tile.  -> per-rank tables [world][slots]."""
    classes = np.asarray(classes)
    perm = np.argsort(classes, kind="stable")
    slots = (len(classes) + world - 1) // world
    tables = np.full((world, slots), -1, np.int32)
    for p, tile in enumerate(perm):
        rnd, idx = divmod(p, world)
        tables[world - 1 - idx if rnd & 1 else idx, rnd] = tile
    return tables


def extract_rank_tiles_dealt(frame, table, width, height, world):
    """The flat dealt buffer a rank with slot -> tile table `table` produces for a full row-major frame."""
    tx, _ = tile_grid(width, height)
    slots = tile_slots(width, height, world)
    px = np.zeros((slots, TILE_H, TILE_W, 3), np.uint8)
    for slot, tile in enumerate(table):
        if tile < 0:
            continue
        x0, y0 = (tile % tx) * TILE_W, (tile // tx) * TILE_H
        blk = frame[y0:y0 + TILE_H, x0:x0 + TILE_W]
        px[slot, :blk.shape[0], :blk.shape[1]] = blk
    return np.concatenate([px.reshape(-1), np.asarray(table, np.int32).view(np.uint8)])


def assemble_dealt_numpy(gathered, width, height, world):
    """[world][dealt_buffer_bytes] -> row-major [H][W][3] (what rrt_assemble_tiles_dealt does on the GPU)."""
    tx, _ = tile_grid(width, height)
    out = np.zeros((height, width, 3), np.uint8)
    for rank in range(world):
        px, table = split_dealt(gathered[rank], width, height, world)
        for slot, tile in enumerate(table):
            if tile < 0:
                continue
            x0, y0 = (tile % tx) * TILE_W, (tile // tx) * TILE_H
            hh, ww = min(TILE_H, height - y0), min(TILE_W, width - x0)
            out[y0:y0 + hh, x0:x0 + ww] = px[slot, :hh, :ww]
    return out
