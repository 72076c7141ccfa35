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


def owner_of_tile(tile, world):
    """(rank, slot) of tile index `tile` (row-major over the tile grid)."""
    return tile % world, tile // world


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
        y0, x0 = (tile // tx) * TILE_H, (tile % tx) * TILE_W
        blk = frame[y0:y0 + TILE_H, x0:x0 + TILE_W]
        out[slot, :blk.shape[0], :blk.shape[1]] = blk
    return out


def assemble_numpy(gathered, width, height, world):
    """[world][slots][8][32][3] -> row-major [H][W][3] (what rrt_assemble_tiles does on the GPU)."""
    tx, ty = tile_grid(width, height)
    out = np.zeros((height, width, 3), gathered.dtype)
    for tile in range(tx * ty):
        rank, slot = owner_of_tile(tile, world)
        y0, x0 = (tile // tx) * TILE_H, (tile % tx) * TILE_W
        hh, ww = min(TILE_H, height - y0), min(TILE_W, width - x0)
        out[y0:y0 + hh, x0:x0 + ww] = gathered[rank, slot, :hh, :ww]
    return out
