#!/usr/bin/env python3
"""Generates the PNG fixtures of tests/golden/png/ (run once, outputs committed):

    python tests/golden/make_png_fixtures.py

The reference's own cubemap PNGs are all 8-bit, non-interlaced RGB / RGBA / palette files, so they pin only
that part of `image::open(..).to_rgb8()` (src/behavior/cubemap.rs:19-44).  These files cover the rest of the
format the decoder in csrc/host_io.cpp accepts: Adam7 interlacing, 16-bit samples, packed grey, grey+alpha,
all five scanline filters, sizes that leave Adam7 passes empty.  They are written by the small encoder below
(PIL cannot write interlaced or 16-bit RGB files); `expected.npz` holds the RGB8 image each must decode to
(alpha dropped, grey replicated, packed grey scaled to 0..255, 16-bit samples reduced to their high byte),
and the test also decodes every file with PIL as an independent reader.
"""
import os
import struct
import zlib

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "png")
ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def pack_rows(samples, depth):
    """samples [h][w*channels] ints -> list of byte rows"""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(int(v) for v in r))
        elif depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            per = 8 // depth
            out = bytearray((len(r) + per - 1) // per)
            for i, v in enumerate(r):
                out[i // per] |= int(v) << ((per - 1 - i % per) * depth)
            rows.append(bytes(out))
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, rng):
    out = b""
    prev = bytes(len(rows[0])) if rows else b""
    for row in rows:
        ft = int(rng.integers(0, 5))
        f = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) // 2, paeth(a, b, c))[ft]
            f[i] = (v - pred) & 255
        out += bytes([ft]) + bytes(f)
        prev = row
    return out


def write_png(path, samples, w, h, ctype, depth, interlace, rng, plte=None):
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(1, channels * depth // 8)
    s = np.asarray(samples).reshape(h, w, channels)
    raw = b""
    if interlace:
        for (x0, y0, dx, dy) in ADAM7:
            sub = s[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += filter_rows(pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, rng)
    else:
        raw = filter_rows(pack_rows(s.reshape(h, -1), depth), bpp, rng)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    if plte is not None:
        data += chunk(b"PLTE", bytes(int(v) for v in np.asarray(plte).ravel()))
    z = zlib.compress(raw, 9)
    half = len(z) // 2
    data += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")      # two IDAT chunks on purpose
    open(path, "wb").write(data)


def main():
    os.makedirs(HERE, exist_ok=True)
    rng = np.random.default_rng(2024)
    expected = {}
    cases = [  # name, w, h, colour type, depth, interlaced
        ("rgb8_adam7", 13, 11, 2, 8, True), ("rgba8_adam7", 9, 5, 6, 8, True), ("rgb16", 7, 6, 2, 16, False),
        ("rgba16_adam7", 6, 9, 6, 16, True), ("grey16_adam7", 10, 3, 0, 16, True), ("grey1", 19, 4, 0, 1, False),
        ("grey2_adam7", 11, 7, 0, 2, True), ("grey4", 5, 5, 0, 4, False), ("greya8_adam7", 8, 8, 4, 8, True),
        ("greya16", 3, 4, 4, 16, False), ("pal4_adam7", 17, 9, 3, 4, True), ("pal1", 9, 2, 3, 1, False),
        ("pal8_adam7", 33, 3, 3, 8, True), ("rgb8_1x1_adam7", 1, 1, 2, 8, True), ("rgb8_2x1_adam7", 2, 1, 2, 8, True),
        ("rgb8_1x5_adam7", 1, 5, 2, 8, True), ("rgb16_adam7_wide", 40, 2, 2, 16, True),
    ]
    for name, w, h, ctype, depth, il in cases:
        channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        plte = None
        if ctype == 3:
            plte = rng.integers(0, 256, (1 << depth, 3))
            s = rng.integers(0, 1 << depth, (h, w, 1))
            rgb = plte[s[..., 0]]
        else:
            s = rng.integers(0, 1 << depth, (h, w, channels))
            v8 = s >> 8 if depth == 16 else (s * 255 // ((1 << depth) - 1) if depth < 8 else s)
            rgb = v8[..., :3] if ctype in (2, 6) else np.repeat(v8[..., :1], 3, axis=2)
        write_png(os.path.join(HERE, name + ".png"), s, w, h, ctype, depth, il, rng, plte)
        expected[name] = rgb.astype(np.uint8)
    np.savez(os.path.join(HERE, "expected.npz"), **expected)
    print("wrote %d PNGs + expected.npz to %s" % (len(cases), HERE))


if __name__ == "__main__":
    main()
