#!/usr/bin/env python3
"""Randomised cross-check of the GPU PNG decoder (abub_png_decode_dev) against zlib / numpy: random geometries, image kinds,
filter choices per row, zlib levels / strategies / memory levels / window sizes, IDAT chunkings, palettes; a third of the
files damaged (the status must be non-zero exactly where zlib or the scanline check refuses, the pixels equal where both
accept).  Usage (GPU box): python tools/fuzz_png.py [seconds] [seed]"""
import os
import struct
import sys
import time
import zlib

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from autobub3hs_amd import hip  # noqa: E402
from test_gpu_png import chunk, expect_grey, filter_rows  # noqa: E402


def unfilter(raw, W, H):
    out = np.zeros((H, W), dtype=np.uint8)
    prev = np.zeros(W, dtype=np.int32)
    for y in range(H):
        ft = raw[y * (W + 1)]
        row = np.frombuffer(raw[y * (W + 1) + 1:(y + 1) * (W + 1)], dtype=np.uint8).astype(np.int32)
        cur = np.zeros(W, dtype=np.int32)
        if ft == 0:
            cur = row
        elif ft == 2:
            cur = (row + prev) & 255
        else:
            a = c = 0
            for x in range(W):
                b = int(prev[x])
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                a = (int(row[x]) + pred) & 255
                cur[x] = a
                c = b
        out[y] = cur
        prev = cur
    return out


def one_batch(rs):
    W = int(rs.choice([4, 8, 12, 64, 100, 128, 260, 512]))
    H = int(rs.choice([1, 2, 5, 17, 40]))
    files, expect = [], []
    for _ in range(24):
        kind = rs.randint(0, 5)
        yy, xx = np.mgrid[0:H, 0:W]
        if kind == 0:
            img = rs.randint(0, 256, (H, W))
        elif kind == 1:
            img = np.clip(80 + xx // 3 + rs.randint(-3, 4, (H, W)), 0, 255)
        elif kind == 2:
            img = np.full((H, W), rs.randint(0, 256))
        elif kind == 3:
            img = (xx // max(1, rs.randint(1, 9)) + yy) % 7 * 30
        else:
            img = rs.randint(0, 4, (H, W)) * 60
        img = img.astype(np.uint8)
        filters = rs.randint(0, 5, H) if rs.rand() < 0.7 else np.full(H, rs.randint(0, 5))
        palette = None
        if rs.rand() < 0.25:
            npal = int(rs.choice([256, 100, 7]))
            palette = rs.randint(0, 256, 3 * npal).astype(np.uint8).tobytes()
            img = (img.astype(np.int32) % npal).astype(np.uint8)
        raw = filter_rows(img, filters)
        level = int(rs.choice([0, 1, 2, 4, 6, 9]))
        strat = int(rs.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]))
        co = zlib.compressobj(level, zlib.DEFLATED, int(rs.choice([9, 11, 15])), int(rs.choice([1, 5, 8, 9])), strat)
        z = bytearray(co.compress(raw) + co.flush())
        damaged = rs.rand() < 0.33 and len(z) > 8
        if damaged:
            how = rs.randint(0, 4)
            if how == 0:
                z = z[:rs.randint(0, len(z))]
            elif how == 1:
                z[rs.randint(0, len(z))] ^= 1 << rs.randint(0, 8)
            elif how == 2:
                z[2 + rs.randint(0, min(30, len(z) - 2))] = rs.randint(0, 256)
            else:
                z[-1 - rs.randint(0, 4)] ^= 0x55
        z = bytes(z)
        want = None
        try:
            d = zlib.decompressobj()
            res = d.decompress(z)
            if d.eof and len(res) == H * (W + 1) and all(res[y * (W + 1)] <= 4 for y in range(H)):
                want = expect_grey(unfilter(res, W, H), palette)
        except zlib.error:
            pass
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 3 if palette is not None else 0, 0, 0, 0))
        if palette is not None:
            png += chunk(b"PLTE", palette)
        step = int(rs.choice([1 << 16, 4096, 97, 7]))
        for k in range(0, max(len(z), 1), step):
            png += chunk(b"IDAT", z[k:k + step])
        files.append(png + chunk(b"IEND", b""))
        expect.append(want)
    got, st = hip.png_decode(files, W, H)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for k, want in enumerate(expect):
        if want is None:
            assert st[k] != 0, ("accepted a stream zlib refuses", W, H, k)
        else:
            assert st[k] == 0, ("refused a good stream", W, H, k, st[k])
            assert np.array_equal(got[k], want), ("pixels", W, H, k)
    return len(files), sum(w is None for w in expect)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        a, b = one_batch(rs)
        n += a
        bad += b
    print(f"OK: {n} files ({bad} of them refused by zlib and by the GPU alike)")


if __name__ == "__main__":
    main()
