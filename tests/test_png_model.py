"""tools/inflate_wave_model.py (the lane-level model of the GPU inflate: same phases, same tables) against zlib, on the CPU."""
import os
import sys
import zlib

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import inflate_wave_model as model  # noqa: E402


def streams():
    rs = np.random.RandomState(3)
    yy, xx = np.mgrid[0:40, 0:200]
    smooth = np.clip(60 + xx // 4 + rs.randint(-3, 4, xx.shape), 0, 255).astype(np.uint8).tobytes()
    data = {"smooth": smooth, "random": rs.randint(0, 256, 6000).astype(np.uint8).tobytes(), "constant": bytes(5000),
            "text": (b"the quick brown fox jumps over the lazy dog " * 90), "empty": b"", "one": b"x"}
    for name, raw in data.items():
        for level, strat in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                             (6, zlib.Z_HUFFMAN_ONLY), (9, zlib.Z_RLE)):
            co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strat)
            yield name, level, strat, raw, co.compress(raw) + co.flush()


def test_model_equals_zlib():
    for name, level, strat, raw, z in streams():
        assert model.inflate_wave(z, len(raw)) == raw, (name, level, strat)


def test_model_refuses_what_zlib_refuses():
    rs = np.random.RandomState(4)
    raw = bytes(rs.randint(0, 4, 3000).astype(np.uint8))
    z = bytearray(zlib.compress(raw, 6))
    for trial in range(40):
        bad = bytearray(z)
        bad[rs.randint(2, len(bad))] ^= 1 << rs.randint(0, 8)
        try:
            want = zlib.decompress(bytes(bad))
        except zlib.error:
            want = None
        try:
            got = model.inflate_wave(bytes(bad), len(raw))
        except model.Corrupt:
            got = None
        if want is None or len(want) != len(raw):
            assert got is None, trial
        else:
            assert got == want, trial
