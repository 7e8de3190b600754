"""CPU tests of the product's host logic (libabub_host.so: sparse contour finder, TC89, polygon
statistics, Otsu-from-histogram, significance) against the oracle's dense, full-image restatement."""
import math

import numpy as np
import pytest
from scipy import ndimage

from autobub3hs_amd import host

rng = np.random.RandomState(2024)


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


def masks():
    H, W = 60, 80
    yield np.zeros((H, W), np.uint8)
    m = np.zeros((H, W), np.uint8)
    m[10:30, 10:40] = 255
    m[15:25, 15:35] = 0
    m[18:22, 20:30] = 255   # island in a hole
    m[0:3, 70:80] = 255     # touches the border
    m[59, 0] = 255          # single pixel in the corner
    m[40:42, 5:7] = 255
    yield m
    y, x = np.mgrid[:H, :W]
    yield ((((y - 30) ** 2 + (x - 40) ** 2) <= 15 ** 2) & (((y - 30) ** 2 + (x - 40) ** 2) >= 14 ** 2)).astype(np.uint8) * 255  # thin ring
    d = (((y - 30) ** 2 + (x - 40) ** 2) <= 15 ** 2) & (((y - 30) ** 2 + (x - 40) ** 2) >= 13.2 ** 2)
    d[28:33, 38:43] = True  # blob inside a thin ring
    yield d.astype(np.uint8) * 255
    for t in range(40):
        m = (rng.rand(H, W) < (0.05 + 0.01 * t)).astype(np.uint8)
        if t % 3 == 0:
            m = ndimage.binary_dilation(m, iterations=1 + t % 2).astype(np.uint8)
        if t % 4 == 1:
            m = ndimage.binary_closing(m, iterations=2).astype(np.uint8)
        yield m * 255


def test_sparse_contours_equal_dense_oracle(oracle):
    n_total = 0
    for m in masks():
        H, W = m.shape
        ref = oracle.find_contours(m)
        idx = np.flatnonzero(m.ravel()).astype(np.uint32)
        rng.shuffle(idx)  # the GPU compaction returns an arbitrary order
        got = host.contours_from_indices(idx, W, H)
        assert len(got) == len(ref)
        for g, (r, _) in zip(got, ref):
            assert np.array_equal(g, r)
            gs, rs = host.blob_stats(g), oracle.blob_from_contour(r, True)
            for k in ("x", "y", "w", "h", "area", "m00", "m10", "m01"):
                assert gs[k] == rs[k]
        n_total += len(got)
    assert n_total > 500


def test_binarize_threshold_equals_oracle(oracle):
    for t in range(60):
        img = np.zeros((50, 70), np.uint8)
        nb = rng.randint(0, 5)
        for _ in range(nb):
            y, x = rng.randint(0, 40), rng.randint(0, 60)
            img[y:y + rng.randint(2, 9), x:x + rng.randint(2, 9)] = rng.randint(1, 120)
        img = np.clip(img + (rng.rand(50, 70) < 0.02) * rng.randint(1, 6, (50, 70)), 0, 255).astype(np.uint8)
        tz = [2, 3][t % 2]
        mask, T = oracle.binarize(img, tz)
        thr = host.binarize_threshold(np.bincount(img.ravel(), minlength=256), img.size, tz)
        assert thr == max(tz, T)
        assert np.array_equal(mask > 0, img > thr)


def test_entropy_equals_oracle(oracle):
    for _ in range(10):
        img = (rng.rand(40, 40) < 0.1) * rng.randint(0, 256, (40, 40))
        img = img.astype(np.uint8)
        h = np.bincount(img.ravel(), minlength=256)
        assert host.entropy(h, 16, img.size) == oracle.entropy16(img)
        assert host.entropy(h, 128, img.size) == oracle.entropy128(img)


def test_significance_equals_oracle(oracle):
    P = 64 * 64
    for tss in (4, 20):
        a = oracle.Analyzer(np.zeros((6, 64, 64), np.uint8), np.zeros((64, 64), np.uint8),
                            np.zeros((64, 64), np.uint8), tss)
        s = host.Significance(tss)
        for step in range(40):
            h = np.zeros(256, np.int64)
            for v, c in zip(rng.randint(1, 30, 4), rng.randint(0, 60, 4)):
                h[v] += c * (15 if step % 7 == 6 else 1)
            h[0] = P - h.sum()
            store = (step % 3) != 2
            r1 = a.significance(h.astype(np.uint32), store)
            r2 = s(h, P, store)
            assert (math.isnan(r1) and math.isnan(r2)) or r1 == r2
            assert a.state()["loc_thres"] == s.loc_thres.value
        a.close()
