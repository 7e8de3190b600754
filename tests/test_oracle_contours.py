"""Pins the oracle's findContours(RETR_EXTERNAL, TC89_L1) + polygon statistics restatement with
known answers and scipy.ndimage cross-checks."""
import numpy as np
from scipy import ndimage

rng = np.random.RandomState(99)


def disc(H, W, cy, cx, r):
    y, x = np.mgrid[:H, :W]
    return (((y - cy) ** 2 + (x - cx) ** 2) <= r * r).astype(np.uint8) * 255


def test_single_pixel(oracle):
    m = np.zeros((9, 9), np.uint8)
    m[4, 6] = 255
    cs = oracle.find_contours(m)
    assert len(cs) == 1
    pts, nchain = cs[0]
    assert nchain == 0 and pts.tolist() == [[6, 4]]
    b = oracle.blob_from_contour(pts, True)
    assert (b["x"], b["y"], b["w"], b["h"]) == (6, 4, 1, 1)
    assert b["area"] == 0 and b["m00"] == 0
    assert (b["cx"], b["cy"]) == (6.0, 4.0)  # genesis fallback: mean of vertices
    b2 = oracle.blob_from_contour(pts, False)
    assert np.isnan(b2["cx"]) and np.isnan(b2["cy"])  # tracking frames: 0/0


def test_filled_rectangle(oracle):
    for (w, h) in [(7, 4), (12, 12), (3, 9), (30, 3), (3, 3)]:
        m = np.zeros((40, 50), np.uint8)
        x0, y0 = 9, 6
        m[y0:y0 + h, x0:x0 + w] = 255
        cs = oracle.find_contours(m)
        assert len(cs) == 1
        pts, nchain = cs[0]
        assert nchain == 2 * (w - 1) + 2 * (h - 1)
        # corners are the dominant points; traversal starts at the raster-first pixel, counter-clockwise
        assert pts.tolist() == [[x0, y0], [x0, y0 + h - 1], [x0 + w - 1, y0 + h - 1], [x0 + w - 1, y0]]
        b = oracle.blob_from_contour(pts, True)
        assert (b["x"], b["y"], b["w"], b["h"]) == (x0, y0, w, h)
        assert b["area"] == (w - 1) * (h - 1)
        assert abs(b["cx"] - (x0 + (w - 1) / 2)) < 1e-5 and abs(b["cy"] - (y0 + (h - 1) / 2)) < 1e-5
        assert abs(b["radius"] - np.sqrt((w - 1) * (h - 1) / 3.14159)) < 1e-12


def test_thin_shapes_collapse_under_tc89(oracle):
    # Teh-Chin pass 3/4 drops one of two adjacent dominant points: 2-px-wide shapes lose a column in
    # the polygon (so boundingRect shrinks).  Behaviour of the restated algorithm, parity unpinned.
    m = np.zeros((40, 50), np.uint8)
    m[6:15, 9:11] = 255
    pts, nchain = oracle.find_contours(m)[0]
    assert nchain == 18 and pts.tolist() == [[10, 14], [10, 6]]
    m[:] = 0
    m[6:8, 9:11] = 255
    pts, nchain = oracle.find_contours(m)[0]
    assert nchain == 4 and pts.tolist() == [[10, 6]]


def test_two_pixel_and_line_blobs(oracle):
    m = np.zeros((10, 10), np.uint8)
    m[3, 3:5] = 255  # 2 px horizontal
    pts, nchain = oracle.find_contours(m)[0]
    assert nchain == 2 and pts.tolist() == [[4, 3]]  # adjacent couple cleaned to one vertex
    m[:] = 0
    m[2:8, 5] = 255  # vertical line: collinear -> zero area -> m00 == 0
    pts, nchain = oracle.find_contours(m)[0]
    assert nchain == 10 and sorted(pts.tolist()) == [[5, 2], [5, 7]]
    b = oracle.blob_from_contour(pts, True)
    assert b["area"] == 0 and (b["w"], b["h"]) == (1, 6) and (b["cx"], b["cy"]) == (5.0, 4.5)


def test_disc_centroid_and_bbox(oracle):
    for r in [3, 5.5, 9, 14.5]:
        m = disc(64, 64, 30, 33, r)
        cs = oracle.find_contours(m)
        assert len(cs) == 1
        b = oracle.blob_from_contour(cs[0][0], True)
        ys, xs = np.nonzero(m)
        # the TC89 polygon may cut corners, but stays within 1 px of the true extent
        assert abs(b["x"] - xs.min()) <= 1 and abs(b["y"] - ys.min()) <= 1
        assert abs(b["w"] - (xs.max() - xs.min() + 1)) <= 2
        assert abs(b["cx"] - 33) <= 0.5 and abs(b["cy"] - 30) <= 0.5
        assert abs(b["radius"] - r) <= 1.0


def test_external_only_and_order(oracle):
    m = np.zeros((60, 60), np.uint8)
    m[5:30, 5:30] = 255
    m[10:25, 10:25] = 0      # thick ring (5 px)
    m[15:20, 15:20] = 255    # island inside the hole -> NOT external
    m[40:45, 8:12] = 255     # blob A (found later in raster order)
    m[50:53, 30:40] = 255    # blob B (found last)
    cs = oracle.find_contours(m)
    assert len(cs) == 3
    firsts = [tuple(p[0]) for p, _ in cs]
    # reverse discovery order: last found first
    assert firsts == [(30, 50), (8, 40), (5, 5)]


def test_border_touching_blob(oracle):
    m = np.zeros((20, 20), np.uint8)
    m[0:4, 0:5] = 255
    m[16:20, 15:20] = 255
    cs = oracle.find_contours(m)
    boxes = sorted((oracle.blob_from_contour(p, True)[k] for k in "xywh") and
                   tuple(oracle.blob_from_contour(p, True)[k] for k in "xywh") for p, _ in cs)
    assert boxes == [(0, 0, 5, 4), (15, 16, 5, 4)]


def test_random_blobs_vs_scipy_label(oracle):
    # sets of outermost 8-connected components and their extents must agree with scipy.ndimage.label
    for trial in range(30):
        H, W = 48, 64
        m = (rng.rand(H, W) < 0.12).astype(np.uint8)
        m = ndimage.binary_dilation(m, iterations=1).astype(np.uint8) if trial % 2 else m
        lab, n = ndimage.label(m, structure=np.ones((3, 3)))
        cs = oracle.find_contours(m * 255)
        # every contour belongs to a distinct component
        comp_of = [lab[p[0][1], p[0][0]] for p, _ in cs]
        assert len(set(comp_of)) == len(cs)
        objs = ndimage.find_objects(lab)
        for (pts, nchain), c in zip(cs, comp_of):
            sl = objs[c - 1]
            # every vertex is a pixel of that same component
            assert all(lab[y, x] == c for x, y in pts)
            b = oracle.blob_from_contour(pts, True)
            assert b["x"] >= sl[1].start and b["x"] + b["w"] <= sl[1].stop
            assert b["y"] >= sl[0].start and b["y"] + b["h"] <= sl[0].stop
            # (the TC89 polygon bbox may be smaller than the component bbox on tiny ragged blobs)
        # components not reported must be nested inside a hole of another component:
        # after filling holes of the reported ones they disappear
        reported = np.isin(lab, comp_of)
        filled = ndimage.binary_fill_holes(reported, structure=np.ones((3, 3)))  # 4-conn background
        missing = [c for c in range(1, n + 1) if c not in comp_of]
        for c in missing:
            assert filled[lab == c].all()
