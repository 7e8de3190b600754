"""Pins the CPU oracle's per-pixel primitives with (a) analytic known-answer tests derivable from the
reference code alone and (b) independent numpy/scipy implementations (SURVEY.md 8c (1),(2)).
The reference ships no tests/goldens for this path, so these are what pins the oracle."""
import numpy as np
import pytest
from scipy import ndimage

rng = np.random.RandomState(1234)


def np_process_frame(cur, ref, sigma):
    c, r, s6 = cur.astype(np.int32), ref.astype(np.int32), 6 * sigma.astype(np.int32)
    pos = np.clip(np.clip(c - r, 0, 255) - s6, 0, 255)
    neg = np.clip(np.clip(r - c, 0, 255) - s6, 0, 255)
    k = np.array([1, 4, 6, 4, 1])

    def g(a):
        t = ndimage.correlate1d(a, k, axis=1, mode="mirror")
        t = ndimage.correlate1d(t, k, axis=0, mode="mirror")
        return (t + 128) >> 8

    return np.abs(g(pos) - g(neg)).astype(np.uint8)


@pytest.mark.parametrize("shape", [(64, 96), (37, 53), (5, 5), (8, 3), (128, 200)])
def test_process_frame_vs_scipy(oracle, shape):
    H, W = shape
    cur = rng.randint(0, 256, (H, W)).astype(np.uint8)
    ref = rng.randint(0, 256, (H, W)).astype(np.uint8)
    sigma = rng.randint(0, 4, (H, W)).astype(np.uint8)
    D = oracle.process_frame(cur, ref, sigma)
    assert np.array_equal(D, np_process_frame(cur, ref, sigma))


def test_process_frame_saturation_and_sigma(oracle):
    H, W = 40, 48
    cur = np.full((H, W), 255, np.uint8)
    ref = np.zeros((H, W), np.uint8)
    sigma = np.full((H, W), 50, np.uint8)  # 6*50 = 300 > 255 -> everything suppressed
    assert not oracle.process_frame(cur, ref, sigma).any()
    sigma[:] = 10  # pos = 255-60 = 195 everywhere; blur of a constant is the constant
    D = oracle.process_frame(cur, ref, sigma)
    assert (D == 195).all()
    D2 = oracle.process_frame(ref, cur, sigma)  # neg branch symmetric
    assert (D2 == 195).all()


def test_process_frame_identical_is_zero(oracle):
    f = rng.randint(0, 256, (33, 47)).astype(np.uint8)
    D = oracle.process_frame(f, f, np.zeros_like(f))
    assert not D.any()
    h = oracle.hist256(D)
    assert h[0] == f.size and h[1:].sum() == 0


def test_process_frame_step_known_answer(oracle):
    # +delta on a k x k square, sigma = 0 -> binomial-blurred plateau with exact (S+128)>>8 values
    H, W, d = 31, 29, 100
    ref = np.full((H, W), 20, np.uint8)
    cur = ref.copy()
    cur[10:17, 8:15] += d
    D = oracle.process_frame(cur, ref, np.zeros_like(ref))
    w = np.array([1, 4, 6, 4, 1])
    for (y, x) in [(13, 11), (10, 8), (9, 8), (8, 8), (8, 6), (12, 16)]:
        S = 0
        for i in range(-2, 3):
            for j in range(-2, 3):
                yy, xx = y + i, x + j
                inside = 10 <= yy < 17 and 8 <= xx < 15
                S += w[i + 2] * w[j + 2] * (d if inside else 0)
        assert D[y, x] == (S + 128) >> 8
    assert D[13, 11] == d  # deep inside the plateau
    assert D[0, 0] == 0


def test_process_frame_roi(oracle):
    H, W = 50, 60
    cur = rng.randint(0, 256, (H, W)).astype(np.uint8)
    ref = rng.randint(0, 256, (H, W)).astype(np.uint8)
    sigma = rng.randint(0, 3, (H, W)).astype(np.uint8)
    rx, ry, rw, rh = 7, 11, 23, 17
    D = oracle.process_frame(cur, ref, sigma, roi=(rx, ry, rw, rh))
    sub = np_process_frame(cur[ry:ry + rh, rx:rx + rw], ref[ry:ry + rh, rx:rx + rw],
                           sigma[ry:ry + rh, rx:rx + rw])  # borders reflect at the ROI edge
    exp = np.zeros((H, W), np.uint8)
    exp[ry:ry + rh, rx:rx + rw] = sub
    assert np.array_equal(D, exp)


def test_hist256_vs_bincount(oracle):
    img = rng.randint(0, 256, (77, 91)).astype(np.uint8)
    assert np.array_equal(oracle.hist256(img), np.bincount(img.ravel(), minlength=256))


@pytest.mark.parametrize("shape", [(40, 56), (3, 3), (17, 5)])
def test_posttrig_vs_scipy(oracle, shape):
    H, W = shape
    f = rng.randint(0, 256, (H, W)).astype(np.uint8)
    mu = rng.randint(0, 256, (H, W)).astype(np.uint8)
    sg = rng.randint(0, 5, (H, W)).astype(np.uint8)
    o = np.clip(np.abs(f.astype(np.int32) - mu) - 6 * sg.astype(np.int32), 0, 255)
    S = ndimage.correlate(o, np.ones((3, 3), np.int32), mode="mirror")
    exp = ((S + 4) // 9).astype(np.uint8)
    assert np.array_equal(oracle.posttrig_frame(f, mu, sg), exp)
    # (S+4)//9 is round-half-nowhere == rint(S/9) for every reachable S
    allS = np.arange(0, 9 * 255 + 1)
    assert np.array_equal((allS + 4) // 9, np.rint(allS / 9.0).astype(int))


def test_welford_known_answers(oracle):
    H, W = 6, 7
    const = np.full((8, H, W), 93, np.uint8)
    mu, sg = oracle.welford(const)
    assert (mu == 93).all() and (sg == 0).all()
    # alternating a,b N times: mean (a+b)/2, var = N/(N-1) * ((b-a)/2)^2
    a, b, N = 10, 31, 10
    st = np.empty((N, H, W), np.uint8)
    st[0::2], st[1::2] = a, b
    mu, sg = oracle.welford(st)
    assert (mu == int((a + b) / 2)).all()
    assert (sg == int(np.sqrt(N / (N - 1.0) * ((b - a) / 2.0) ** 2))).all()


def test_welford_vs_numpy_float32(oracle):
    N, H, W = 13, 9, 11
    st = rng.randint(0, 256, (N, H, W)).astype(np.uint8)
    mean = np.zeros((H, W), np.float32)
    m2 = np.zeros((H, W), np.float32)
    for k in range(N):
        x = st[k].astype(np.float32)
        d = x - mean
        mean = mean + d / np.float32(k + 1)
        m2 = m2 + d * (x - mean)
    sd = np.sqrt(m2 / np.float32(N - 1))
    mu, sg = oracle.welford(st)
    assert np.array_equal(mu, mean.astype(np.int32).astype(np.uint8))
    assert np.array_equal(sg, sd.astype(np.int32).astype(np.uint8))
    # and it is close to the float64 statistics (truncation semantics)
    assert np.abs(mu.astype(float) - np.floor(st.mean(0))).max() <= 1
    assert np.abs(sg.astype(float) - np.floor(st.std(0, ddof=1))).max() <= 1


def test_welford_single_frame_defined_as_zero_sigma(oracle):
    st = rng.randint(0, 256, (1, 4, 5)).astype(np.uint8)
    mu, sg = oracle.welford(st)
    assert np.array_equal(mu, st[0]) and (sg == 0).all()


def test_entropy16(oracle):
    H, W = 32, 32
    img = np.zeros((H, W), np.uint8)
    assert oracle.entropy16(img) == 0.0
    img[:, :16] = 200  # two equally likely bins -> 1 bit
    assert abs(oracle.entropy16(img) - 1.0) < 1e-6
    img = (np.arange(H * W).reshape(H, W) % 256).astype(np.uint8)  # uniform over 16 bins -> 4 bits
    assert abs(oracle.entropy16(img) - 4.0) < 1e-5
    # pair entropy uses the saturating f1 - f0
    f0 = np.full((H, W), 100, np.uint8)
    f1 = np.full((H, W), 90, np.uint8)
    assert oracle.pair_entropy16(f1, f0) == 0.0  # all saturate to 0
    f1[0, 0] = 180  # one pixel in bin 5
    p = 1.0 / (H * W)
    exp = -(p * np.log2(p) + (1 - p) * np.log2(1 - p))
    assert abs(oracle.pair_entropy16(f1, f0) - exp) < 1e-6


def np_otsu(hist):
    # independent restatement of the maximised between-class variance (float64)
    h = hist.astype(np.float64)
    N = h.sum()
    p = h / N
    best, T = 0.0, 0
    i = np.arange(256)
    mu = (i * p).sum()
    for t in range(256):
        q1 = p[: t + 1].sum()
        q2 = 1 - q1
        if min(q1, q2) < np.finfo(np.float32).eps or max(q1, q2) > 1 - np.finfo(np.float32).eps:
            continue
        m1 = (i[: t + 1] * p[: t + 1]).sum() / q1
        m2 = (mu - q1 * m1) / q2
        s = q1 * q2 * (m1 - m2) ** 2
        if s > best * (1 + 1e-12):
            best, T = s, t
    return T


def test_otsu(oracle):
    assert oracle.otsu(np.bincount([0] * 100, minlength=256)) == 0  # all-zero image -> T=0
    h = np.zeros(256, np.uint32)
    h[0], h[40] = 1000, 30
    assert oracle.otsu(h) == 0  # first max wins: any T in [0,39] separates; i=0 is first
    h = np.zeros(256, np.uint32)
    h[0], h[10], h[200] = 5000, 300, 200
    assert oracle.otsu(h) == np_otsu(h)
    for _ in range(20):
        img = np.clip(rng.normal(60, 20, 4000), 0, 255).astype(np.uint8)
        img[:1500] = np.clip(rng.normal(170, 15, 1500), 0, 255).astype(np.uint8)
        h = np.bincount(img, minlength=256).astype(np.uint32)
        assert abs(oracle.otsu(h) - np_otsu(h)) <= 1


def test_binarize(oracle):
    img = np.zeros((20, 20), np.uint8)
    img[5:9, 5:9] = 2       # below loc_thres 3 -> removed by TOZERO
    img[10:15, 10:15] = 50
    m, T = oracle.binarize(img, 3)
    assert T == 0
    assert m[12, 12] == 255 and m[6, 6] == 0 and m.sum() == 25 * 255
    m2, _ = oracle.binarize(np.zeros((8, 8), np.uint8), 3)
    assert not m2.any()


def test_div9_multiply_shift_used_by_k3():
    # the fast post-trigger kernel computes (S+4)/9 as ((S+4)*7282)>>16 on 24-bit multipliers
    S = np.arange(0, 9 * 255 + 1, dtype=np.int64)
    assert np.array_equal(((S + 4) * 7282) >> 16, (S + 4) // 9)
    assert int(((S + 4) * 7282).max()) < (1 << 24)
