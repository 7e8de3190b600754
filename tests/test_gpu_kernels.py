"""GPU parity tests proper: every HIP kernel, called through the C-ABI (device-pointer layer), must
match the CPU oracle bit for bit on the same seeded inputs."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from autobub3hs_amd import hip, synth  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _k2_defaults():
    """Every test starts from (and leaves behind) the default K2 launcher options."""
    yield
    hip.k2_set_option("bound", 1)
    hip.k2_set_option("chain", -1)
    hip.k2_set_option("budget", 1024)
    hip.k2_set_option("split", 1)
    hip.k2_set_option("list", 0)
    hip.k2_set_option("wg", -1)
    hip.k2_set_option("sync", -1)
    hip.k2_set_option("scanpf", -1)


def rnd_frames(rs, n, H, W, base=None, amp=12):
    if base is None:
        base = rs.randint(30, 200, (H, W))
    fr = base[None] + rs.randint(-amp, amp + 1, (n, H, W))
    return np.clip(fr, 0, 255).astype(np.uint8)


def oracle_hists(oracle, frames, sigma, jobs):
    Ds, hs = [], []
    for (c, r, m, o) in jobs:
        D = oracle.process_frame(frames[c], frames[r], sigma[m])
        Ds.append(D)
        hs.append(oracle.hist256(D))
    return np.stack(Ds), np.stack(hs)


K2_SHAPES = [
    (64, 1280, 0), (50, 1680, 0), (37, 256, 8), (40, 512, 16), (33, 768, 0), (16, 1024, 8),
    (9, 1536, 0), (21, 2048, 8), (64, 100, 16),      # fast path NDW = 5,7,1,2,3,4,6,8,1(25 lanes)
    (37, 53, 0), (5, 5, 0), (12, 6, 0), (1, 16, 0), (2, 8, 0), (3, 4, 0),  # generic / degenerate
]


@pytest.mark.parametrize("H,W,R", K2_SHAPES)
def test_k2_diff_hist_parity(oracle, H, W, R):
    rs = np.random.RandomState(H * 10007 + W)
    n = 6
    frames = rnd_frames(rs, n, H, W)
    sigma = rs.randint(0, 3, (2, H, W)).astype(np.uint8)
    jobs = [(i, max(i - 2, 0), i % 2, k) for k, i in enumerate(range(1, n))]
    Dref, href = oracle_hists(oracle, frames, sigma, jobs)
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    assert np.array_equal(s6.cpu().numpy(), np.minimum(6 * sigma.astype(int), 255).astype(np.uint8))
    j_d = hip.make_jobs(jobs, DEV)
    # bound = 1: bound-and-verify pass (scan + exact groups + handed-over rows), bound = 0: the row machine alone;
    # budget 8 fills the per-chunk LDS lists in the middle of chunks (the rest of the chunk is handed to the row
    # machine); list = 1: the listed suspects go to the launch's global list and sus_tail_list evaluates them,
    # list = 0: the scanning waves evaluate their suspects themselves
    for bound, budget, lst in ((1, 1024, 1), (0, 1024, 1), (1, 8, 1), (1, 1024, 0), (1, 8, 0)):
        hip.k2_set_option("bound", bound)
        hip.k2_set_option("budget", budget)
        hip.k2_set_option("list", lst)
        for store in (True, False):
            hist, diff = hip.diff_hist(f_d, s6, j_d, W, H, store=store, rows_per_chunk=R)
            torch.cuda.synchronize()
            assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href), (H, W, store, bound, budget, lst)
            if store:
                assert np.array_equal(diff.cpu().numpy(), Dref), (H, W, bound, budget, lst)


def test_k2_extreme_values(oracle):
    # saturation everywhere: 255 vs 0 with sigma 0 / large sigma, constant planes
    H, W = 24, 1280
    frames = np.zeros((3, H, W), np.uint8)
    frames[1] = 255
    frames[2, ::2, ::3] = 255
    sigma = np.zeros((2, H, W), np.uint8)
    sigma[1] = 50
    jobs = [(1, 0, 0, 0), (0, 1, 0, 1), (2, 0, 0, 2), (2, 1, 0, 3), (1, 0, 1, 4), (2, 1, 1, 5)]
    Dref, href = oracle_hists(oracle, frames, sigma, jobs)
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    hist, diff = hip.diff_hist(f_d, s6, hip.make_jobs(jobs, DEV), W, H, store=True)
    assert np.array_equal(diff.cpu().numpy(), Dref)
    assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href)
    hist, _ = hip.diff_hist(f_d, s6, hip.make_jobs(jobs, DEV), W, H, store=False)  # trigger-only kernel
    assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href)



@pytest.mark.parametrize("W,H", [(1280, 40), (1680, 30), (512, 24)])
def test_k2_chained_scan_decision_boundary_and_saturation(oracle, W, H):
    """The chained scan (v_sad_u8 masses with HI = min(r + s, 255), LO = sat(r - s) taken from every other frame of the
    chain) on inputs that sit on both sides of every decision: pixel values at 0 / 255 with
    sigma6 from 0 to 255 (r + s and r - s saturate), isolated supra-threshold pixels of excess 1..5 and small clusters
    (bound = 21 vs 22), at the image edges, in every lane segment; stride-1 and stride-2 chains of odd and even length.
    Histograms and stored D must equal the oracle's for every jobs-per-wave / workgroup setting."""
    rs = np.random.RandomState(W + 7 * H)
    n = 11
    base = rs.randint(0, 256, (H, W)).astype(np.int64)
    base[:, : W // 8] = 0            # c, r at the low rail
    base[:, W // 8: W // 4] = 255    # ... and at the high rail
    frames = np.repeat(base[None], n, 0)
    for f in range(n):
        k = rs.randint(100, 500)
        ys, xs = rs.randint(0, H, k), rs.randint(0, W, k)
        frames[f, ys, xs] += rs.choice([-1, 1], k) * rs.randint(1, 7, k)
        for _ in range(20):  # tight clusters whose masses cross the bound only together
            y, x = rs.randint(0, H - 1), rs.randint(0, W - 2)
            frames[f, y, x] += rs.randint(1, 4)
            frames[f, y + rs.randint(0, 2), x + rs.randint(0, 3)] += rs.randint(1, 4)
        for (y, x) in [(0, 0), (0, 1), (1, 0), (H - 1, W - 1), (H - 2, W - 1), (H - 1, W - 2), (0, W - 1), (H - 1, 0)]:
            frames[f, y, x] += rs.randint(-6, 7)
        if f % 3 == 2:  # large excursions: |c - r| up to 255
            ys, xs = rs.randint(0, H, 60), rs.randint(0, W, 60)
            frames[f, ys, xs] = rs.choice([0, 255], 60)
    frames = np.clip(frames, 0, 255).astype(np.uint8)
    sigma = np.zeros((2, H, W), np.uint8)
    sigma[1] = rs.choice([0, 0, 1, 1, 2, 7, 20, 42, 43, 255], (H, W))  # sigma6 = 0 .. 252, 255 (saturated)
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    for model in (0, 1):
        for off in (1, 2):
            jl = [(i, max(i - off, 0), model, i - 1) for i in range(1, n)]
            Dref, href = oracle_hists(oracle, frames, sigma, jl)
            jobs = hip.make_jobs(jl, DEV)
            for pf, K, wg, sync in ((1, 4, 1, 0), (2, 4, 2, 1), (1, 2, 4, 0), (-1, -1, -1, -1)):
                hip.k2_set_option("scanpf", pf)
                hip.k2_set_option("chain", K)
                hip.k2_set_option("wg", wg)
                hip.k2_set_option("sync", sync)
                hist, D = hip.diff_hist(f_d, s6, jobs, W, H, store=True, chain=(n - 1, off))
                assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href), (model, off, pf, K, wg, sync)
                assert np.array_equal(D.cpu().numpy(), Dref), (model, off, pf, K, wg, sync)
                hist, _ = hip.diff_hist(f_d, s6, jobs, W, H, store=False, chain=(n - 1, off))
                assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href), (model, off, pf, K, wg, sync)


@pytest.mark.parametrize("H,W,R", [(96, 1280, 0), (70, 1680, 16), (40, 256, 8), (33, 2048, 0), (64, 100, 16)])
def test_k2_trigger_only_bound_and_verify(oracle, H, W, R):
    """The trigger-only kernel proves most rows zero from a bound and recomputes the rest exactly.  Inputs that sit
    on both sides of the decision: isolated supra-threshold pixels of value 1..5 (36*3 = 108 < 128 <= 36*4), pairs and
    small clusters whose sums cross 128 only together, the same at the image edges and corners (reflected taps),
    and at chunk boundaries; sigma = 0 so that X = |cur - ref| exactly."""
    rs = np.random.RandomState(W * 31 + H)
    n = 7
    ref = rs.randint(20, 200, (H, W)).astype(np.int64)
    frames = np.repeat(ref[None], n, 0)
    for f in range(1, n):
        k = rs.randint(40, 400)
        ys, xs = rs.randint(0, H, k), rs.randint(0, W, k)
        frames[f, ys, xs] += rs.choice([-1, 1], k) * rs.randint(1, 6, k)
        for _ in range(12):  # tight clusters: two / three small values next to each other
            y, x = rs.randint(0, H - 1), rs.randint(0, W - 2)
            frames[f, y, x] += rs.randint(1, 4)
            frames[f, y + rs.randint(0, 2), x + rs.randint(0, 3)] += rs.randint(1, 4)
        for (y, x) in [(0, 0), (0, 1), (1, 0), (H - 1, W - 1), (H - 2, W - 1), (H - 1, W - 2), (0, W - 1), (H - 1, 0)]:
            frames[f, y, x] += rs.randint(-4, 5)
        if R:
            for y in range(R - 2, H, R):  # around chunk boundaries
                frames[f, min(y + rs.randint(0, 4), H - 1), rs.randint(0, W)] += rs.randint(3, 6)
    frames = np.clip(frames, 0, 255).astype(np.uint8)
    sigma = np.zeros((1, H, W), np.uint8)
    jobs = [(i, 0, 0, i - 1) for i in range(1, n)]
    _, href = oracle_hists(oracle, frames, sigma, jobs)
    assert 0 < (href[:, 1:].sum(1) > 0).sum()  # some frames show something, most pixels show nothing
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    hist, _ = hip.diff_hist(f_d, s6, hip.make_jobs(jobs, DEV), W, H, store=False, rows_per_chunk=R)
    assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href)
    # store mode of the same pass: the scan writes the zero rows, the exact groups their pixels
    Dref, _ = oracle_hists(oracle, frames, sigma, jobs)
    for budget, lst in ((1024, 1), (16, 1), (1024, 0), (16, 0)):
        hip.k2_set_option("budget", budget)
        hip.k2_set_option("list", lst)
        hist, D = hip.diff_hist(f_d, s6, hip.make_jobs(jobs, DEV), W, H, store=True, rows_per_chunk=R)
        assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href)
        assert np.array_equal(D.cpu().numpy(), Dref), (budget, lst)
        hist, _ = hip.diff_hist(f_d, s6, hip.make_jobs(jobs, DEV), W, H, store=False, rows_per_chunk=R)
        assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href), (budget, lst)


def test_k2_stack_jobs_and_synthetic_event(oracle):
    W, H, F = 1280, 96, 16
    spec = synth.EventSpec(F, t0=9, bubbles=[(400, 40, 40), (900, 70, -40)])
    fr = synth.render_event(W, H, spec, 5, 0)
    tr = synth.training_pairs(W, H, 6, 0, F)
    mu, sg = oracle.welford(tr)
    f_d = torch.from_numpy(fr).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sg[None]).to(DEV))
    jobs = hip.stack_jobs(1, F, 1, F - 1, 2, 1, DEV)
    jn = jobs.cpu().numpy()
    assert [tuple(r) for r in jn] == [(i, max(i - 2, 0), 0, i - 1) for i in range(1, F)]
    hist, _ = hip.diff_hist(f_d, s6, jobs, W, H)
    _, href = oracle.bench_trigger_pass(fr, sg, 2, 1, F - 1, want_hists=True)
    assert np.array_equal(hist.cpu().numpy().astype(np.uint32), href)
    assert href[spec.t0 - 1, 1:].sum() > 0  # the bubble is visible in the genesis frame


def test_k2_roi_parity(oracle):
    H, W = 60, 72
    rs = np.random.RandomState(3)
    frames = rnd_frames(rs, 2, H, W, amp=30)
    sigma = rs.randint(0, 3, (H, W)).astype(np.uint8)
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    for roi in [(7, 11, 23, 17), (0, 0, W, H), (70, 58, 2, 2), (5, 5, 1, 9)]:
        D, h = hip.diff_roi(f_d, 0, 1, s6, W, H, roi)
        Dref = oracle.process_frame(frames[0], frames[1], sigma, roi=roi)
        assert np.array_equal(D.cpu().numpy(), Dref), roi
        assert np.array_equal(h.cpu().numpy().astype(np.uint32), oracle.hist256(Dref))


@pytest.mark.parametrize("N,H,W", [(2, 16, 32), (13, 9, 11), (40, 32, 64), (7, 5, 3), (1, 4, 8)])
def test_k1_welford_parity(oracle, N, H, W):
    rs = np.random.RandomState(N * 31 + W)
    st = rs.randint(0, 256, (N, H, W)).astype(np.uint8)
    if N == 40:
        st = rnd_frames(rs, N, H, W, amp=3)  # realistic: small sigma, truncation boundaries
    mu_r, sg_r = oracle.welford(st)
    mu, sg = hip.train(torch.from_numpy(st).to(DEV), W, H)
    assert np.array_equal(mu.cpu().numpy(), mu_r)
    assert np.array_equal(sg.cpu().numpy(), sg_r)
    # with an index list (training set = frames 0,1 of the good events only)
    if N >= 4:
        idx = np.array([0, 1, N - 2, N - 1], np.int32)
        mu_r, sg_r = oracle.welford(st[idx])
        mu, sg = hip.train(torch.from_numpy(st).to(DEV), W, H, idx=torch.from_numpy(idx).to(DEV))
        assert np.array_equal(mu.cpu().numpy(), mu_r) and np.array_equal(sg.cpu().numpy(), sg_r)


def test_k1b_pair_hist_parity(oracle):
    H, W = 48, 80
    rs = np.random.RandomState(5)
    fr = rnd_frames(rs, 6, H, W, amp=40)
    pairs = [(1, 0, 0, 0), (3, 2, 0, 1), (5, 4, 0, 2), (0, 0, 0, 3)]
    h = hip.pair_hist(torch.from_numpy(fr).to(DEV), hip.make_jobs(pairs, DEV), W, H).cpu().numpy()
    for (a, b, _, o) in pairs:
        d = np.clip(fr[a].astype(int) - fr[b].astype(int), 0, 255).astype(np.uint8)
        assert np.array_equal(h[o].astype(np.uint32), oracle.hist256(d))


@pytest.mark.parametrize("H,W", [(40, 56), (3, 3), (17, 5), (64, 1280), (1, 7), (50, 1680), (9, 2048), (33, 768), (2, 512), (1, 256)])
def test_k3_posttrig_parity(oracle, H, W):
    rs = np.random.RandomState(H + W)
    fr = rnd_frames(rs, 3, H, W, amp=25)
    mu = rs.randint(0, 256, (2, H, W)).astype(np.uint8)
    mu[0] = np.clip(fr[0].astype(int) + rs.randint(-2, 3, (H, W)), 0, 255)
    sg = rs.randint(0, 3, (2, H, W)).astype(np.uint8)
    jobs = [(0, 0, 0, 0), (1, 0, 1, 1), (2, 0, 0, 2)]
    hist, img = hip.posttrig(torch.from_numpy(fr).to(DEV), torch.from_numpy(mu).to(DEV),
                             hip.sigma6(torch.from_numpy(sg).to(DEV)), hip.make_jobs(jobs, DEV), W, H)
    for (c, _, m, o) in jobs:
        O = oracle.posttrig_frame(fr[c], mu[m], sg[m])
        assert np.array_equal(img[o].cpu().numpy(), O)
        assert np.array_equal(hist[o].cpu().numpy().astype(np.uint32), oracle.hist256(O))


def test_k3_suspect_list_paths(oracle):
    """K3's suspect groups travel from the scanning waves to a global list that a second kernel evaluates
    (k3_tail_list).  Full-height frames with five narrow vertical stripes -- about 30 suspect groups in every row, so rows
    are never 'hot' -- fill the per-chunk LDS lists (flushes in mid-scan) and, with three jobs, overflow the global
    list's minimum capacity of 64 K entries (the scanning waves then evaluate the rest themselves); a tracked-bubble
    frame and a quiet frame go through the ordinary path.  Image, histogram and fused candidate list == oracle."""
    import ctypes as C

    from autobub3hs_amd import _lib

    W, H = 1280, 1024
    rs = np.random.RandomState(5)
    mu = rs.randint(40, 180, (1, H, W)).astype(np.uint8)
    sg = np.ones((1, H, W), np.uint8)
    sg[0, ::97, ::89] = 0  # a few hot pixels (sigma 0)
    fr = np.repeat(mu, 5, axis=0).astype(np.int32) + rs.randint(-1, 2, (5, H, W))
    for k in range(3):  # stripes, 9 px wide, different phase per frame
        for x in range(40 + 13 * k, W - 20, 250):
            fr[k, :, x:x + 9] += 30 + k
    yy, xx = np.ogrid[:H, :W]
    fr[3][(yy - 500) ** 2 + (xx - 700) ** 2 <= 45 ** 2] += 50
    fr = np.clip(fr, 0, 255).astype(np.uint8)
    jobs = [(k, 0, 0, k) for k in range(5)]
    f_d, mu_d = torch.from_numpy(fr).to(DEV), torch.from_numpy(mu).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sg).to(DEV))
    j_d = hip.make_jobs(jobs, DEV)
    hist, img = hip.posttrig(f_d, mu_d, s6, j_d, W, H)
    O = [oracle.posttrig_frame(fr[k], mu[0], sg[0]) for k in range(5)]
    for k in range(5):
        assert np.array_equal(img[k].cpu().numpy(), O[k]), k
        assert np.array_equal(hist[k].cpu().numpy().astype(np.uint32), oracle.hist256(O[k])), k
    assert int((O[0] > 0).sum()) > 5 * 7 * H and int((O[4] > 0).sum()) < 100
    L = _lib.lib()
    cap = 1 << 21
    pairs = torch.zeros((cap, 2), dtype=torch.int32, device=DEV)
    count = torch.zeros((1,), dtype=torch.int32, device=DEV)
    cthr = torch.tensor([3, 0, 5, 3, 3], dtype=torch.int32, device=DEV)
    hist3 = torch.empty((5, 256), dtype=torch.int32, device=DEV)
    _lib.check(L.abub_posttrig_compact_dev(f_d.data_ptr(), mu_d.data_ptr(), s6.data_ptr(), j_d.data_ptr(), 5, W, H,
                                           hist3.data_ptr(), None, cthr.data_ptr(), pairs.data_ptr(), cap,
                                           count.data_ptr(), 7, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    n = int(count.item())
    assert n < cap
    pr = pairs[:n].cpu().numpy().astype(np.uint32)
    assert torch.equal(hist3, hist)
    total = 0
    for k in range(5):
        sel = pr[(pr[:, 0] & 0xFFFFFF) == 7 + k]
        exp = np.flatnonzero(O[k].ravel() > int(cthr[k]))
        order = np.argsort(sel[:, 1])
        assert np.array_equal(sel[order, 1], exp), k
        assert np.array_equal((sel[order, 0] >> 24).astype(np.uint8), O[k].ravel()[exp]), k
        total += len(exp)
    assert total == n


def test_k4_foreground_compaction(oracle):
    H, W = 64, 80
    rs = np.random.RandomState(9)
    img = np.zeros((3, H, W), np.uint8)
    img[0, 10:20, 30:45] = rs.randint(1, 200, (10, 15))
    img[1] = rs.randint(0, 256, (H, W))
    thr = np.array([3, 250, 0], np.int32)
    idx, cnt = hip.fg_compact(torch.from_numpy(img).to(DEV), torch.from_numpy(thr).to(DEV), 4096)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    for k in range(3):
        exp = np.flatnonzero(img[k].ravel() > thr[k])
        assert cnt[k] == len(exp)
        assert np.array_equal(np.sort(idx[k, :cnt[k]]), exp)
    # overflow is reported through the true count
    idx, cnt = hip.fg_compact(torch.from_numpy(img[1:2]).to(DEV), torch.from_numpy(np.array([10], np.int32)).to(DEV), 64)
    assert cnt.item() == int((img[1] > 10).sum()) > 64


def test_full_size_properties():
    """BASELINE-size checks that need no oracle: linearity of the histogram total, D == 0 for identical
    frames, and store / no-store agreement on a 1280x1024 stack."""
    W, H, F = 1280, 1024, 8
    spec = synth.EventSpec(F, t0=4, bubbles=[(640, 512, 40)])
    fr = synth.render_event(W, H, spec, 11, 0, xp="torch", device=DEV)
    sg = torch.ones((1, H, W), dtype=torch.uint8, device=DEV)
    s6 = hip.sigma6(sg)
    jobs = hip.stack_jobs(1, F, 1, F - 1, 2, 1, DEV)
    h1, D = hip.diff_hist(fr, s6, jobs, W, H, store=True)
    h2, _ = hip.diff_hist(fr, s6, jobs, W, H, store=False)
    assert torch.equal(h1, h2)
    assert (h1.sum(1) == W * H).all()
    for k in range(F - 1):
        assert torch.equal(torch.bincount(D[k].flatten().long(), minlength=256).int(), h1[k])
    same = hip.make_jobs([(3, 3, 0, 0)], DEV)
    h, D0 = hip.diff_hist(fr, s6, same, W, H, store=True)
    assert int(h[0, 0]) == W * H and not D0.any()
    # the bubble (contrast 40 > 6 sigma) shows up from its genesis frame on and is centred correctly
    nz = D[spec.t0 - 1].nonzero()
    assert len(nz) > 0
    assert abs(nz[:, 0].float().mean().item() - 512) < 1.0 and abs(nz[:, 1].float().mean().item() - 640) < 1.0


def test_fused_compaction_lists(oracle):
    """K2 / K3 with the fused candidate list: entries = exactly the pixels with value > cthr, with their
    values, tagged slot_base + job.out; histograms unchanged; images not materialised."""
    import ctypes as C

    from autobub3hs_amd import _lib

    W, H = 1280, 96
    rs = np.random.RandomState(77)
    fr = rnd_frames(rs, 5, H, W, amp=14)
    fr[3, 40:60, 600:640] = np.clip(fr[3, 40:60, 600:640].astype(int) + 70, 0, 255)
    mu = fr[0:1].copy()
    sg = rs.randint(0, 2, (1, H, W)).astype(np.uint8)
    f_d = torch.from_numpy(fr).to(DEV)
    mu_d = torch.from_numpy(mu).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sg).to(DEV))
    L = _lib.lib()
    cap = 1 << 20
    pairs = torch.zeros((cap, 2), dtype=torch.int32, device=DEV)
    count = torch.zeros((1,), dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    # K2: two jobs, slots 0,1
    jobs2 = hip.make_jobs([(3, 1, 0, 0), (4, 2, 0, 1)], DEV)
    cthr2 = torch.tensor([2, 3], dtype=torch.int32, device=DEV)
    hist2 = torch.empty((2, 256), dtype=torch.int32, device=DEV)
    _lib.check(L.abub_diff_hist_compact_dev(f_d.data_ptr(), s6.data_ptr(), jobs2.data_ptr(), 2, W, H, hist2.data_ptr(),
                                            None, cthr2.data_ptr(), pairs.data_ptr(), cap, count.data_ptr(), 0, st))
    # K3: two jobs appended to the same list with slot_base 2
    jobs3 = hip.make_jobs([(3, 0, 0, 0), (4, 0, 0, 1)], DEV)
    cthr3 = torch.tensor([3, 3], dtype=torch.int32, device=DEV)
    hist3 = torch.empty((2, 256), dtype=torch.int32, device=DEV)
    _lib.check(L.abub_posttrig_compact_dev(f_d.data_ptr(), mu_d.data_ptr(), s6.data_ptr(), jobs3.data_ptr(), 2, W, H,
                                           hist3.data_ptr(), None, cthr3.data_ptr(), pairs.data_ptr(), cap,
                                           count.data_ptr(), 2, st))
    torch.cuda.synchronize()
    n = int(count.item())
    pr = pairs[:n].cpu().numpy().astype(np.uint32)
    imgs = [oracle.process_frame(fr[3], fr[1], sg[0]), oracle.process_frame(fr[4], fr[2], sg[0]),
            oracle.posttrig_frame(fr[3], mu[0], sg[0]), oracle.posttrig_frame(fr[4], mu[0], sg[0])]
    thr = [2, 3, 3, 3]
    total = 0
    for slot in range(4):
        sel = pr[(pr[:, 0] & 0xFFFFFF) == slot]
        exp = np.flatnonzero(imgs[slot].ravel() > thr[slot])
        order = np.argsort(sel[:, 1])
        assert np.array_equal(sel[order, 1], exp), slot
        assert np.array_equal((sel[order, 0] >> 24).astype(np.uint8), imgs[slot].ravel()[exp])
        total += len(exp)
    assert total == n and n > 100
    assert np.array_equal(hist2.cpu().numpy().astype(np.uint32), np.stack([oracle.hist256(imgs[0]), oracle.hist256(imgs[1])]))
    assert np.array_equal(hist3.cpu().numpy().astype(np.uint32), np.stack([oracle.hist256(imgs[2]), oracle.hist256(imgs[3])]))


@pytest.mark.parametrize("W,H", [(1280, 1024), (1680, 1050)])
def test_k2_trigger_only_equals_store_mode_full_size(W, H):
    """Size-independent cross-check at BASELINE's frame sizes: the trigger-only pass (bound scan + exact groups +
    hand-over) and the store-mode pass (full row machine) are different code paths and must give the same
    histograms, which must also be the histograms of the stored D.  The stack mixes quiet frames, growing bubbles
    (up to ~45 px radius), a flicker frame, a frame that differs everywhere (dense: the scan hands whole chunks over)
    and a frame with a dense band in the middle of a chunk (partial hand-over)."""
    F = 24
    spec = synth.EventSpec(F, t0=6, bubbles=[(W // 3, H // 2, 40), (2 * W // 3, H // 3, -40)], flicker=4)
    fr = synth.render_event(W, H, spec, 77, 0, xp="torch", device=DEV)
    fr[20] = torch.clamp(fr[20].to(torch.int16) + 25, 0, 255).to(torch.uint8)
    band = fr[22, 200:260].to(torch.int16)
    band[:, ::3] += 30
    fr[22, 200:260] = torch.clamp(band, 0, 255).to(torch.uint8)
    sg = torch.ones((1, H, W), dtype=torch.uint8, device=DEV)
    sg[0, :, W // 2:] = 2
    s6 = hip.sigma6(sg)
    jobs = hip.stack_jobs(1, F, 1, F - 1, 2, 1, DEV)
    h_trig, _ = hip.diff_hist(fr, s6, jobs, W, H, store=False)
    h_bstore, D_b = hip.diff_hist(fr, s6, jobs, W, H, store=True)  # bound-and-verify, store mode
    h_cstore, D_c = hip.diff_hist(fr, s6, jobs, W, H, store=True, chain=(F - 1, 2))  # chained scan, store mode
    hip.k2_set_option("chain", 3)
    h_c3, D_c3 = hip.diff_hist(fr, s6, jobs, W, H, store=True, chain=(F - 1, 2))
    # workgroups of several waves on one chain, with and without the soft sync; one or two rows fetched ahead
    for wg, sync, pf, K in ((4, 2, 1, 4), (2, 0, 2, 4), (1, 0, 1, 2), (4, 2, 2, 4), (1, 0, -1, -1)):
        hip.k2_set_option("wg", wg)
        hip.k2_set_option("sync", sync)
        hip.k2_set_option("scanpf", pf)
        hip.k2_set_option("chain", K)
        h_w, D_w = hip.diff_hist(fr, s6, jobs, W, H, store=True, chain=(F - 1, 2))
        h_wt, _ = hip.diff_hist(fr, s6, jobs, W, H, store=False, chain=(F - 1, 2))
        assert torch.equal(h_w, h_c3) and torch.equal(D_w, D_c3) and torch.equal(h_wt, h_c3), (wg, sync, pf, K)
    hip.k2_set_option("chain", 3)
    hip.k2_set_option("list", 1)  # suspects through the global list and sus_tail_list instead of the in-wave tails
    h_l1, D_l1 = hip.diff_hist(fr, s6, jobs, W, H, store=True, chain=(F - 1, 2))
    h_l1t, _ = hip.diff_hist(fr, s6, jobs, W, H, store=False)
    assert torch.equal(h_l1, h_c3) and torch.equal(D_l1, D_c3) and torch.equal(h_l1t, h_c3)
    hip.k2_set_option("list", 0)
    hip.k2_set_option("bound", 0)
    h_store, D = hip.diff_hist(fr, s6, jobs, W, H, store=True)  # the row machine alone
    torch.cuda.synchronize()
    assert torch.equal(h_trig, h_store) and torch.equal(h_bstore, h_store) and torch.equal(h_cstore, h_store)
    assert torch.equal(h_c3, h_store)
    assert torch.equal(D_b, D) and torch.equal(D_c, D) and torch.equal(D_c3, D)
    for k in (0, 5, 12, 19, 21, 22):
        assert torch.equal(torch.bincount(D[k].flatten().to(torch.int64), minlength=256).to(h_store.dtype), h_store[k])
    assert int(h_store[19, 1:].sum()) > W * H // 2  # the dense frame really is dense


@pytest.mark.parametrize("W,H,F,off", [(1280, 64, 12, 2), (1680, 40, 9, 2), (256, 48, 7, 1), (1280, 33, 6, 3), (512, 40, 5, 2),
                                       (1540, 24, 6, 2), (1400, 24, 6, 2), (1100, 24, 6, 2), (1792, 20, 5, 2), (1028, 20, 5, 1)])
def test_k2_chained_trigger_pass(oracle, W, H, F, off):
    """abub_diff_hist_chained_dev: the stack-structured job list (job i refs the cur frame of job i - off) scanned with
    shared row loads must give the histograms of the plain entry and of the oracle; a WRONG hint (stride or length that
    do not describe the list) is detected on the device and still gives the same histograms."""
    rs = np.random.RandomState(F * 131 + W)
    nst = 3
    frames = np.zeros((nst * F, H, W), np.uint8)
    for s in range(nst):
        base = rs.randint(30, 200, (H, W))
        for f in range(F):
            fr = base + rs.randint(-3, 4, (H, W))
            k = rs.randint(20, 200)
            fr[rs.randint(0, H, k), rs.randint(0, W, k)] += rs.randint(5, 12, k)
            if f >= F // 2:
                yy, xx = np.ogrid[:H, :W]
                fr = np.where((yy - H // 2) ** 2 + (xx - W // 3 - 40 * s) ** 2 <= (2 + 3 * (f - F // 2)) ** 2, fr + 40, fr)
            frames[s * F + f] = np.clip(fr, 0, 255)
    sigma = rs.randint(0, 3, (2, H, W)).astype(np.uint8)
    f_d = torch.from_numpy(frames).to(DEV)
    s6 = hip.sigma6(torch.from_numpy(sigma).to(DEV))
    jobs = hip.stack_jobs(nst, F, 1, F - 1, off, 2, DEV)
    plain, _ = hip.diff_hist(f_d, s6, jobs, W, H)
    jn = jobs.cpu().numpy()
    _, href = oracle_hists(oracle, frames, sigma, [tuple(r) for r in jn])
    assert np.array_equal(plain.cpu().numpy().astype(np.uint32), href)
    Dref, _ = oracle_hists(oracle, frames, sigma, [tuple(r) for r in jn])
    # chain = jobs per wave; split = the scan's lane mapping (whole 16/8/4-byte pieces per lane where the width allows
    # it -- 1280 {4,1}, 1400 {4,2}, 1540 / 1680 / 1792 {4,2,1}, 1100 {4,1} with a partial last segment -- or blocked)
    # wg = waves per workgroup (consecutive segments of one chain, DESIGN.md "chained scan"), sync = row steps a wave may run
    # ahead of its workgroup's slowest wave (0: never waits)
    # pf = rows the chained scan fetches ahead (2 exists for the split mapping at 5 dwords per lane; -1 = automatic)
    for K, split, lst, wg, sync, pf in ((2, 1, 1, 1, 0, 1), (4, 1, 1, 4, 2, 1), (2, 0, 1, 2, 1, 1), (-1, 1, 1, 4, 0, 2),
                                        (4, 1, 0, 3, 4, 2), (2, 0, 0, 4, 1, -1), (-1, 1, 0, -1, -1, -1), (4, 1, 0, 2, 2, 1),
                                        (4, 0, 1, 1, 0, -1)):
        hip.k2_set_option("scanpf", pf)
        hip.k2_set_option("chain", K)
        hip.k2_set_option("split", split)
        hip.k2_set_option("list", lst)  # 0: the scanning waves evaluate their suspects themselves
        hip.k2_set_option("wg", wg)
        hip.k2_set_option("sync", sync)
        for L, S in [(F - 1, off), (F - 1, off + 1), (F - 1, 1), ((F - 1) * nst, off), (1, 1)]:
            if ((F - 1) * nst) % L:
                continue
            got, _ = hip.diff_hist(f_d, s6, jobs, W, H, chain=(L, S))
            assert torch.equal(got, plain), (K, split, lst, wg, sync, pf, L, S)
            got, D = hip.diff_hist(f_d, s6, jobs, W, H, chain=(L, S), store=True)
            assert torch.equal(got, plain), (K, split, lst, wg, sync, pf, L, S)
            assert np.array_equal(D.cpu().numpy(), Dref), (K, split, lst, wg, sync, pf, L, S)


@pytest.mark.parametrize("W,H,F", [(1280, 96, 14), (1680, 64, 10), (512, 80, 9)])
def test_k2_deferred_pieces(W, H, F):
    """abub_diff_hist_chained_deferred_dev leaves the handed-over rows of dense frames to a later, per-job call: jobs it
    does not flag have their final histograms at once; flagged jobs get theirs from abub_diff_hist_pieces_dev, in any
    number of instalments; untouched flagged jobs stay partial.  Reference: the ordinary chained pass."""
    nst = 3
    spec = synth.EventSpec(F, t0=4, bubbles=[(W // 3, H // 2, 40)])
    fr = torch.cat([synth.render_event(W, H, spec, 40 + s, 0, xp="torch", device=DEV) for s in range(nst)])
    fr[F + 6] = torch.clamp(fr[F + 6].to(torch.int16) + 30, 0, 255).to(torch.uint8)      # a frame that differs everywhere
    band = fr[2 * F + 5, 20:50].to(torch.int16)
    band[:, ::2] += 25
    fr[2 * F + 5, 20:50] = torch.clamp(band, 0, 255).to(torch.uint8)                     # a dense band inside a chunk
    sg = torch.ones((1, H, W), dtype=torch.uint8, device=DEV)
    s6 = hip.sigma6(sg)
    jobs = hip.stack_jobs(nst, F, 1, F - 1, 2, 1, DEV)
    ref, _ = hip.diff_hist(fr, s6, jobs, W, H, chain=(F - 1, 2))
    hist, state = hip.diff_hist_deferred(fr, s6, jobs, W, H, chain=(F - 1, 2))
    torch.cuda.synchronize()
    inc = state[2].cpu().numpy().astype(bool)
    assert int(state[1].item()) > 0 and inc.any() and not inc.all()
    h = hist.cpu().numpy()
    r = ref.cpu().numpy()
    assert np.array_equal(h[~inc], r[~inc])                      # complete jobs: final
    assert not np.array_equal(h[inc], r[inc])                    # the others really are partial
    idx = np.flatnonzero(inc)
    first, rest = idx[: len(idx) // 2], idx[len(idx) // 2:]
    for part in (first, rest):
        want = torch.zeros((jobs.shape[0],), dtype=torch.uint8, device=DEV)
        want[torch.from_numpy(part).to(DEV)] = 1
        hip.diff_hist_pieces(fr, s6, jobs, W, H, hist, state, want)
        torch.cuda.synchronize()
        h = hist.cpu().numpy()
        assert np.array_equal(h[part], r[part])
        if part is first and len(rest):
            assert not np.array_equal(h[rest], r[rest])          # not asked for yet: untouched
    assert np.array_equal(hist.cpu().numpy(), r)


def test_scratch_release_and_reuse():
    """The trigger-only pass keeps its work list in library-owned scratch keyed by stream: releasing it (twice, and for
    a stream that never had one) is harmless, and the next launch on that stream allocates again and gives the same
    histograms."""
    from autobub3hs_amd import _lib

    W, H, F = 1280, 64, 8
    fr = synth.render_event(W, H, synth.EventSpec(F, t0=3, bubbles=[(300, 30, 40)]), 9, 0, xp="torch", device=DEV)
    s6 = hip.sigma6(torch.ones((1, H, W), dtype=torch.uint8, device=DEV))
    jobs = hip.stack_jobs(1, F, 1, F - 1, 2, 1, DEV)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        h1, _ = hip.diff_hist(fr, s6, jobs, W, H, chain=(F - 1, 2))
    side.synchronize()
    L = _lib.lib()
    assert L.abub_scratch_release(side.cuda_stream) == 0
    assert L.abub_scratch_release(side.cuda_stream) == 0
    assert L.abub_scratch_release(torch.cuda.Stream().cuda_stream) == 0
    with torch.cuda.stream(side):
        h2, _ = hip.diff_hist(fr, s6, jobs, W, H, chain=(F - 1, 2))
    side.synchronize()
    assert torch.equal(h1, h2) and int(h1[:, 1:].sum()) > 0
