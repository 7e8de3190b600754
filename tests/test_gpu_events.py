"""End-to-end parity on the GPU: whole (event, camera) analyses through the C++ mirror of the reference
API (Trainer / L3Localizer driven like AutoBubStart3.cpp's AnyCamAnalysis) versus the CPU oracle.
Integers, status codes and boxes must match bit for bit; centroids/radii within 1e-4 (they are the
same double arithmetic on identical polygons)."""
import os
import struct

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from autobub3hs_amd import host, synth  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


def write_bmp8(path, img):
    """8-bit palettised BMP (grey palette), bottom-up -- the format of cam_masks/*.bmp."""
    H, W = img.shape
    stride = (W + 3) // 4 * 4
    pal = b"".join(struct.pack("<BBBB", i, i, i, 0) for i in range(256))
    data = b"".join(img[y].tobytes() + b"\0" * (stride - W) for y in range(H - 1, -1, -1))
    off = 14 + 40 + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off))
        f.write(struct.pack("<IiiHHIIiiII", 40, W, H, 1, 8, 0, len(data), 2835, 2835, 256, 0))
        f.write(pal + data)


def write_bmp1(path, img):
    """1-bit BMP, palette {black, white}."""
    H, W = img.shape
    stride = ((W + 31) // 32) * 4
    rows = []
    for y in range(H - 1, -1, -1):
        bits = np.packbits((img[y] > 0).astype(np.uint8))
        rows.append(bits.tobytes() + b"\0" * (stride - len(bits)))
    data = b"".join(rows)
    pal = struct.pack("<BBBB", 0, 0, 0, 0) + struct.pack("<BBBB", 255, 255, 255, 0)
    off = 14 + 40 + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off))
        f.write(struct.pack("<IiiHHIIiiII", 40, W, H, 1, 1, 0, len(data), 2835, 2835, 2, 0))
        f.write(pal + data)


def compare(res, ref):
    staged, state, bubbles, err = res
    staged_r, state_r, bubbles_r = ref
    assert staged == staged_r, (staged, staged_r, err)
    assert state == state_r
    assert len(bubbles) == len(bubbles_r)
    for b, r in zip(bubbles, bubbles_r):
        assert len(b["desc"]) == len(r["desc"])
        for d, e in zip(b["desc"], r["desc"]):
            assert tuple(d[k] for k in "xywh") == tuple(e[k] for k in "xywh")
            for k in ("area", "radius", "m00", "m10", "m01", "cx", "cy"):
                if np.isnan(e[k]):
                    assert np.isnan(d[k])
                else:
                    assert abs(d[k] - e[k]) <= 1e-4 * max(1.0, abs(e[k])), (k, d[k], e[k])
        assert b["dz"] == pytest.approx(r["dz"], nan_ok=True)
        assert b["dzdt"] == pytest.approx(r["dzdt"], nan_ok=True) and b["drdt"] == pytest.approx(r["drdt"], nan_ok=True)


def oracle_event(oracle, fr, mu, sg, tss, **kw):
    a = oracle.Analyzer(fr, mu, sg, tss, **kw)
    out = a.any_cam_analysis()
    a.close()
    return out


def run_with_model(frames_by_event, mu, sg, tss, cam=0):
    run = host.Run()
    for ev, fr in frames_by_event.items():
        if isinstance(fr, tuple):
            run.add_event(ev, cam, fr[0], ok=fr[1])
        else:
            run.add_event(ev, cam, fr)
    run.set_model(cam, mu, sg, tss)
    return run


@pytest.mark.parametrize("W,H", [(640, 256), (1280, 200), (322, 150)])
def test_events_trained_on_gpu(oracle, W, H):
    F, cam, nev = 30, 0, 8
    run = host.Run()
    stacks = {}
    for e in range(nev):
        spec = synth.random_spec(W, H, F, e, cam, p_second=0.5, margin=30)
        stacks[e] = (spec, synth.render_event(W, H, spec, e, cam))
        run.add_event(e, cam, stacks[e][1])
    # one event with a disturbed second frame: vetoed by the 16-bin entropy test (Trainer.cpp:287)
    bad = synth.render_event(W, H, synth.EventSpec(F, t0=1, bubbles=[(W // 2, H // 2, 90)]), 99, cam)
    bad[1, H // 4:3 * H // 4, W // 4:3 * W // 4] = np.clip(bad[1, H // 4:3 * H // 4, W // 4:3 * W // 4].astype(int) + 60, 0, 255)
    run.add_event(99, cam, bad)
    st, tss, mu, sg = run.train(cam)
    good = [stacks[e][1][:2] for e in range(nev)]
    assert oracle.pair_entropy16(bad[1], bad[0]) > 0.0005
    assert all(oracle.pair_entropy16(g[1], g[0]) <= 0.0005 for g in good)
    mu_r, sg_r = oracle.welford(np.concatenate(good))
    assert st == 0 and tss == 2 * nev
    assert np.array_equal(mu, mu_r) and np.array_equal(sg, sg_r)
    for e in range(nev):
        spec, fr = stacks[e]
        res = run.analyze(e, cam)
        ref = oracle_event(oracle, fr, mu_r, sg_r, tss)
        compare(res, ref)
        assert res[0] == 0 and res[1]["trig"] in (spec.t0, spec.t0 + 1)
    run.close()


def test_small_training_set_one_frame_offset(oracle):
    W, H, F = 640, 200, 28
    spec = synth.EventSpec(F, t0=13, bubbles=[(200, 100, -40)])
    fr = synth.render_event(W, H, spec, 3, 1)
    tr = synth.training_pairs(W, H, 2, 1, F)  # 4 frames < 6: threshold 5.0, one-frame offset, loc_thres 3
    mu, sg = oracle.welford(tr)
    run = run_with_model({5: fr}, mu, sg, len(tr), cam=1)
    compare(run.analyze(5, 1), oracle_event(oracle, fr, mu, sg, len(tr)))
    run.close()


def test_status_codes(oracle):
    W, H = 320, 128
    tr = synth.training_pairs(W, H, 6, 0, 20)
    mu, sg = oracle.welford(tr)
    quiet = synth.render_event(W, H, synth.EventSpec(20), 1, 0)
    short4 = quiet[:4]
    five = synth.render_event(W, H, synth.EventSpec(5, t0=2, bubbles=[(100, 60, 40)]), 2, 0)
    bub = synth.render_event(W, H, synth.EventSpec(20, t0=10, bubbles=[(100, 60, 40)]), 3, 0)
    ok = np.ones(20, np.uint8)
    ok[4] = 0
    run = run_with_model({1: quiet, 2: short4, 3: five, 4: (bub, ok)}, mu, sg, len(tr))
    r = run.analyze(1, 0)
    assert r[0] == -3
    compare(r, oracle_event(oracle, quiet, mu, sg, len(tr)))
    r = run.analyze(2, 0)
    assert r[0] == -9
    compare(r, oracle_event(oracle, short4, mu, sg, len(tr)))
    r = run.analyze(3, 0)
    assert r[0] == -8
    compare(r, oracle_event(oracle, five, mu, sg, len(tr)))
    r = run.analyze(4, 0)
    assert r[0] == -9
    compare(r, oracle_event(oracle, bub, mu, sg, len(tr), frame_ok=ok))
    run.close()


def test_noisy_sigma_zero_flicker_and_retry(oracle):
    W, H, F = 256, 96, 36
    sg = np.zeros((H, W), np.uint8)
    spec = synth.EventSpec(F, t0=22, bubbles=[(120, 50, -40)], flicker=9, flicker_adu=12)
    fr = synth.render_event(W, H, spec, 3, 1)
    run = run_with_model({1: fr}, fr[0], sg, 20, cam=1)
    res = run.analyze(1, 1)
    compare(res, oracle_event(oracle, fr, fr[0], sg, 20))
    # persistent +1 step without any blob above threshold: retried until the end (-3) or a late blob
    fr2 = synth.render_event(W, H, synth.EventSpec(30), 4, 1)
    fr2[12:] = np.clip(fr2[12:].astype(int) + 1, 0, 255).astype(np.uint8)
    run.add_event(2, 1, fr2)
    compare(run.analyze(2, 1), oracle_event(oracle, fr2, fr[0], sg, 20))
    run.close()


def test_masks_from_bmp_files(oracle, tmp_path):
    W, H, F = 640, 256, 30
    spec = synth.EventSpec(F, t0=12, bubbles=[(300, 120, 40), (520, 60, 40)])
    fr = synth.render_event(W, H, spec, 9, 0)
    tr = synth.training_pairs(W, H, 6, 0, F)
    mu, sg = oracle.welford(tr)
    fid = np.full((300, 700), 255, np.uint8)
    fid[100:140, 280:320] = 0  # first bubble outside the fiducial volume
    bel = np.zeros((300, 700), np.uint8)
    bel[40:80, 500:540] = 200  # second bubble inside the bellows region
    write_bmp1(os.path.join(tmp_path, "cam0_mask.bmp"), fid)
    write_bmp8(os.path.join(tmp_path, "cam0_bellows_mask.bmp"), bel)
    run = run_with_model({1: fr}, mu, sg, len(tr))
    res = run.analyze(1, 0, maskdir=str(tmp_path))
    ref = oracle_event(oracle, fr, mu, sg, len(tr), fid_mask=fid, bel_mask=bel)
    compare(res, ref)
    # only a bellows mask that covers everything: contours are re-found and kept (no template)
    os.remove(os.path.join(tmp_path, "cam0_mask.bmp"))
    write_bmp8(os.path.join(tmp_path, "cam0_bellows_mask.bmp"), np.full((300, 700), 255, np.uint8))
    res = run.analyze(1, 0, maskdir=str(tmp_path))
    ref = oracle_event(oracle, fr, mu, sg, len(tr), bel_mask=np.full((300, 700), 255, np.uint8))
    compare(res, ref)
    assert res[0] == 0 and len(res[2]) == 2
    run.close()


def test_many_random_events_parity(oracle):
    """30 seeded events with two cameras' worth of seeds, 1280-wide (fast kernel) frames."""
    W, H, F = 1280, 160, 41
    tr = synth.training_pairs(W, H, 10, 0, F)
    mu, sg = oracle.welford(tr)
    run = host.Run()
    run.set_model(0, mu, sg, len(tr))
    n_ok = 0
    for e in range(30):
        spec = synth.random_spec(W, H, F, 1000 + e, 0, p_second=0.4, p_none=0.15, p_flicker=0.3, margin=25)
        fr = synth.render_event(W, H, spec, 1000 + e, 0)
        run.add_event(e, 0, fr)
        res = run.analyze(e, 0)
        compare(res, oracle_event(oracle, fr, mu, sg, len(tr)))
        n_ok += res[0] == 0
    assert n_ok >= 20
    run.close()


@pytest.mark.parametrize("W,H", [(1280, 128), (322, 120)])  # fused-compaction path / generic-kernel path
def test_batched_pipeline_equals_oracle(oracle, W, H):
    """The run-level batched driver (HBM-resident slab, shared launches) must give exactly what the
    oracle gives per (event, camera): two cameras, one of them with a small training set."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    F, E, C = 41, 7, 2
    slab = np.zeros((E, C, F, H, W), np.uint8)
    for e in range(E):
        for c in range(C):
            spec = synth.random_spec(W, H, F, 500 + e, c, p_second=0.4, p_none=0.2, p_flicker=0.3, margin=25)
            slab[e, c] = synth.render_event(W, H, spec, 500 + e, c)
    # a stack whose first trigger yields no accepted bubble (persistent +1 step), to exercise the retry rounds
    quiet = synth.render_event(W, H, synth.EventSpec(F), 900, 0)
    quiet[12:] = np.clip(quiet[12:].astype(int) + 1, 0, 255)
    slab[E - 1, 0] = quiet
    tr0 = synth.training_pairs(W, H, 10, 0, F)
    tr1 = synth.training_pairs(W, H, 2, 1, F)  # 4 training frames: one-frame offset path
    models = [oracle.welford(tr0), oracle.welford(tr1)]
    tss = [len(tr0), len(tr1)]
    d_slab = torch.from_numpy(slab).to(dev)
    d_mu = torch.from_numpy(np.stack([m[0] for m in models])).to(dev)
    d_s6 = hip.sigma6(torch.from_numpy(np.stack([m[1] for m in models])).to(dev))
    pipe = host.Pipeline(0, W, H, F, E, C, tss, nthreads=4)
    for rep in range(2):  # a pipeline object is reusable
        pipe.run(d_slab, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
        for e in range(E):
            for c in range(C):
                staged, state, bubbles, err = pipe.result(e * C + c)
                ref = oracle_event(oracle, slab[e, c], models[c][0], models[c][1], tss[c])
                assert (staged, state) == (ref[0], ref[1]), (e, c, staged, state, ref[0], ref[1], err)
                assert len(bubbles) == len(ref[2])
                for b, r in zip(bubbles, ref[2]):
                    assert [tuple(d[k] for k in "xywh") for d in b["desc"]] == [tuple(d[k] for k in "xywh") for d in r["desc"]]
                    for d, q in zip(b["desc"], r["desc"]):
                        assert abs(d["cx"] - q["cx"]) <= 1e-4 and abs(d["cy"] - q["cy"]) <= 1e-4
    t = pipe.timing()
    assert t["rounds"] >= 1
    pipe.close()


@pytest.mark.parametrize("regime", ["default", "post_trigger_dense", "noisy"])
def test_lazy_trigger_search_equals_oracle_in_every_regime(oracle, monkeypatch, regime):
    """The pipeline's trigger search is lazy in two ways -- frame blocks evaluated on demand (ABUB_PIPE_LAZY / _BLOCK0 /
    _BLOCK) and the row machine on dense frames only for the jobs a search reaches (ABUB_PIPE_DEFER).  Whatever the
    combination, in quiet data, with everything behind the trigger dense, and with hot pixels everywhere (every frame
    dense: every reached frame is completed on demand), the results are the oracle's."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    W, H, F, E, C = 1280, 96, 41, 6, 2
    slab = np.zeros((E, C, F, H, W), np.uint8)
    for e in range(E):
        for c in range(C):
            spec = synth.random_spec(W, H, F, 700 + e, c, p_second=0.3, p_none=0.15, p_flicker=0.3, margin=25, regime=regime)
            slab[e, c] = synth.render_event(W, H, spec, 700 + e, c)
    models, tss = [], []
    for c in range(C):
        tr = np.concatenate([slab[e, c, :2] for e in range(E)])
        models.append(oracle.welford(tr))
        tss.append(len(tr))
    refs = {(e, c): oracle_event(oracle, slab[e, c], models[c][0], models[c][1], tss[c]) for e in range(E) for c in range(C)}
    d_slab = torch.from_numpy(slab).to(dev)
    d_mu = torch.from_numpy(np.stack([m[0] for m in models])).to(dev)
    d_s6 = hip.sigma6(torch.from_numpy(np.stack([m[1] for m in models])).to(dev))
    seen_on_demand = False
    for lazy, block0, block, defer in (("1", "", "", "1"), ("1", "12", "5", "1"), ("1", "12", "5", "0"), ("0", "", "", "1"),
                                       ("0", "", "", "0"), ("1", "30", "4", "1")):
        monkeypatch.setenv("ABUB_PIPE_LAZY", lazy)
        monkeypatch.setenv("ABUB_PIPE_DEFER", defer)
        for k, v in (("ABUB_PIPE_BLOCK0", block0), ("ABUB_PIPE_BLOCK", block)):
            if v:
                monkeypatch.setenv(k, v)
            else:
                monkeypatch.delenv(k, raising=False)
        pipe = host.Pipeline(0, W, H, F, E, C, tss, nthreads=4)
        pipe.run(d_slab, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
        for e in range(E):
            for c in range(C):
                staged, state, bubbles, err = pipe.result(e * C + c)
                ref = refs[(e, c)]
                assert (staged, state) == (ref[0], ref[1]), (regime, lazy, block0, defer, e, c, staged, state, ref[0], ref[1], err)
                assert [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in bubbles] == \
                       [[tuple(d[k] for k in "xywh") for d in r["desc"]] for r in ref[2]]
        t = pipe.timing()
        if lazy == "0":
            assert t["trigger_jobs"] == E * C * (F - 1)
        elif block0 == "12":
            assert t["trigger_jobs"] < E * C * (F - 1)
        seen_on_demand = seen_on_demand or t["jobs_completed_on_demand"] > 0
        if defer == "0":
            assert t["jobs_completed_on_demand"] == 0
        pipe.close()
    assert seen_on_demand or regime != "noisy"  # (hot pixels everywhere: every reached frame had to be completed)


def test_image_entropy_methods_free_function(oracle):
    import ctypes as C

    L = host.lib()
    L.abh_entropy_frame.restype = C.c_float
    L.abh_entropy_frame.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int]
    rs = np.random.RandomState(4)
    for shape in [(64, 96), (33, 47)]:
        img = ((rs.rand(*shape) < 0.2) * rs.randint(0, 256, shape)).astype(np.uint8)
        got = L.abh_entropy_frame(img.ctypes.data_as(C.POINTER(C.c_uint8)), shape[1], shape[0])
        assert got == oracle.entropy16(img)


def test_streamed_pipeline_equals_resident(oracle):
    """BASELINE configs[4] path: the run starts in pinned host memory and is uploaded group by group while earlier
    groups are processed; results must equal the HBM-resident run."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    W, H, F, E, C = 1280, 96, 41, 8, 2
    slab = np.zeros((E, C, F, H, W), np.uint8)
    for e in range(E):
        for c in range(C):
            spec = synth.random_spec(W, H, F, 800 + e, c, p_second=0.3, p_none=0.2, margin=25)
            slab[e, c] = synth.render_event(W, H, spec, 800 + e, c)
    models = [oracle.welford(synth.training_pairs(W, H, 8, c, F)) for c in range(C)]
    d_mu = torch.from_numpy(np.stack([m[0] for m in models])).to(dev)
    d_s6 = hip.sigma6(torch.from_numpy(np.stack([m[1] for m in models])).to(dev))
    h_slab = torch.from_numpy(slab).pin_memory()
    os.environ["ABUB_PIPE_GROUPS"] = "4"
    try:
        pipe = host.Pipeline(0, W, H, F, E, C, [16, 16], nthreads=4)
    finally:
        del os.environ["ABUB_PIPE_GROUPS"]
    pipe.run_host(h_slab, d_mu, d_s6)
    streamed = [pipe.result(s)[:3] for s in range(E * C)]
    pipe.run(h_slab.to(dev), d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
    resident = [pipe.result(s)[:3] for s in range(E * C)]
    assert repr(streamed) == repr(resident)
    for s in (0, 5, 11):
        ref = oracle_event(oracle, slab[s // C, s % C], models[s % C][0], models[s % C][1], 16)
        assert (streamed[s][0], streamed[s][1]) == (ref[0], ref[1])
    pipe.close()


@pytest.mark.parametrize("W,H", [(1280, 1024), (1680, 1050)])  # BASELINE.json's two camera geometries, full frames
def test_full_frame_geometries_equal_oracle(oracle, W, H):
    """Whole (event, camera) analyses at the real frame sizes: frames and the trained model are generated in HBM
    (synth is integer-only, so torch == numpy), the batched pipeline runs them, and every stack is checked against
    the oracle on the very same bytes (model included: K1 must equal the oracle's Welford)."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    F, E, C = 41, 3, 2
    d_slab = torch.empty((E, C, F, H, W), dtype=torch.uint8, device=dev)
    for e in range(E):
        for c in range(C):
            spec = synth.random_spec(W, H, F, 40 + e, c, p_second=0.5, p_none=0.0, p_flicker=0.5, margin=40)
            synth.render_event(W, H, spec, 40 + e, c, xp="torch", device=dev, out=d_slab[e, c])
    mus, sgs, tss = [], [], []
    for c in range(C):
        tr = synth.training_pairs(W, H, 6, c, F, xp="torch", device=dev)
        mu, sg = hip.train(tr, W, H)
        mu_o, sg_o = oracle.welford(tr.cpu().numpy())
        assert np.array_equal(mu.cpu().numpy(), mu_o) and np.array_equal(sg.cpu().numpy(), sg_o)
        mus.append(mu), sgs.append(sg), tss.append(tr.shape[0])
    d_mu, d_s6 = torch.stack(mus), hip.sigma6(torch.stack(sgs))
    pipe = host.Pipeline(0, W, H, F, E, C, tss, nthreads=4)
    pipe.run(d_slab, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
    slab = d_slab.cpu().numpy()
    n_bub = 0
    for e in range(E):
        for c in range(C):
            ref = oracle_event(oracle, slab[e, c], mus[c].cpu().numpy(), sgs[c].cpu().numpy(), tss[c])
            staged, state, bubbles, err = pipe.result(e * C + c)
            assert (staged, state) == (ref[0], ref[1]), (e, c, staged, state, ref[0], ref[1], err)
            assert len(bubbles) == len(ref[2])
            for b, r in zip(bubbles, ref[2]):
                assert [tuple(d[k] for k in "xywh") for d in b["desc"]] == [tuple(d[k] for k in "xywh") for d in r["desc"]]
                for d, q in zip(b["desc"], r["desc"]):
                    assert abs(d["cx"] - q["cx"]) <= 1e-4 and abs(d["cy"] - q["cy"]) <= 1e-4
                assert b["dzdt"] == pytest.approx(r["dzdt"], nan_ok=True) and b["drdt"] == pytest.approx(r["drdt"], nan_ok=True)
            n_bub += len(ref[2])
    assert n_bub >= E * C  # every stack holds at least one bubble
    pipe.close()


def test_candidate_list_grows_on_overflow(oracle):
    """A candidate-list capacity far below what the batch needs: the pipeline must notice the overflow (the kernels
    keep counting), grow the lists and redo the batch -- same results as with the default capacity."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    W, H, F, E, C = 1280, 96, 41, 4, 1
    slab = np.zeros((E, C, F, H, W), np.uint8)
    for e in range(E):
        spec = synth.random_spec(W, H, F, 300 + e, 0, p_second=0.5, margin=25)
        slab[e, 0] = synth.render_event(W, H, spec, 300 + e, 0)
    mu, sg = oracle.welford(synth.training_pairs(W, H, 8, 0, F))
    d_slab = torch.from_numpy(slab).to(dev)
    d_mu = torch.from_numpy(mu[None]).to(dev)
    d_s6 = hip.sigma6(torch.from_numpy(sg[None]).to(dev))
    res = []
    for cap in ("64", None):
        if cap:
            os.environ["ABUB_PIPE_PAIRCAP"] = cap
        try:
            pipe = host.Pipeline(0, W, H, F, E, C, [16], nthreads=2)
        finally:
            os.environ.pop("ABUB_PIPE_PAIRCAP", None)
        pipe.run(d_slab, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
        assert pipe.timing()["pairs"] > 64
        res.append([pipe.result(s)[:3] for s in range(E * C)])
        pipe.close()
    assert repr(res[0]) == repr(res[1])
    ref = oracle_event(oracle, slab[0, 0], mu, sg, 16)
    assert (res[0][0][0], res[0][0][1]) == (ref[0], ref[1])


def test_pipeline_ring_batches_in_flight(oracle):
    """Three batches in flight on a ring of three pipelines (host stages of one under the GPU stages of the next):
    every batch must come out exactly as from a single pipeline, whichever pipeline object served it."""
    from autobub3hs_amd import hip

    dev = "cuda:0"
    W, H, F, E, C = 1280, 96, 41, 3, 2
    batches = []
    for b in range(5):
        slab = np.zeros((E, C, F, H, W), np.uint8)
        for e in range(E):
            for c in range(C):
                spec = synth.random_spec(W, H, F, 600 + 10 * b + e, c, p_second=0.3, p_none=0.2, margin=25)
                slab[e, c] = synth.render_event(W, H, spec, 600 + 10 * b + e, c)
        batches.append(slab)
    models = [oracle.welford(synth.training_pairs(W, H, 8, c, F)) for c in range(C)]
    d_mu = torch.from_numpy(np.stack([m[0] for m in models])).to(dev)
    d_s6 = hip.sigma6(torch.from_numpy(np.stack([m[1] for m in models])).to(dev))
    d_batches = [torch.from_numpy(b).to(dev) for b in batches]
    single = host.Pipeline(0, W, H, F, E, C, [16, 16], nthreads=2)
    want = []
    for db in d_batches:
        single.run(db, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
        want.append(repr([single.result(s)[:3] for s in range(E * C)]))
    single.close()
    ring = host.PipelineRing(3, 0, W, H, F, E, C, [16, 16], nthreads=2)
    got = {}
    tms = ring.run_batches(d_batches, d_mu, d_s6, torch.cuda.current_stream().cuda_stream,
                           on_done=lambda k, p: got.__setitem__(k, repr([p.result(s)[:3] for s in range(E * C)])))
    ring.close()
    assert len(tms) == 5 and [got[k] for k in range(5)] == want
    ref = oracle_event(oracle, batches[4][1, 1], models[1][0], models[1][1], 16)
    assert eval(got[4], {"nan": float("nan")})[1 * C + 1][:2] == (ref[0], ref[1])


def test_private_frame_statistics_a8(oracle):
    """AnalyzerUnit::calculateEntropyFrame (128 bins) / calculateEntropySignificance and the image
    overload of calculateSignificanceFrame (AnalyzerUnit.cpp:386-504) are private and never called upstream; reached
    here through the abub::AnalyzerProbe friend.  The histogram comes from the GPU (K1b), the statistics from
    hostlogic; compared with the oracle's orc_entropy128 / orc_significance and a plain restatement of the z-score."""
    W, H, n = 320, 200, 7
    rs = np.random.RandomState(8)
    imgs = np.zeros((n, H, W), np.uint8)
    for k in range(n):
        m = rs.rand(H, W) < 0.002 * (k + 1)
        imgs[k][m] = rs.randint(1, 40 + 30 * k, int(m.sum()))
    imgs[3, 50:70, 100:130] = rs.randint(0, 256, (20, 30))
    run = host.Run()
    run.add_event(5, 0, np.zeros((8, H, W), np.uint8))
    mu = np.full((H, W), 50, np.uint8)
    sg = np.ones((H, W), np.uint8)
    tss = 12
    run.set_model(0, mu, sg, tss)
    got = run.probe_frame_stats(5, 0, imgs)
    a = oracle.Analyzer(np.zeros((8, H, W), np.uint8), mu, sg, tss)
    hist_e = []
    for k in range(n):
        e = np.float32(oracle.entropy128(imgs[k]))
        assert np.float32(got[k, 0]) == e, k
        hist_e.append(float(e))
        mean = sum(hist_e) / len(hist_e)              # CalcMean / CalcStdDev<double> (AnalyzerUnit.cpp:514-532)
        sd = np.sqrt(np.float64(sum(v * v for v in hist_e)) / len(hist_e) - mean * mean)
        with np.errstate(all="ignore"):
            z = (np.float64(e) - mean) / sd
        assert (np.isnan(z) and np.isnan(got[k, 1])) or z == got[k, 1] or abs(z - got[k, 1]) <= 1e-9 * abs(z), (k, z, got[k, 1])
        s = a.significance(oracle.hist256(imgs[k]), True)
        assert (np.isnan(s) and np.isnan(got[k, 2])) or s == got[k, 2], (k, s, got[k, 2])
    a.close()
    run.close()
