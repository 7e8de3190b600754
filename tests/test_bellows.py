"""Bellows-movement veto (scope row 8f #3; L3Localizer.cpp:292-390, :462-543): template matching terms on the GPU,
normalisation / sub-pixel maximum on the host, synthetic-frame ProcessFrame on the overlap ROI, subtraction and
re-thresholding -- product versus oracle."""
import os
import struct

import numpy as np
import pytest
from PIL import Image

from autobub3hs_amd import host, synth

rng = np.random.RandomState(31)


@pytest.fixture(scope="module", autouse=True)
def _built():
    host.build()


def exact_terms(img, tmpl):
    I = img.astype(np.uint64)
    T = tmpl.astype(np.uint64)
    th, tw = tmpl.shape
    rh, rw = img.shape[0] - th + 1, img.shape[1] - tw + 1
    num = np.zeros((rh, rw), np.uint64)
    w2 = np.zeros((rh, rw), np.uint64)
    for y in range(rh):
        for x in range(rw):
            w = I[y:y + th, x:x + tw]
            num[y, x] = (w * T).sum()
            w2[y, x] = (w * w).sum()
    return num, w2


def test_best_match_host_logic_equals_oracle(oracle):
    for trial in range(6):
        H, W, th, tw = 40 + trial, 64, 9 + trial, 12
        img = rng.randint(0, 256, (H, W)).astype(np.uint8)
        y0, x0 = rng.randint(0, H - th), rng.randint(0, W - tw)
        tmpl = np.clip(img[y0:y0 + th, x0:x0 + tw].astype(int) + rng.randint(-6, 7, (th, tw)), 0, 255).astype(np.uint8)
        if trial == 5:  # maximum in a corner: neighbours outside the result matrix
            tmpl = img[:th, :tw].copy()
        num, w2 = exact_terms(img, tmpl)
        bx, by = host.best_match(num, w2, tmpl)
        ox, oy = oracle.track_feature(img, tmpl)
        assert (bx, by) == (ox, oy)
        if trial < 5:
            assert abs(bx - x0) < 1 and abs(by - y0) < 1


def write_bmp8(path, img):
    H, W = img.shape
    stride = (W + 3) // 4 * 4
    pal = b"".join(struct.pack("<BBBB", i, i, i, 0) for i in range(256))
    data = b"".join(img[y].tobytes() + b"\0" * (stride - W) for y in range(H - 1, -1, -1))
    off = 14 + 40 + len(pal)
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off))
        f.write(struct.pack("<IiiHHIIiiII", 40, W, H, 1, 8, 0, len(data), 2835, 2835, 256, 0))
        f.write(pal + data)


@pytest.mark.gpu
def test_match_terms_kernel_exact():
    import ctypes as C

    import torch

    from autobub3hs_amd import _lib

    dev = "cuda:0"
    for (H, W, th, tw) in [(50, 70, 11, 13), (33, 40, 33, 7), (24, 1280, 9, 300), (17, 19, 1, 1)]:
        img = rng.randint(0, 256, (H, W)).astype(np.uint8)
        tmpl = rng.randint(0, 256, (th, tw)).astype(np.uint8)
        rh, rw = H - th + 1, W - tw + 1
        d_img, d_t = torch.from_numpy(img).to(dev), torch.from_numpy(tmpl).to(dev)
        num = torch.zeros((rh, rw), dtype=torch.int64, device=dev)
        w2 = torch.zeros((rh, rw), dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().abub_match_ccorr_dev(d_img.data_ptr(), W, H, d_t.data_ptr(), tw, th, num.data_ptr(),
                                                   w2.data_ptr(), torch.cuda.current_stream().cuda_stream))
        en, ew = exact_terms(img, tmpl)
        assert np.array_equal(num.cpu().numpy().astype(np.uint64), en)
        assert np.array_equal(w2.cpu().numpy().astype(np.uint64), ew)


def bellows_event(W, H, F, t0, shift):
    """A textured 'bellows' block that starts creeping at frame t0 (one pixel per frame, `shift` steps... then
    keeps moving): the trigger fires and every genesis contour falls inside the bellows mask."""
    spec = synth.EventSpec(F)
    fr = synth.render_event(W, H, spec, 77, 0).astype(int)
    yy, xx = np.mgrid[:50, :30]
    tex = (60 + 50 * ((yy // 5 + xx // 5) % 2) + 25 * np.sin(xx / 2.0) + 20 * np.cos(yy / 3.0)).astype(int)
    bx0, by0 = 140, 40
    for f in range(F):
        x = bx0 + (min(f - t0 + 1, 8) if f >= t0 else 0) * (1 if shift > 0 else -1)
        fr[f, by0:by0 + 50, x:x + 30] = tex
    return np.clip(fr, 0, 255).astype(np.uint8), np.clip(tex, 0, 255).astype(np.uint8), (bx0, by0)


@pytest.mark.gpu
@pytest.mark.parametrize("with_template", [True, False])
def test_bellows_veto_event_parity(tmp_path, oracle, with_template):
    W, H, F, t0 = 200, 120, 24, 12
    fr, tex, (bx0, by0) = bellows_event(W, H, F, t0, shift=2)
    tr = synth.training_pairs(W, H, 8, 0, F)
    # the training frames must contain the bellows block too (it is part of the background)
    for k in range(len(tr)):
        tr[k, by0:by0 + 50, bx0:bx0 + 30] = tex
    mu, sg = oracle.welford(tr)
    bel = np.zeros((H, W), np.uint8)
    bel[by0 - 10:by0 + 60, bx0 - 10:bx0 + 45] = 255
    write_bmp8(os.path.join(tmp_path, "cam0_bellows_mask.bmp"), bel)
    if with_template:
        Image.fromarray(tex).save(os.path.join(tmp_path, "cam0_bellows_template.png"))
    run = host.Run()
    run.set_model(0, mu, sg, len(tr))
    run.add_event(1, 0, fr)
    staged, state, bubbles, err = run.analyze(1, 0, maskdir=str(tmp_path))
    a = oracle.Analyzer(fr, mu, sg, len(tr), bel_mask=bel, bel_template=tex if with_template else None)
    staged_r, state_r, bubbles_r = a.any_cam_analysis()
    a.close()
    assert (staged, state) == (staged_r, state_r), (staged, state, staged_r, state_r, err)
    assert [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in bubbles] == \
           [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in bubbles_r]
    assert state["trig"] in (t0, t0 + 1) or staged != 0
    run.close()


@pytest.mark.gpu
def test_pipeline_falls_back_to_drop_in_path_for_bellows(tmp_path, oracle):
    import torch

    from autobub3hs_amd import hip

    dev = "cuda:0"
    W, H, F, t0 = 200, 120, 24, 12
    fr, tex, (bx0, by0) = bellows_event(W, H, F, t0, shift=2)
    tr = synth.training_pairs(W, H, 8, 0, F)
    for k in range(len(tr)):
        tr[k, by0:by0 + 50, bx0:bx0 + 30] = tex
    mu, sg = oracle.welford(tr)
    bel = np.zeros((H, W), np.uint8)
    bel[by0 - 10:by0 + 60, bx0 - 10:bx0 + 45] = 255
    write_bmp8(os.path.join(tmp_path, "cam0_bellows_mask.bmp"), bel)
    Image.fromarray(tex).save(os.path.join(tmp_path, "cam0_bellows_template.png"))
    plain = synth.render_event(W, H, synth.EventSpec(F, t0=10, bubbles=[(60, 60, 40)]), 5, 0)
    slab = np.stack([fr, plain])[:, None]  # [E=2][C=1][F][H][W]
    d_slab = torch.from_numpy(np.ascontiguousarray(slab)).to(dev)
    d_mu = torch.from_numpy(mu[None]).to(dev)
    d_sg = torch.from_numpy(sg[None]).to(dev)
    pipe = host.Pipeline(0, W, H, F, 2, 1, [len(tr)], nthreads=2, maskdir=str(tmp_path))
    pipe.run(d_slab, d_mu, hip.sigma6(d_sg), torch.cuda.current_stream().cuda_stream, sigma=d_sg)
    for s, stack in enumerate((fr, plain)):
        staged, state, bubbles, err = pipe.result(s)
        a = oracle.Analyzer(stack, mu, sg, len(tr), bel_mask=bel, bel_template=tex)
        ref = a.any_cam_analysis()
        a.close()
        assert (staged, state) == (ref[0], ref[1]), (s, staged, state, ref[0], ref[1], err)
        assert [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in bubbles] == \
               [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in ref[2]]
    pipe.close()
