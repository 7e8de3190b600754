"""Known-answer and cross-implementation tests of the oracle's significance / FindTriggerFrame /
LocalizeOMatic / AnyCamAnalysis restatement (README error-code table)."""
import math

import numpy as np
import pytest

import pyref
from autobub3hs_amd import synth

rng = np.random.RandomState(7)


def make_case(W=160, H=128, F=30, event=1, cam=0, n_train=20, noise_gain=1, **kw):
    spec = synth.random_spec(W, H, F, event, cam, **kw)
    fr = synth.render_event(W, H, spec, event, cam)
    tr = synth.training_pairs(W, H, n_train, cam, F)
    return spec, fr, tr


def test_significance_vs_python(oracle):
    P = 64 * 64
    for tss in (4, 20):
        a = oracle.Analyzer(np.zeros((6, 64, 64), np.uint8), np.zeros((64, 64), np.uint8),
                            np.zeros((64, 64), np.uint8), tss)
        ref = pyref.Sig(P, tss)
        for step in range(25):
            h = np.zeros(256, np.int64)
            nb = rng.randint(1, 6)
            vals = rng.randint(1, 30, nb)
            cnts = rng.randint(0, 40, nb)
            if step % 7 == 6:
                cnts = cnts * 15
            for v, c in zip(vals, cnts):
                h[v] += c
            h[0] = P - h.sum()
            store = (step % 3) != 2
            s1 = a.significance(h.astype(np.uint32), store)
            s2 = ref(h, store)
            assert (math.isnan(s1) and math.isnan(s2)) or s1 == s2, (step, s1, s2)
            assert a.state()["loc_thres"] == ref.loc_thres
        a.close()


def test_significance_first_frames_known(oracle):
    # first stored frame: mean == b, sd == 0 -> no contribution; second differing frame -> z-score of 2 samples
    P = 32 * 32
    a = oracle.Analyzer(np.zeros((6, 32, 32), np.uint8), np.zeros((32, 32), np.uint8),
                        np.zeros((32, 32), np.uint8), 20)
    h = np.zeros(256, np.uint32)
    h[0], h[5] = P - 10, 10
    assert a.significance(h, True) == 0.0
    h2 = np.zeros(256, np.uint32)
    h2[0], h2[5] = P - 30, 30
    # bin 5 samples {10,30}: mean 20, sd 10 -> (30-20)/10 = 1; bins 2..4 : 0==mean, sd 0 -> skipped
    assert a.significance(h2, True) == 1.0
    # not stored, count 100: (100-20)/10 = 8
    h3 = np.zeros(256, np.uint32)
    h3[0], h3[5] = P - 100, 100
    assert a.significance(h3, False) == 8.0
    # a never-seen bin while not storing: sd == 0, b != mean -> +inf
    h4 = np.zeros(256, np.uint32)
    h4[0], h4[9] = P - 1, 1
    assert a.significance(h4, False) == math.inf
    a.close()


def test_quiet_event_no_trigger(oracle):
    W, H, F = 96, 80, 12
    f = rng.randint(0, 256, (H, W)).astype(np.uint8)
    frames = np.repeat(f[None], F, 0)
    a = oracle.Analyzer(frames, f, np.zeros_like(f), 20)
    st = a.find_trigger(1)
    assert st["status"] == -3 and not st["ok"]
    assert np.nanmax(a.sig_trace()) == 0.0
    staged, st, bub = a.any_cam_analysis()
    assert staged == -3 and bub == []
    a.close()


def test_malformed_sequences(oracle):
    W, H = 64, 48
    f = rng.randint(0, 256, (4, H, W)).astype(np.uint8)
    a = oracle.Analyzer(f, f[0], np.zeros_like(f[0]), 20)
    st = a.find_trigger(1)
    assert st["status"] == -9 and not st["ok"]  # AnalyzerUnit.cpp:122-126
    a.close()
    # corrupt image in the sequence -> -9 (AnalyzerUnit.cpp:207-213)
    spec, fr, tr = make_case(F=20)
    mu, sg = oracle.welford(tr)
    ok = np.ones(20, np.uint8)
    ok[5] = 0
    a = oracle.Analyzer(fr, mu, sg, len(tr), frame_ok=ok)
    staged, st, _ = a.any_cam_analysis()
    assert staged == -9 and st["status"] == -9
    a.close()


def test_five_frames_localizer_refuses(oracle):
    # exactly 5 frames: trigger search runs, LocalizeOMatic refuses (L3Localizer.cpp:889) -> -8
    W, H = 96, 80
    spec = synth.EventSpec(5, t0=2, bubbles=[(40, 40, 40)])
    fr = synth.render_event(W, H, spec, 1, 0)
    tr = synth.training_pairs(W, H, 10, 0, 5)
    mu, sg = oracle.welford(tr)
    a = oracle.Analyzer(fr, mu, sg, len(tr))
    staged, st, bub = a.any_cam_analysis()
    assert st["status"] == 0 and st["trig"] == 2 and staged == -8 and bub == []
    a.close()


@pytest.mark.parametrize("event,tss_events", [(1, 20), (2, 20), (5, 2), (8, 20)])
def test_find_trigger_vs_python(oracle, event, tss_events):
    spec, fr, tr = make_case(event=event, n_train=tss_events, p_second=0.5)
    mu, sg = oracle.welford(tr)
    a = oracle.Analyzer(fr, mu, sg, len(tr))
    st = a.find_trigger(1)
    status, trig, ok, sobj = pyref.find_trigger(list(fr), sg, len(tr))
    assert st["status"] == status and st["ok"] == ok
    if status == 0:
        assert st["trig"] == trig
        assert trig in (spec.t0, spec.t0 + 1)
    assert st["loc_thres"] == sobj.loc_thres
    a.close()


def test_trigger_with_noisy_sigma_zero_and_flicker(oracle):
    # sigma = 0 and mu irrelevant: every noise excursion survives -> CUSUM statistics are exercised;
    # an LED flicker frame (+12 ADU on one frame) must be vetoed by the 2-frame look-ahead.
    W, H, F = 128, 96, 36
    spec = synth.EventSpec(F, t0=22, bubbles=[(60, 50, -40)], flicker=9, flicker_adu=12)
    fr = synth.render_event(W, H, spec, 3, 1)
    sg = np.zeros((H, W), np.uint8)
    a = oracle.Analyzer(fr, fr[0], sg, 20)
    st = a.find_trigger(1)
    status, trig, ok, sobj = pyref.find_trigger(list(fr), sg, 20)
    assert (st["status"], st["trig"]) == (status, trig) == (0, 22)
    tr = a.sig_trace()
    assert tr[9] > 3.5  # the flicker frame did exceed the threshold ...
    assert st["loc_thres"] == sobj.loc_thres
    a.close()


def test_retry_loop_runs_until_bubble_or_end(oracle):
    # a flicker that survives the look-ahead but yields no blob -> retried from the next frame
    W, H, F = 128, 96, 30
    spec = synth.EventSpec(F, t0=None)
    fr = synth.render_event(W, H, spec, 4, 0)
    fr[12:] = np.clip(fr[12:].astype(int) + 1, 0, 255).astype(np.uint8)  # persistent +1 step: no blob > thr
    sg = np.zeros((H, W), np.uint8)
    a = oracle.Analyzer(fr, fr[0], sg, 20)
    staged, st, bub = a.any_cam_analysis()
    assert staged in (-3, 0)
    if staged == -3:
        assert bub == []
    a.close()


def test_full_event_known_answer(oracle):
    spec, fr, tr = make_case(W=320, H=256, F=30, event=3)
    mu, sg = oracle.welford(tr)
    a = oracle.Analyzer(fr, mu, sg, len(tr))
    staged, st, bub = a.any_cam_analysis()
    assert staged == 0 and st["trig"] == spec.t0
    assert len(bub) == len(spec.bubbles)
    (cx, cy, _), b = spec.bubbles[0], bub[0]
    assert len(b["desc"]) == 11
    for k, d in enumerate(b["desc"]):
        assert abs(d["cx"] - cx) <= 0.5 and abs(d["cy"] - cy) <= 0.5
        assert abs(d["radius"] - (2 + 1.5 * k)) <= 1.5
    # bubble::dZdT / dRdT (bubble.cpp:101-118)
    x0, xn = b["desc"][0]["x"], b["desc"][-1]["x"]
    assert b["dzdt"] == pytest.approx((x0 - xn) / 10.0)
    a.close()


def test_masks(oracle):
    spec, fr, tr = make_case(W=320, H=256, F=30, event=3)
    mu, sg = oracle.welford(tr)
    cx, cy, _ = spec.bubbles[0]
    fid = np.full((300, 400), 255, np.uint8)
    fid[cy - 20:cy + 20, cx - 20:cx + 20] = 0  # bubble genesis outside the fiducial area
    a = oracle.Analyzer(fr, mu, sg, len(tr), fid_mask=fid)
    staged, st, bub = a.any_cam_analysis()
    assert bub == [] and staged == -3  # retried until the end of the event, then "no trigger"
    a.close()
    # all genesis contours inside the bellows mask (and no template): contours re-found, kept
    bel = np.full((300, 400), 255, np.uint8)
    a = oracle.Analyzer(fr, mu, sg, len(tr), bel_mask=bel)
    staged, st, bub = a.any_cam_analysis()
    assert staged == 0 and len(bub) == 1
    a.close()
