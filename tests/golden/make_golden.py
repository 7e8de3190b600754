#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle (oracle/abub_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
Inputs: seeded synthetic scenes (autobub3hs_amd/synth.py) and small crops of the four real sample frames the
reference ships as DATA under cam_masks/40l-19/ (read only at generation time; the crops are committed here, the
GPU box never needs /root/reference).  Outputs are what the oracle computes; tests/test_golden.py checks both the
oracle (CPU) and the HIP path (GPU) against them bit for bit.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from autobub3hs_amd import synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

REF_FRAMES = "/root/reference/cam_masks/40l-19"


def real_crops():
    from PIL import Image

    out = {}
    for name in sorted(os.listdir(REF_FRAMES)):
        if name.endswith(".png") and "image3" in name:
            im = np.array(Image.open(os.path.join(REF_FRAMES, name)).convert("L"))
            # a textured region and a flat one, 96 x 160 each
            out[name.replace(".png", "") + "_a"] = im[400:496, 700:860].copy()
            out[name.replace(".png", "") + "_b"] = im[100:196, 200:360].copy()
    return out


def main():
    rs = np.random.RandomState(20201)
    crops = real_crops()
    names = sorted(crops)
    # ---- per-pixel primitives on real-image crops -------------------------------------------------
    prim = {}
    for k, n in enumerate(names):
        base = crops[n].astype(int)
        f0 = np.clip(base + rs.randint(-2, 3, base.shape), 0, 255).astype(np.uint8)
        f1 = np.clip(base + rs.randint(-2, 3, base.shape), 0, 255).astype(np.uint8)
        f2 = np.clip(base + rs.randint(-2, 3, base.shape), 0, 255).astype(np.uint8)
        yy, xx = np.mgrid[:base.shape[0], :base.shape[1]]
        blob = ((yy - 40 - k) ** 2 + (xx - 70 - 3 * k) ** 2) <= (4 + k) ** 2
        f2 = np.where(blob, np.clip(f2.astype(int) - 45, 0, 255), f2).astype(np.uint8)
        train = np.stack([np.clip(base + rs.randint(-2, 3, base.shape), 0, 255) for _ in range(8)]).astype(np.uint8)
        mu, sg = orc.welford(train)
        D = orc.process_frame(f2, f0, sg)
        O = orc.posttrig_frame(f2, mu, sg)
        mask, T = orc.binarize(O, 3)
        prim[f"{k}_frames"] = np.stack([f0, f1, f2])
        prim[f"{k}_train"] = train
        prim[f"{k}_mu"], prim[f"{k}_sigma"] = mu, sg
        prim[f"{k}_D"], prim[f"{k}_Dhist"] = D, orc.hist256(D)
        prim[f"{k}_O"], prim[f"{k}_Ohist"] = O, orc.hist256(O)
        prim[f"{k}_mask"], prim[f"{k}_T"] = mask, np.int32(T)
        cs = orc.find_contours(mask)
        prim[f"{k}_ncontours"] = np.int32(len(cs))
        for c, (pts, _) in enumerate(cs):
            prim[f"{k}_contour{c}"] = pts
    prim["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "primitives.npz"), **prim)

    # ---- whole events (synthetic), expected AnyCamAnalysis outcome -------------------------------
    W, H, F = 240, 96, 24
    events, expected = {}, {}
    tr = synth.training_pairs(W, H, 8, 0, F)
    mu, sg = orc.welford(tr)
    events["train"], events["mu"], events["sigma"] = tr, mu, sg
    specs = {
        "one": synth.EventSpec(F, t0=12, bubbles=[(100, 60, 40)]),
        "two": synth.EventSpec(F, t0=11, bubbles=[(60, 30, -40), (180, 70, 40)]),
        "none": synth.EventSpec(F),
        "flicker": synth.EventSpec(F, t0=14, bubbles=[(150, 50, 40)], flicker=6, flicker_adu=12),
        "late": synth.EventSpec(F, t0=20, bubbles=[(120, 48, 40)]),
    }
    for k, (name, spec) in enumerate(specs.items()):
        fr = synth.render_event(W, H, spec, 40 + k, 0)
        events[name] = fr
        a = orc.Analyzer(fr, mu, sg, len(tr))
        staged, state, bubbles = a.any_cam_analysis()
        expected[name] = {
            "staged": staged, "state": state,
            "bubbles": [{"desc": [{kk: (None if isinstance(v, float) and v != v else v) for kk, v in d.items()}
                                  for d in b["desc"]]} for b in bubbles],
            "sig": [None if s != s else (s if np.isfinite(s) else str(s)) for s in a.sig_trace().tolist()],
        }
        a.close()
    np.savez_compressed(os.path.join(HERE, "events.npz"), **events)
    with open(os.path.join(HERE, "events_expected.json"), "w") as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    print("wrote", [n for n in os.listdir(HERE) if n.endswith((".npz", ".json"))])


if __name__ == "__main__":
    main()
