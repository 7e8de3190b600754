#!/usr/bin/env python3
"""BASELINE.json configs[0] / SURVEY 8d "Config 1": one synthetic 30l-16-like event, camera 0, F = 50 frames of
1280x1024 u8, through the CPU oracle.  The frames are NOT committed (65 MB): they are regenerated from the seed by
autobub3hs_amd/synth.py (integer-only, numpy == torch); what is committed is what the oracle finds in them
(tests/golden/config1_expected.json).  Run from the repo root:  python tests/golden/make_config1.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from autobub3hs_amd import synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

W, H, F, EVENT, CAM, NTRAIN = 1280, 1024, 50, 3016, 0, 10


def scene():
    spec = synth.EventSpec(F, t0=31, bubbles=[(412, 633, -40), (901, 250, 40)], flicker=17)
    return spec


def main():
    orc.build()
    fr = synth.render_event(W, H, scene(), EVENT, CAM)
    tr = synth.training_pairs(W, H, NTRAIN, CAM, F)
    mu, sg = orc.welford(tr)
    a = orc.Analyzer(fr, mu, sg, len(tr))
    staged, state, bubbles = a.any_cam_analysis()
    a.close()
    _, hists = orc.bench_trigger_pass(fr, sg, 2, 1, F - 1, want_hists=True)
    out = {
        "W": W, "H": H, "F": F, "event": EVENT, "cam": CAM, "ntrain_events": NTRAIN,
        "staged": staged, "state": dict(state),
        "bubbles": [{"desc": [{k: (None if isinstance(v, float) and v != v else v) for k, v in d.items()} for d in b["desc"]]}
                    for b in bubbles],
        "mu_sum": int(mu.astype(np.int64).sum()), "sigma_hist": np.bincount(sg.ravel(), minlength=8)[:8].tolist(),
        "trigger_hist_nonzero": hists[:, 1:].sum(1).astype(int).tolist(),
        "trigger_hist_checksum": int((hists.astype(np.uint64) * (1 + np.arange(256, dtype=np.uint64))[None, :]).sum() % (1 << 61)),
    }
    json.dump(out, open(os.path.join(HERE, "config1_expected.json"), "w"), indent=1)
    print("staged", staged, "state", state, "bubbles", len(bubbles))


if __name__ == "__main__":
    main()
