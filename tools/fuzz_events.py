#!/usr/bin/env python3
"""Randomised end-to-end parity: seeded synthetic events (bubbles, second bubbles, flicker frames, empty events, a slow
global drift that triggers without a bubble) through the batched run pipeline on the GPU versus the CPU oracle, stack by
stack: staged status, trigger frame, status code, loc_thres, every box of every tracked bubble, centroids to 1e-4.
Usage (GPU box): python tools/fuzz_events.py [seconds] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from autobub3hs_amd import hip, host, synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

DEV = "cuda:0"


def batch(rs, k):
    W, H = [(1280, 128), (1680, 96), (322, 120), (1280, 200), (640, 256)][k % 5]
    F = int(rs.choice([41, 41, 30, 12, 7]))
    E, C = 6, 2
    ntr = [int(rs.choice([8, 8, 2])), 8]  # a small training set switches camera 0 to the one-frame offset
    slab = np.zeros((E, C, F, H, W), np.uint8)
    for e in range(E):
        for c in range(C):
            ev = int(rs.randint(0, 1 << 20))
            spec = synth.random_spec(W, H, F, ev, c, p_second=0.4, p_none=0.2, p_flicker=0.4, margin=min(25, H // 4))
            slab[e, c] = synth.render_event(W, H, spec, ev, c)
            if rs.rand() < 0.1:  # persistent +1 step: trigger candidates without an accepted bubble (retry rounds)
                t = int(rs.randint(3, max(4, F - 2)))
                slab[e, c, t:] = np.clip(slab[e, c, t:].astype(int) + 1, 0, 255)
    models = [orc.welford(synth.training_pairs(W, H, ntr[c], c, F)) for c in range(C)]
    tss = [2 * ntr[c] for c in range(C)]
    d_slab = torch.from_numpy(slab).to(DEV)
    d_mu = torch.from_numpy(np.stack([m[0] for m in models])).to(DEV)
    d_s6 = hip.sigma6(torch.from_numpy(np.stack([m[1] for m in models])).to(DEV))
    pipe = host.Pipeline(0, W, H, F, E, C, tss, nthreads=8)
    pipe.run(d_slab, d_mu, d_s6, torch.cuda.current_stream().cuda_stream)
    nb = 0
    for e in range(E):
        for c in range(C):
            staged, state, bubbles, err = pipe.result(e * C + c)
            a = orc.Analyzer(slab[e, c], models[c][0], models[c][1], tss[c])
            rs_, rstate, rb = a.any_cam_analysis()
            a.close()
            assert (staged, dict(state)) == (rs_, dict(rstate)), (W, H, F, e, c, staged, state, rs_, rstate, err)
            assert [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in bubbles] == \
                   [[tuple(d[k] for k in "xywh") for d in b["desc"]] for b in rb], (W, H, F, e, c)
            for b, r in zip(bubbles, rb):
                for d, q in zip(b["desc"], r["desc"]):
                    if q["cx"] == q["cx"]:
                        assert abs(d["cx"] - q["cx"]) <= 1e-4 and abs(d["cy"] - q["cy"]) <= 1e-4
            nb += len(rb)
    pipe.close()
    return E * C, nb


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    host.build()
    orc.build()
    t0, ns, nb, k = time.time(), 0, 0, 0
    while time.time() - t0 < secs:
        a, b = batch(rs, k)
        ns += a
        nb += b
        k += 1
        if k % 10 == 0:
            print(f"{ns} stacks, {nb} bubbles, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {ns} stacks, {nb} bubbles identical to the oracle")


if __name__ == "__main__":
    main()
