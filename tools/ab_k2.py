#!/usr/bin/env python3
"""A/B of K2 launcher options on the bench's trigger pass (200 stacks x 40 jobs), interleaved on one box.
usage: python3 tools/ab_k2.py [--width 1280 --height 1024] [--reps 6] cfg [cfg ...]
  cfg = comma list of option=value (chain, wg, sync, split, list, budget), e.g.  chain=3,wg=4,sync=2
Prints one JSON line per configuration: mean / min ms per launch of the pass, us per job, compulsory TB/s."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from autobub3hs_amd import hip, synth

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1280)
ap.add_argument("--height", type=int, default=1024)
ap.add_argument("--events", type=int, default=100)
ap.add_argument("--frames", type=int, default=41)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--store", type=int, default=0)
ap.add_argument("--nocheck", action="store_true", help="do not compare the histograms of the configurations (timing experiments)")
ap.add_argument("--lib", action="append", default=[], help="name=path of a libabub_hip.so variant; select it in a cfg with lib=name (not supported: one library per process)")
ap.add_argument("cfgs", nargs="+")
a = ap.parse_args()
W, H, F, E, C = a.width, a.height, a.frames, a.events, 2
dev = "cuda:0"
S = E * C
slab = torch.empty((S, F, H, W), dtype=torch.uint8, device=dev)
bgs = [synth.background(W, H, synth.BASE_SEED + c, "torch", dev) for c in range(C)]
for e in range(E):
    for c in range(C):
        spec = synth.random_spec(W, H, F, e, c, p_second=0.2)
        synth.render_event(W, H, spec, e, c, xp="torch", device=dev, out=slab[e * C + c], bg=bgs[c])
sgs = []
for c in range(C):
    idx = torch.tensor([((e * C + c) * F + f) for e in range(20) for f in (0, 1)], dtype=torch.int32, device=dev)
    mu, sg = hip.train(slab, W, H, idx=idx)
    sgs.append(sg)
s6 = hip.sigma6(torch.stack(sgs).contiguous())
njobs = S * (F - 1)
jobs = hip.stack_jobs(S, F, 1, F - 1, 2, C, dev)
hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
D = torch.empty((njobs, H, W), dtype=torch.uint8, device=dev) if a.store else None
DEFAULTS = {"chain": -1, "wg": -1, "sync": -1, "split": 1, "list": 0, "budget": 512, "bound": 1, "pf": 1, "scanpf": -1}
cfgs = []
for c in a.cfgs:
    d = dict(DEFAULTS)
    for kv in c.split(","):
        if kv and kv != "default":
            k, v = kv.split("=")
            d[k] = int(v)
    cfgs.append((c, d))


def apply(d):
    for k, v in d.items():
        hip.k2_set_option(k, v)


def launch():
    if a.store:
        hip.diff_hist(slab, s6, jobs, W, H, store=True, hist=hist, diff=D, chain=(F - 1, 2))
    else:
        hip.diff_hist(slab, s6, jobs, W, H, store=False, hist=hist, chain=(F - 1, 2))


ref = None
times = {c: [] for c, _ in cfgs}
for rep in range(a.reps + 1):
    for name, d in cfgs:
        apply(d)
        launch()
        torch.cuda.synchronize()
        if rep == 0:  # warm-up + equality of the results
            h = hist.clone()
            if ref is None:
                ref = h
            assert a.nocheck or torch.equal(h, ref), f"{name}: histograms differ from {cfgs[0][0]}"
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch()
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
P = W * H
for name, _ in cfgs:
    t = times[name]
    ms = sum(t) / len(t)
    nP = 2 if a.store else 1
    print(json.dumps({"cfg": name, "W": W, "H": H, "store": a.store, "jobs": njobs, "ms": round(ms, 4), "ms_min": round(min(t), 4),
                      "us_per_job": round(1e3 * ms / njobs, 4), "compulsory_TBps": round(nP * P * njobs / ms / 1e9, 3),
                      "frac_of_8TBps": round(nP * P * njobs / ms / 1e9 / 8, 4)}), flush=True)
