#!/usr/bin/env python3
"""Randomised cross-check of the trigger-only K2 pass (bound scan / chained scan + exact groups + hand-over) against
the store-mode row machine on the GPU: random widths (all NDW), heights, chunking regimes (few / many jobs), noise
densities from empty to dense, blobs, dense bands, sigma patterns, chain hints right and wrong.
Usage (GPU box): python tools/fuzz_trigger_pass.py [seconds] [seed]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from autobub3hs_amd import hip  # noqa: E402

DEV = "cuda:0"
WIDTHS = [4, 8, 64, 100, 256, 320, 512, 768, 1024, 1280, 1536, 1680, 1792, 2048]


def case(rs):
    W = int(rs.choice(WIDTHS))
    H = int(rs.choice([1, 2, 3, 5, 16, 17, 40, 64, 130, 257]))
    nst = int(rs.choice([1, 2, 5]))
    F = int(rs.choice([2, 3, 6, 9]))
    off = int(rs.choice([1, 2, 3]))
    dens = float(rs.choice([0, 1e-4, 1e-3, 1e-2, 0.2]))
    base = torch.randint(20, 200, (H, W), device=DEV, dtype=torch.int16)
    fr = base[None].repeat(nst * F, 1, 1)
    if dens > 0:
        mask = torch.rand((nst * F, H, W), device=DEV) < dens
        fr = fr + mask.to(torch.int16) * torch.randint(-12, 13, (nst * F, H, W), device=DEV, dtype=torch.int16)
    for _ in range(int(rs.randint(0, 4))):  # blobs
        f = int(rs.randint(0, nst * F)); cy, cx, r = int(rs.randint(0, H)), int(rs.randint(0, W)), int(rs.randint(1, 40))
        yy = torch.arange(H, device=DEV)[:, None]; xx = torch.arange(W, device=DEV)[None, :]
        fr[f] = torch.where((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r, fr[f] + int(rs.choice([-40, 40])), fr[f])
    if rs.rand() < 0.3:  # dense band / dense frame
        f = int(rs.randint(0, nst * F)); a = int(rs.randint(0, H)); b = int(rs.randint(a, H)) + 1
        fr[f, a:b] += int(rs.randint(8, 30))
    fr = fr.clamp(0, 255).to(torch.uint8)
    sg = torch.randint(0, 3, (2, H, W), device=DEV, dtype=torch.uint8)
    s6 = hip.sigma6(sg)
    jobs = hip.stack_jobs(nst, F, 1, F - 1, off, 2, DEV)
    hip.k2_set_option("bound", 0)  # the reference of this tool: the plain row machine, storing every pixel
    ref, D = hip.diff_hist(fr, s6, jobs, W, H, store=True)
    hip.k2_set_option("bound", 1)
    sh, sD = hip.diff_hist(fr, s6, jobs, W, H, store=True)  # store mode of the bound-and-verify pass (fill + non-zero pixels)
    assert torch.equal(sD, D), ("store image", W, H, nst, F, off, dens)
    outs = {"store": sh, "plain": hip.diff_hist(fr, s6, jobs, W, H)[0]}
    if F > 1:
        outs["chain"] = hip.diff_hist(fr, s6, jobs, W, H, chain=(F - 1, off))[0]
        outs["chain-wrong-stride"] = hip.diff_hist(fr, s6, jobs, W, H, chain=(F - 1, off % 3 + 1))[0]
        outs["chain-one-block"] = hip.diff_hist(fr, s6, jobs, W, H, chain=(nst * (F - 1), off))[0]
        # deferred form: the scan alone, then the row machine on a random subset of the jobs it flagged
        dh, st = hip.diff_hist_deferred(fr, s6, jobs, W, H, chain=(F - 1, off))
        inc = st[2].clone()
        want = (torch.rand(inc.shape, device=DEV) < 0.5).to(torch.uint8)
        hip.diff_hist_pieces(fr, s6, jobs, W, H, dh, st, want)
        final = (inc == 0) | (want != 0)
        assert torch.equal(dh[final], ref[final]), ("deferred", W, H, nst, F, off, dens)
    torch.cuda.synchronize()
    chk = torch.stack([torch.bincount(D[j].flatten().to(torch.int64), minlength=256) for j in range(min(3, D.shape[0]))])
    assert torch.equal(chk.to(ref.dtype), ref[:chk.shape[0]]), ("store hist != bincount(D)", W, H)
    for k, v in outs.items():
        assert torch.equal(v, ref), (k, W, H, nst, F, off, dens)
    return W * H * nst * (F - 1)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rs = np.random.RandomState(seed)
    torch.manual_seed(seed)
    t0, n, px = time.time(), 0, 0
    while time.time() - t0 < secs:
        px += case(rs)
        n += 1
        if n % 50 == 0:
            print(f"{n} cases, {px / 1e9:.2f} Gpx, {time.time() - t0:.0f} s", flush=True)
    print(f"OK: {n} cases, {px / 1e9:.2f} Gpx")


if __name__ == "__main__":
    main()
