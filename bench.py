#!/usr/bin/env python3
"""bench.py -- frames/s of the bubble-detection hot path on synthetic 1280x1024 8-bit stacks.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`, one process per
GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks reduction: the data path
has no collective, events shard embarrassingly), rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): one synthetic 40l-19-like run,
E events x cams {0,1} x F=41 frames of 1280x1024 u8, resident in HBM before the timed region.
A "step" is one pass of the detect path over the whole run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--events", type=int, default=100)
    ap.add_argument("--cams", type=int, default=2)
    ap.add_argument("--frames", type=int, default=41)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--train-events", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch

    from autobub3hs_amd import hip, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device(dev))

    W, H, F, E, C = args.width, args.height, args.frames, args.events, args.cams
    P = W * H
    S = E * C  # stacks on this rank

    # ---- synthetic run, generated straight into HBM (rank r owns events r*E .. r*E+E-1) ----------
    t0 = time.time()
    slab = torch.empty((S, F, H, W), dtype=torch.uint8, device=dev)
    specs = []
    bgs = [synth.background(W, H, synth.BASE_SEED + c, "torch", dev) for c in range(C)]
    for e in range(E):
        ev = rank * E + e
        for c in range(C):
            spec = synth.random_spec(W, H, F, ev, c, p_second=0.2)
            specs.append(spec)
            synth.render_event(W, H, spec, ev, c, xp="torch", device=dev, out=slab[e * C + c], bg=bgs[c])
    # per-camera model from frames 0,1 of the first train-events events (K1 on the GPU)
    mus, sgs = [], []
    for c in range(C):
        idx = torch.tensor([((e * C + c) * F + f) for e in range(min(args.train_events, E)) for f in (0, 1)],
                           dtype=torch.int32, device=dev)
        mu, sg = hip.train(slab, W, H, idx=idx)
        mus.append(mu)
        sgs.append(sg)
    mu_d = torch.stack(mus).contiguous()
    sg_d = torch.stack(sgs).contiguous()
    s6_d = hip.sigma6(sg_d)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    # ---- the step ------------------------------------------------------------------------------
    jobs = hip.stack_jobs(S, F, 1, F - 1, 2, C, dev)
    njobs = jobs.shape[0]
    hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
    hist_h = torch.empty((njobs, 256), dtype=torch.int32).pin_memory()
    k2_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(args.steps)]

    def step(k=None):
        if k is not None:
            k2_ev[k][0].record()
        hip.diff_hist(slab, s6_d, jobs, W, H, store=False, hist=hist)
        if k is not None:
            k2_ev[k][1].record()
        hist_h.copy_(hist, non_blocking=True)
        torch.cuda.current_stream().synchronize()

    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t1
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    frames_per_step = S * F * world
    value = frames_per_step * args.steps / dt
    k2_ms = sum(a.elapsed_time(b) for a, b in k2_ev) / max(1, args.steps)
    alg_bytes = 3.0 * P * njobs  # trigger-only mode: read cur, ref, sigma; D not materialised
    achieved = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0

    out = {
        "metric": "frames/s end-to-end detect @1280x1024 8-bit",
        "value": value,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"synthetic 40l-19-like run: {E} events x {C} cams x {F} frames {W}x{H} u8 per GPU, HBM-resident",
            "stage": "trigger search (K2 fused ProcessFrame+hist, all stacks in one launch) + histogram D2H",
            "events_per_gpu": E, "cams": C, "frames_per_stack": F, "width": W, "height": H,
            "parallelism": f"events sharded over {world} GPU(s), no collective",
            "gen_seconds": round(gen_s, 1),
        },
        "roofline": {
            "kernel": "k2_rows<5,false> (fused ProcessFrame + 256-bin histogram, trigger-only mode: 3*W*H B/job)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
            "ms_per_launch": k2_ms, "jobs_per_launch": njobs,
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc

        orc.build()
        sg0 = sg_d[0].cpu().numpy()
        done, tcpu = 0, 0.0
        s = 0
        while tcpu < args.cpu_seconds and s < S:
            st = slab[s].cpu().numpy()
            c = s % C
            tc = time.perf_counter()
            _, hh = orc.bench_trigger_pass(st, sg_d[c].cpu().numpy(), 2, 1, F - 1, want_hists=True)
            tcpu += time.perf_counter() - tc
            # the CPU sample doubles as a full-size parity check of the GPU histograms
            if not np.array_equal(hh, hist_h[s * (F - 1):(s + 1) * (F - 1)].numpy().astype(np.uint32)):
                raise SystemExit("bench.py: GPU histograms differ from the CPU oracle on the sampled stack")
            done += F
            s += max(1, S // 8)
        out["cpu_baseline"] = {
            "value": done / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} frames ({done // F} stacks of {F}) of the same workload, ProcessFrame+calcHist "
                      f"restatement (oracle/abub_oracle.c, gcc -O2), {tcpu:.1f} s on 1 core of {os.cpu_count()}",
        }
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
