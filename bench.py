#!/usr/bin/env python3
"""bench.py -- frames/s of the bubble-detection hot path on synthetic 1280x1024 8-bit stacks.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`, one process per
GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks reduction: the data path
has no collective, events shard embarrassingly), rank 0 prints ONE JSON line.

Workload at N=1 (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): one synthetic 40l-19-like run,
E events x cams {0,1} x F=41 frames of 1280x1024 u8, resident in HBM before the timed region.
A "step" is one pass of the detect path over the whole run.  Steps are software-pipelined (--inflight, default 3:
host.PipelineRing): the host stages of step k run while the GPU works on step k+1; all K steps start and finish
inside the timed region.
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
    ap.add_argument("--threads", type=int, default=16, help="host threads of the per-event state machines")
    ap.add_argument("--inflight", type=int, default=3,
                    help="steps in flight at once (each on its own pipeline object and host thread)")
    ap.add_argument("--stream-steps", type=int, default=2,
                    help="extra steps with the run uploaded from pinned host memory (PCIe-inclusive rate; 0 = skip)")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch

    from autobub3hs_amd import hip, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local = local % ndev  # (rehearsals of N ranks on a smaller box share devices; see ABUB_BENCH_BACKEND)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist = None
    backend = os.environ.get("ABUB_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for rehearsals
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend)

    W, H, F, E, C = args.width, args.height, args.frames, args.events, args.cams
    P = W * H
    S = E * C  # stacks on this rank

    # ---- synthetic run, generated straight into HBM (rank r owns events r, r+N, r+2N, ...) ----------
    t0 = time.time()
    slab = torch.empty((S, F, H, W), dtype=torch.uint8, device=dev)
    specs = []
    bgs = [synth.background(W, H, synth.BASE_SEED + c, "torch", dev) for c in range(C)]
    ev_ids = shard.global_event_ids(E, rank, world)  # round-robin over ranks, like schedule(static,1)
    for e in range(E):
        ev = ev_ids[e]
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

    # ---- the step: end-to-end detect of the whole run (host/pipeline.cpp) -------------------------
    from autobub3hs_amd import host

    tss = [2 * min(args.train_events, E)] * C
    # --inflight N (host.PipelineRing): N pipeline objects, each driven by its own host thread; step k runs on pipeline
    # k % N, so the host stages of one step (state machines, contours) overlap the GPU stages of the next.  Every step
    # is still a full pass over the same batch, and all K of them complete inside the timed region.
    ninfl = max(1, args.inflight)
    ring = host.PipelineRing(ninfl, local, W, H, F, E, C, tss, nthreads=max(1, args.threads))
    pipes = ring.pipes
    pipe = pipes[0]
    stream = torch.cuda.current_stream().cuda_stream
    njobs = S * (F - 1)

    def run_steps(n):
        return ring.run_batches([slab] * n, mu_d, s6_d, stream)

    nwarm = max(args.warmup, ninfl if args.warmup else 0)  # every pipeline object of the ring gets one untimed step
    run_steps(nwarm)
    fingerprint = pipe.summary()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stage = {"stage1_ms": 0.0, "stage2_ms": 0.0, "stage3_ms": 0.0, "stage4_ms": 0.0, "total_ms": 0.0, "s3_gpu_ms": 0.0,
             "s3_list_ms": 0.0, "s3_bucket_ms": 0.0, "pairs": 0.0, "rounds": 0}
    for tm in run_steps(args.steps):
        for kk in stage:
            stage[kk] += tm[kk]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t1, dev if backend == "nccl" else None)
    for p_ in pipes[:max(1, min(ninfl, args.steps))]:
        assert p_.summary() == fingerprint, "results changed between steps"
    n_trig = sum(1 for r in fingerprint if r[0] == 0)
    n_bub = sum(r[2] for r in fingerprint)

    # ---- PCIe-inclusive rate (reported in config only, never `value`): the same run streamed from pinned host memory
    pcie = None
    if args.stream_steps > 0 and rank == 0 and world == 1:
        h_slab = torch.empty(slab.shape, dtype=torch.uint8).pin_memory()
        h_slab.copy_(slab)
        torch.cuda.synchronize()
        pipe.run_host(h_slab, mu_d, s6_d)  # warm-up (allocates the pipeline-owned slab)
        assert pipe.summary() == fingerprint, "streamed run differs from the resident run"
        ts = time.perf_counter()
        for _ in range(args.stream_steps):
            pipe.run_host(h_slab, mu_d, s6_d)
        tstream = (time.perf_counter() - ts) / args.stream_steps
        pcie = {"frames_per_s": S * F / tstream, "ms_per_run": tstream * 1e3,
                "GBps_host_to_hbm": S * F * P / tstream / 1e9}
        del h_slab

    # ---- dominant kernel alone, HIP events on the launch stream (roofline object) ---------------
    jobs = hip.stack_jobs(S, F, 1, F - 1, 2, C, dev)
    hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
    kreps = max(3, min(10, args.steps))
    chain = (F - 1, 2)  # the pipeline's own call: stack-structured job list, FindTriggerFrame's two-frame offset
    hip.diff_hist(slab, s6_d, jobs, W, H, store=False, hist=hist, chain=chain)
    k2_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(kreps)]
    for a_, b_ in k2_ev:
        a_.record()
        hip.diff_hist(slab, s6_d, jobs, W, H, store=False, hist=hist, chain=chain)
        b_.record()
    torch.cuda.synchronize()
    k2_ms = sum(a_.elapsed_time(b_) for a_, b_ in k2_ev) / kreps
    hist_h = hist.cpu()

    frames_per_step = S * F * world
    value = frames_per_step * args.steps / dt
    alg_bytes = 3.0 * P * njobs  # trigger-only mode: read cur, ref, sigma; D not materialised
    achieved = alg_bytes / (k2_ms * 1e-3) / 1e9 if k2_ms > 0 else 0.0
    # HBM traffic of this kernel from rocprofv3 PMC passes (tools/prof_k2.sh: FETCH_SIZE x2 + WRITE_SIZE, separate
    # --pmc runs on the native microbench of the same kernel and frame size), scaled per job; null if not measured
    traffic, traffic_src = None, None
    for cand in sorted((os.path.join(ROOT, "profiles", d, "k2_hist_pmc_summary.json")
                        for d in os.listdir(os.path.join(ROOT, "profiles"))
                        if os.path.isdir(os.path.join(ROOT, "profiles", d))), reverse=True):
        if os.path.exists(cand):
            pm = json.load(open(cand))
            if pm["micro"]["W"] == W and pm["micro"]["H"] == H:
                traffic = pm["hbm_bytes_per_job"] * njobs
                traffic_src = os.path.relpath(cand, ROOT)
                break

    out = {
        "metric": f"frames/s end-to-end detect @{W}x{H} 8-bit",  # BASELINE.json's metric at the default 1280x1024
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
            "workload": f"synthetic 40l-19-like run: {E} events x {C} cams x {F} frames {W}x{H} u8 per GPU, HBM-resident; "
                        "per step: trigger search over every frame, genesis localisation, <=10-frame tracking, per-bubble records",
            "events_per_gpu": E, "cams": C, "frames_per_stack": F, "width": W, "height": H,
            "parallelism": f"events sharded over {world} GPU(s), no collective",
            "host_threads": args.threads,
            "warmup_steps_run": nwarm,
            "steps_in_flight": ninfl,  # stage_ms below are wall times inside one step: with >1 in flight they include
                                       # queueing behind the other steps' kernels and no longer add up to ms_per_step
            "triggered_stacks": n_trig, "bubbles": n_bub,
            "stage_ms": {k: round(v / args.steps, 3) for k, v in stage.items()},
            "gen_seconds": round(gen_s, 1),
            "pcie_inclusive": pcie,
        },
        "roofline": {
            "kernel": "K2 trigger-only pass = k2_bound_chain<5,2> (dominant) + k2_exact_groups + k2_rows<5> on handed-over "
                      "rows + k_hist_bin0, timed together: fused ProcessFrame + 256-bin histogram, 3*W*H B/job",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
            "ms_per_launch": k2_ms, "jobs_per_launch": njobs,
            "note": "frac can exceed 1: the contract's algorithmic accounting charges cur, ref and sigma6 (3*W*H per job) to "
                    "HBM, but a frame's second use and sigma6 are served on-chip (see traffic: measured HBM bytes per launch)",
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline = the oracle's whole detect path (AnyCamAnalysis restatement) on a bounded sample
        # of the same stacks, 1 core.  It doubles as a full-size parity check of the GPU results.
        from oracle import pyoracle as orc

        orc.build()
        done, tcpu, nstk = 0, 0.0, 0
        s_i = 0
        mu_h = mu_d.cpu().numpy()
        sg_h = sg_d.cpu().numpy()
        while tcpu < args.cpu_seconds and s_i < S:
            st = slab[s_i].cpu().numpy()
            c = s_i % C
            tc = time.perf_counter()
            a = orc.Analyzer(st, mu_h[c], sg_h[c], tss[c])
            staged_r, state_r, bub_r = a.any_cam_analysis()
            tcpu += time.perf_counter() - tc
            a.close()
            staged, state, bub, err = pipe.result(s_i)
            same = (staged, state) == (staged_r, state_r) and len(bub) == len(bub_r) and all(
                [tuple(d[k] for k in "xywh") for d in b["desc"]] == [tuple(d[k] for k in "xywh") for d in r["desc"]]
                for b, r in zip(bub, bub_r))
            if not same:
                raise SystemExit(f"bench.py: GPU result of stack {s_i} differs from the CPU oracle: "
                                 f"{(staged, state)} vs {(staged_r, state_r)} {err}")
            # and the trigger-pass histograms of that stack, bit for bit
            _, hh = orc.bench_trigger_pass(st, sg_h[c], 2, 1, F - 1, want_hists=True)
            if not np.array_equal(hh, hist_h[s_i * (F - 1):(s_i + 1) * (F - 1)].numpy().astype(np.uint32)):
                raise SystemExit("bench.py: GPU histograms differ from the CPU oracle on the sampled stack")
            done += F
            nstk += 1
            s_i += max(1, S // 24)
        # the same oracle, event-parallel over the host cores (the reference's own OpenMP loop is over events,
        # AutoBubStart3.cpp:342): reported beside the 1-core figure, not instead of it
        from concurrent.futures import ThreadPoolExecutor

        ncore = max(1, min(len(os.sched_getaffinity(0)), 32, S))
        stacks_mc = [(slab[s].cpu().numpy(), s % C) for s in range(0, S, max(1, S // ncore))][:ncore]

        def _one(item):
            st, c = item
            a = orc.Analyzer(st, mu_h[c], sg_h[c], tss[c])  # ctypes drops the GIL inside the oracle
            a.any_cam_analysis()
            a.close()

        with ThreadPoolExecutor(len(stacks_mc)) as ex:
            tc = time.perf_counter()
            list(ex.map(_one, stacks_mc))
            tmc = time.perf_counter() - tc
        out["cpu_baseline"] = {
            "value": done / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
            "all_cores": {"value": len(stacks_mc) * F / tmc, "cores": len(stacks_mc),
                          "sample": f"{len(stacks_mc)} stacks at once, one per thread, {tmc:.1f} s"},
            "sample": f"{nstk} stacks ({done} frames) of the same workload through the oracle's end-to-end detect "
                      f"(oracle/abub_oracle.c, gcc -O2, results identical to the GPU's), {tcpu:.1f} s on 1 core of {os.cpu_count()}",
        }
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
