#!/usr/bin/env python3
"""bench.py -- frames/s of the bubble-detection hot path on synthetic 1280x1024 8-bit stacks.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`, one process per
GPU (torch.distributed / RCCL only for the barrier and the max-over-ranks reduction: the data path
has no collective, events shard embarrassingly), rank 0 prints ONE JSON line.  With --gpus N > 1 and no
launcher environment (WORLD_SIZE unset) this process only spawns `python -m torch.distributed.run` with N
ranks of itself, relays rank 0's line and exits with the worst return code; it never touches HIP itself.

Workload at N=1 (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): one synthetic 40l-19-like run,
E events x cams {0,1} x F=41 frames of 1280x1024 u8, resident in HBM before the timed region.
A "step" is one pass of the detect path over the whole run.  Steps are software-pipelined (--inflight, default 6:
host.PipelineRing): the host stages of step k run while the GPU works on step k+1; all K steps of a block start
and finish inside that block's timed region.  The K-step block (barrier + synchronize on both sides, max over
ranks) is repeated until --min-seconds have been measured; `ms_per_step` / `value` are the MEDIAN block.

Also reported (config.microbench): BASELINE configs[2], the fused kernel alone on a 10k-frame slab, store mode and
trigger-only mode, bound-and-verify and the plain row machine (the dense-regime worst case).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse(argv=None):
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
    ap.add_argument("--decode-threads", type=int, default=0,
                    help="decode threads of the ingestion leg (0 = the cores this process may run on, at most 128)")
    ap.add_argument("--inflight", type=int, default=6,
                    help="steps in flight at once (each on its own pipeline object and host thread); measured on one MI355X: "
                         "2 -> 2.46, 3 -> 2.59, 4 -> 2.65, 6 -> 2.76, 8 -> 2.79, 12 -> 2.58 M frames/s")
    ap.add_argument("--pipe-threads", type=int, default=0,
                    help="host threads per pipeline object in flight (0 = --threads, capped so that all ranks of the node together stay near 400)")
    ap.add_argument("--stream-steps", type=int, default=10,
                    help="runs streamed from pinned host memory, two in flight (BASELINE configs[4] on this GPU: PCIe-inclusive "
                         "rate, reported in config only; 0 = skip)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="repeat the timed K-step block until this much time has been measured (0 = one block)")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--micro-frames", type=int, default=10000,
                    help="frames of the configs[2] kernel microbench slab (0 = skip)")
    ap.add_argument("--regime-steps", type=int, default=12,
                    help="steps per extra data regime (config.regimes: post_trigger_dense, noisy); 0 = skip")
    ap.add_argument("--no-masks", action="store_true", help="run without fiducial / bellows mask files")
    ap.add_argument("--slabs", type=int, default=0, help="copies of the run in HBM (0 = one per step in flight)")
    ap.add_argument("--regime", default="default", choices=["default", "post_trigger_dense", "noisy"],
                    help="data regime of the MAIN workload (default: BASELINE's quiet-noise run; the others are reported under "
                         "config.regimes by the default run and can be made the main workload for A/B runs)")
    ap.add_argument("--ingest-events", type=int, default=96,
                    help="events of the run written as a PNG zip archive on local disk and detected from there through the "
                         "batched ingestion path (decode-inclusive rate, config.ingest_inclusive; 0 = skip)")
    ap.add_argument("--latency-steps", type=int, default=5, help="steps run one at a time for the latency figure (0 = skip)")
    ap.add_argument("--dry", action="store_true",
                    help="no GPU work: every rank only walks the launch / barrier / max-over-ranks / report protocol "
                         "(rehearsal of the multi-rank path on CPU with ABUB_BENCH_BACKEND=gloo)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N > 1 outside a launcher: start N ranks of this script, relay rank 0's JSON line, propagate the worst rc.
    Nothing here imports torch.cuda or calls HIP (a process that has initialised the GPU must not exec/fork workers)."""
    import socket

    pre = os.environ.get("LD_PRELOAD", "") + " " + " ".join(k for k in os.environ if k.startswith(("ROCPROF", "ROCPROFILER")))
    if "rocprof" in pre.lower():
        # the profiler's preloaded library has initialised the GPU in THIS process already: starting rank processes from
        # it is the launcher hop that must not follow GPU initialisation on this pool
        print("bench.py: --gpus N > 1 under rocprofv3 is refused: profile a single rank (python3 bench.py without --gpus)",
              file=sys.stderr)
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith('{"metric"'):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line:
        print(line)
    elif rc == 0:
        rc = 3
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
    sys.exit(rc)


def usable_cores():
    """Cores this process may use: the affinity mask AND the cgroup's CPU quota (a container may see 256 cores and be
    allowed 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and int(period) > 0:
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def gather_floats(x, world, rank, dist):
    if not dist:
        return [x]
    out = [None] * world
    dist.all_gather_object(out, float(x))
    return out


def main():
    args = parse()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE result line and nothing else: everything the libraries underneath print there (the host
    # library mirrors the reference's printf progress lines, e.g. "Camera 0 training ... complete.") goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        import ctypes

        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)  # C stdio buffers of the redirected period
        except OSError:
            pass
        os.write(result_fd, (json.dumps(obj) + "\n").encode())
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch

    backend = os.environ.get("ABUB_BENCH_BACKEND", "nccl")  # nccl == RCCL on ROCm; gloo only for rehearsals
    dev = None
    if not args.dry:
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
        local = local % ndev  # (rehearsals of N ranks on a smaller box share devices; see ABUB_BENCH_BACKEND)
        torch.cuda.set_device(local)
        dev = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl" and not args.dry:
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group("gloo")
    red_dev = dev if (backend == "nccl" and not args.dry) else None

    from autobub3hs_amd import shard

    W, H, F, E, C = args.width, args.height, args.frames, args.events, args.cams
    P = W * H
    S = E * C  # stacks on this rank

    def sync():
        if not args.dry:
            torch.cuda.synchronize()

    def timed_block(fn):
        """barrier + synchronize, fn(), synchronize + barrier; returns (max-over-ranks seconds, this rank's own
        seconds up to its synchronize -- before it waits for the others --, fn's result)"""
        if dist:
            dist.barrier()
        sync()
        t = time.perf_counter()
        r = fn()
        sync()
        own = time.perf_counter() - t
        if dist:
            dist.barrier()
        return shard.max_over_ranks(time.perf_counter() - t, red_dev), own, r

    if args.dry:
        # protocol rehearsal: a "step" is a short sleep whose length depends on the rank (a visible straggler)
        def run_steps(n):
            time.sleep(n * 0.002 * (1 + rank))
            return []

        run_steps(args.warmup)
        gen_s, nwarm, ninfl, pipe_threads = 0.0, args.warmup, 1, 0
        fingerprint, pipes, pipe = [], [], None
    else:
        from autobub3hs_amd import hip, host, synth

        # ---- synthetic run, generated straight into HBM (rank r owns events r, r+N, r+2N, ...) ----------
        t0 = time.time()
        import tempfile

        # fiducial + bellows masks like cam_masks/40l-19 (BASELINE configs[1] runs with -m cam_masks): 1-bit BMP files in
        # a temporary directory; bubbles are placed inside the fiducial region and outside the bellows strip
        maskdir, accept_for = "", None
        if not args.no_masks:
            maskdir = tempfile.mkdtemp(prefix="abub_masks_") + "/"
            accept_for = synth.write_masks(maskdir, W, H, C)
        bgs = [synth.background(W, H, synth.BASE_SEED + c, "torch", dev) for c in range(C)]
        ev_ids = shard.global_event_ids(E, rank, world)  # round-robin over ranks, like schedule(static,1)

        def make_run(regime):
            """-> (slab [S,F,H,W] in HBM, mu, sigma, sigma6) of one synthetic run in the given data regime"""
            sl = torch.empty((S, F, H, W), dtype=torch.uint8, device=dev)
            for e in range(E):
                ev = ev_ids[e]
                for c in range(C):
                    spec = synth.random_spec(W, H, F, ev, c, p_second=0.2, accept=accept_for(c) if accept_for else None,
                                             regime=regime)
                    synth.render_event(W, H, spec, ev, c, xp="torch", device=dev, out=sl[e * C + c], bg=bgs[c])
            # per-camera model from frames 0,1 of the first train-events events (K1 on the GPU)
            mus, sgs = [], []
            for c in range(C):
                idx = torch.tensor([((e * C + c) * F + f) for e in range(min(args.train_events, E)) for f in (0, 1)],
                                   dtype=torch.int32, device=dev)
                mu, sg = hip.train(sl, W, H, idx=idx)
                mus.append(mu)
                sgs.append(sg)
            mu_t = torch.stack(mus).contiguous()
            sg_t = torch.stack(sgs).contiguous()
            return sl, mu_t, sg_t, hip.sigma6(sg_t)

        slab, mu_d, sg_d, s6_d = make_run(args.regime)
        torch.cuda.synchronize()
        gen_s = time.time() - t0

        # ---- the step: end-to-end detect of the whole run (host/pipeline.cpp) -------------------------
        tss = [2 * min(args.train_events, E)] * C
        # --inflight N (host.PipelineRing): N pipeline objects, each driven by its own host thread; step k runs on
        # pipeline k % N, so the host stages of one step (state machines, contours) overlap the GPU stages of the
        # next.  Every step is still a full pass over the same batch, and all K of them complete inside the block.
        ninfl = max(1, args.inflight)
        # host threads per pipeline object: --threads each (the host stages sit on every step's critical path: 16 -> 2.75,
        # 8 -> 2.68, 5 -> 2.43 M frames/s at six steps in flight), but at most two threads per usable core of this rank's share
        # (per rank: the cores this process may run on, shared by the ranks of the node and the steps in flight)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        cores = usable_cores()
        pipe_threads = args.pipe_threads if args.pipe_threads > 0 else min(max(1, args.threads),
                                                                            max(4, (2 * cores // max(1, local_world)) // ninfl))
        ring = host.PipelineRing(ninfl, local, W, H, F, E, C, tss, nthreads=pipe_threads, maskdir=maskdir)
        pipes = ring.pipes
        pipe = pipes[0]
        stream = torch.cuda.current_stream().cuda_stream
        for p_ in pipes:
            p_.set_sigma(sg_d)  # (stacks that need the bellows veto are re-run one at a time and need sigma, not 6 sigma)
        # one copy of the run per pipeline in flight: steps that overlap in time never read the same addresses
        free_b, _ = torch.cuda.mem_get_info()
        nslabs = max(1, min(args.slabs if args.slabs > 0 else ninfl, 1 + int((free_b * 0.8) // slab.numel())))
        slabs = [slab] + [slab.clone() for _ in range(nslabs - 1)]

        def run_steps(n, use=None):
            sl = use if use is not None else slabs
            return ring.run_batches([sl[k % len(sl)] for k in range(n)], mu_d, s6_d, stream)

        nwarm = max(args.warmup, ninfl if args.warmup else 0)  # every pipeline object of the ring gets one untimed step
        run_steps(nwarm)
        fingerprint = pipe.summary()

    # ---- timed region: blocks of EXACTLY K steps, repeated until --min-seconds are on the clock ---------
    stage_keys = ("stage1_ms", "stage2_ms", "stage3_ms", "stage4_ms", "total_ms", "s3_gpu_ms", "s3_list_ms",
                  "s3_bucket_ms", "pairs", "trigger_jobs", "dropin_stacks", "jobs_completed_on_demand", "rounds")
    stage = {k: 0.0 for k in stage_keys}
    nstage = 0
    block_s, rank_block_s = [], []
    total = 0.0
    while True:
        dt, own, tms = timed_block(lambda: run_steps(args.steps))
        rank_block_s.append(own)
        block_s.append(dt)
        total += dt
        for tm in tms:
            for kk in stage:
                stage[kk] += tm[kk]
            nstage += 1
        # (dt is the max over ranks, so every rank takes the same decision)
        if total >= args.min_seconds or len(block_s) >= args.max_blocks:
            break
    bs = sorted(block_s)
    med_s = bs[len(bs) // 2] if len(bs) % 2 else 0.5 * (bs[len(bs) // 2 - 1] + bs[len(bs) // 2])
    my = sorted(rank_block_s)
    per_rank_ms = gather_floats(my[len(my) // 2] / args.steps * 1e3, world, rank, dist)
    if not args.dry:
        for p_ in pipes[:max(1, min(ninfl, args.steps))]:
            assert p_.summary() == fingerprint, "results changed between steps"
    n_trig = sum(1 for r in fingerprint if r[0] == 0)
    n_bub = sum(r[2] for r in fingerprint)

    frames_per_step = S * F * world
    out = {
        "metric": f"frames/s end-to-end detect @{W}x{H} 8-bit",  # BASELINE.json's metric at the default 1280x1024
        "value": frames_per_step * args.steps / med_s,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": med_s / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "regime": args.regime,
            "workload": f"synthetic 40l-19-like run: {E} events x {C} cams x {F} frames {W}x{H} u8 per GPU, HBM-resident; "
                        "per step: trigger search over every frame, genesis localisation, <=10-frame tracking, per-bubble records",
            "events_per_gpu": E, "cams": C, "frames_per_stack": F, "width": W, "height": H,
            "parallelism": f"events dealt round-robin over {world} GPU(s) (event % N), no collective",
            "host_threads": args.threads, "host_threads_per_pipeline": pipe_threads,
            "warmup_steps_run": nwarm,
            "steps_in_flight": ninfl,  # stage_ms below are wall times inside one step: with >1 in flight they include
                                       # queueing behind the other steps' kernels and no longer add up to ms_per_step
            "timing": {"blocks": len(block_s), "steps_per_block": args.steps, "timed_seconds": total,
                       "ms_per_step_median": med_s / args.steps * 1e3, "ms_per_step_min": bs[0] / args.steps * 1e3,
                       "ms_per_step_max": bs[-1] / args.steps * 1e3, "ms_per_step_first_block": block_s[0] / args.steps * 1e3,
                       "ms_per_step_by_rank": per_rank_ms,
                       "note": "value and ms_per_step are the median K-step block (max over ranks per block)"},
            "triggered_stacks": n_trig, "bubbles": n_bub,
            "stage_ms": {k: round(v / max(1, nstage), 3) for k, v in stage.items()},
            "gen_seconds": round(gen_s, 1),
        },
    }
    if not args.dry:
        lazy = os.environ.get("ABUB_PIPE_LAZY", "1") != "0"
        out["config"]["trigger_search"] = {
            "lazy": lazy, "jobs_per_step": stage["trigger_jobs"] / max(1, nstage), "jobs_if_every_frame": S * (F - 1),
            "note": "the reference walks a stack's frames in order and stops at the trigger (AnalyzerUnit.cpp:191, break at "
                    ":307): the pipeline evaluates the frame differences block by block -- the first block (frames 1 .. F/2+4) "
                    "for every stack, later blocks only for the stacks whose search reaches them (ABUB_PIPE_LAZY=0: every frame "
                    "of every stack up front, round 2's behaviour); results are identical either way (tests)"}
        out["config"]["slabs"] = {"copies_of_the_run_in_hbm": len(slabs), "bytes_each": slab.numel(),
                                  "note": "step k reads copy k % copies: steps in flight at the same time never share addresses"}
        out["config"]["masks"] = None if args.no_masks else {
            "maskdir": "synthetic cam<N>_mask.bmp (fiducial ellipse, 79 % of the frame) for every camera and cam1_bellows_mask.bmp "
                       "(strip, 12 % of the frame), 1-bit BMP like cam_masks/40l-19; bubbles are generated inside the fiducial "
                       "region and outside the bellows strip",
            "stacks_through_the_bellows_fallback_per_step": stage["dropin_stacks"] / max(1, nstage)}
    if args.dry:
        out["data"] = "none (dry protocol rehearsal, no kernels)"
        out["config"]["dry"] = True
        if rank == 0:
            emit(out)
        if dist:
            dist.destroy_process_group()
        return

    # ---- latency: the same step, one at a time (no overlap between steps) -----------------------------------------
    if args.latency_steps > 0:
        lat = []
        for _ in range(args.latency_steps):
            torch.cuda.synchronize()
            tl = time.perf_counter()
            pipe.run(slab, mu_d, s6_d, stream)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - tl) * 1e3)
        lat.sort()
        out["config"]["latency_one_step_at_a_time_ms"] = {"median": lat[len(lat) // 2], "min": lat[0], "steps": len(lat)}

    # ---- PCIe-inclusive rate (reported in config only, never `value`): the same run streamed from pinned host memory
    pcie = None
    if args.stream_steps > 0 and rank == 0 and world == 1:
        h_slab = torch.empty(slab.shape, dtype=torch.uint8).pin_memory()
        h_slab.copy_(slab)
        torch.cuda.synchronize()
        nstream = min(2, len(pipes))
        sring = host.PipelineRing.__new__(host.PipelineRing)  # the first pipelines of the ring, each with its own HBM slab
        sring.device, sring.pipes = local, pipes[:nstream]
        sring.run_batches([h_slab] * nstream, mu_d, s6_d, host=True)  # warm-up (allocates the pipeline-owned slabs)
        assert all(p_.summary() == fingerprint for p_ in sring.pipes), "streamed run differs from the resident run"
        ts = time.perf_counter()
        sring.run_batches([h_slab] * args.stream_steps, mu_d, s6_d, host=True)
        tstream = (time.perf_counter() - ts) / args.stream_steps
        pcie = {"frames_per_s": S * F / tstream, "ms_per_run": tstream * 1e3, "runs": args.stream_steps, "runs_in_flight": nstream,
                "GBps_host_to_hbm": S * F * P / tstream / 1e9,
                "note": "every run starts in pinned host memory; stack groups are uploaded with hipMemcpyAsync on a copy stream "
                        "while earlier groups / the previous run are in their detect stages"}
        del h_slab
    out["config"]["pcie_inclusive"] = pcie

    # ---- decode-inclusive rate: the first events of the run as a PNG zip archive on local disk -> ZipParser -> pinned
    # batches -> the same pipeline (what `abub3hs -z` does; reported in config only)
    if args.ingest_events > 0 and rank == 0 and world == 1:
        out["config"]["ingest_inclusive"] = ingest_inclusive(args, slab, pipe, E, C, F, W, H)
        out["config"]["png_decode_on_gpu"] = png_decode_leg(torch, slab, C, F, W, H)

    # ---- dominant kernel alone, HIP events on the launch stream (roofline object) ---------------
    njobs = S * (F - 1)
    jobs = hip.stack_jobs(S, F, 1, F - 1, 2, C, dev)
    hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
    kreps = max(5, min(20, args.steps))
    chain = (F - 1, 2)  # the pipeline's own call: stack-structured job list, FindTriggerFrame's two-frame offset

    def time_launch(fn, reps):
        fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a_, b_ in evs:
            a_.record()
            fn()
            b_.record()
        torch.cuda.synchronize()
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        return sum(ts) / len(ts), ts[0]

    k2_ms, k2_min = time_launch(lambda: hip.diff_hist(slab, s6_d, jobs, W, H, store=False, hist=hist, chain=chain), kreps)
    default_pieces, _ = hip.bound_counts()
    hist_h = hist.cpu()
    comp_bytes = 1.0 * P * njobs  # compulsory: every frame of a chain read once (the pass does not write D)
    alg_bytes = 3.0 * P * njobs   # SURVEY 8(d) accounting: cur, ref and sigma6 charged per job
    achieved = comp_bytes / (k2_ms * 1e-3) / 1e9
    # HBM traffic of THIS pass from rocprofv3 --pmc runs wrapping `python3 bench.py` itself (tools/prof_bench_pmc.sh:
    # FETCH_SIZE x2 + WRITE_SIZE of the pass's kernels, per launch); null if no summary matches this geometry
    traffic, traffic_src = None, None
    prof_root = os.path.join(ROOT, "profiles")
    for d in sorted((d for d in os.listdir(prof_root) if os.path.isdir(os.path.join(prof_root, d))), reverse=True):
        for name in sorted(n for n in os.listdir(os.path.join(prof_root, d)) if n.startswith("bench") and n.endswith("pmc_summary.json")):
            cand = os.path.join(prof_root, d, name)
            pm = json.load(open(cand))
            if pm.get("W") == W and pm.get("H") == H and pm.get("jobs_per_launch") == njobs:
                traffic = pm["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(cand, ROOT)
                break
        if traffic is not None:
            break
    out["roofline"] = {
        "kernel": "K2 trigger-only pass over EVERY frame of the run = k2_sad_chain (dominant: bound scan on v_sad_u8 group masses + "
                  "the waves' own exact tails) + k2_rows on handed-over rows + k_hist_bin0, timed together with HIP events on the "
                  "launch stream: fused ProcessFrame + 256-bin histogram of every frame pair of the run (the pipeline itself "
                  "launches it block by block and skips the frames behind a stack's trigger: config.trigger_search)",
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
        "bytes_accounting": "compulsory: 1*W*H per job (each frame of a chain is read from HBM once; D is not written)",
        "bytes_per_launch": comp_bytes, "ms_per_launch": k2_ms, "ms_per_launch_min": k2_min, "jobs_per_launch": njobs,
        "frac_of_traffic": (traffic / (k2_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
        "contract_algorithmic": {"bytes_per_job": 3 * P, "GBps": alg_bytes / (k2_ms * 1e-3) / 1e9,
                                 "note": "SURVEY 8(d) figure (cur + ref + sigma6 charged per job); exceeds the peak because "
                                         "the second use of a frame and sigma6 never reach HBM -- not a roofline fraction"},
    }

    # ---- BASELINE configs[2]: the fused kernel on a contiguous 10k-frame slab, i = 2..F-1, ref = i-2 -------------
    # (the single-GPU extras -- microbench, CPU baseline, regimes, ingestion -- run at N = 1 only: in a multi-rank run every
    # rank is done at the same point and nobody waits for rank 0)
    if args.micro_frames > 4 and rank == 0 and world == 1:
        out["config"]["microbench"] = microbench(args, torch, hip, dev, W, H)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, np, torch, slab, mu_d, sg_d, hist_h, pipe, tss, S, F, C,
                                           None if args.no_masks else synth.camera_masks)

    # ---- other data regimes (reported in config only): the same run with (a) everything behind the trigger dense
    # ("decompression": 35 % of the image differs by > 6 sigma from frame t0 + 3 on) and (b) hot pixels at ten times the
    # default's density of supra-threshold pixels; end-to-end rate, the K2 pass over EVERY frame alone, hand-over pieces
    if args.regime_steps > 0 and rank == 0 and world == 1 and args.regime == "default":
        out["config"]["regimes"] = {}
        del slabs[1:]
        for name in ("post_trigger_dense", "noisy"):
            sl, mu_r, sg_r, s6_r = make_run(name)
            for p_ in pipes:
                p_.set_sigma(sg_r)
            ring.run_batches([sl] * len(pipes), mu_r, s6_r, stream)  # every pipeline object once, untimed
            torch.cuda.synchronize()
            tr = time.perf_counter()
            tms_r = ring.run_batches([sl] * args.regime_steps, mu_r, s6_r, stream)
            torch.cuda.synchronize()
            dt_r = (time.perf_counter() - tr) / args.regime_steps
            fp_r = pipe.summary()
            ms_all, _ = time_launch(lambda: hip.diff_hist(sl, s6_r, jobs, W, H, store=False, hist=hist, chain=chain), 4)
            pieces, _ = hip.bound_counts()
            out["config"]["regimes"][name] = {
                "value_frames_per_s": S * F / dt_r, "ms_per_step": dt_r * 1e3, "steps": args.regime_steps,
                "trigger_jobs_per_step": sum(t["trigger_jobs"] for t in tms_r) / len(tms_r),
                "triggered_stacks": sum(1 for r in fp_r if r[0] == 0), "bubbles": sum(r[2] for r in fp_r),
                "stage_ms": {k: round(sum(t[k] for t in tms_r) / len(tms_r), 3) for k in stage_keys},
                "k2_pass_over_every_frame": {"ms": ms_all, "jobs": njobs, "handed_over_pieces_of_32_rows": pieces,
                                             "frac_of_8TBps_compulsory": 1.0 * P * njobs / (ms_all * 1e-3) / 1e9 / HBM_PEAK_GBPS},
            }
            del sl
        out["config"]["regimes"]["default"] = {
            "value_frames_per_s": out["value"], "ms_per_step": out["ms_per_step"],
            "k2_pass_over_every_frame": {"ms": out["roofline"]["ms_per_launch"], "jobs": njobs,
                                         "handed_over_pieces_of_32_rows": default_pieces}}
    if rank == 0:
        emit(out)
    if dist:
        dist.barrier()  # all ranks leave together
        dist.destroy_process_group()


def png_decode_leg(torch, slab, C, F, W, H):
    """abub_png_decode_dev alone: one batch of 4 x CU-count encoded frames (what the inflate kernel runs at a time) resident
    in HBM -> frames, timed with HIP events; the pixels are checked against the frames that were encoded."""
    import io

    import numpy as np
    from PIL import Image

    from autobub3hs_amd import _lib, hip

    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    n = 4 * ncu
    src = slab[0].cpu().numpy()  # [F, H, W] of stack 0
    enc = []
    for k in range(F):
        b = io.BytesIO()
        Image.fromarray(src[k]).save(b, format="PNG", compress_level=1)
        enc.append(b.getvalue())
    files = [enc[k % F] for k in range(n)]
    frames_np = np.zeros((n, 8), dtype=np.uint32)
    segs, blob, zoff, P = [], bytearray(), 0, W * H
    for i, data in enumerate(files):
        sg, _ = hip.png_parse(data, W, H)
        base = len(blob)
        zlen = sum(l for _, l in sg)
        frames_np[i] = (len(segs), len(sg), zoff, zlen, 0xFFFFFFFF, 0, (i * P) & 0xFFFFFFFF, (i * P) >> 32)
        segs += [(base + o, l) for o, l in sg]
        blob += data
        blob += b"\0" * ((-len(blob)) % 4)
        zoff += ((zlen + 15) & ~15) + 16
    blob += b"\0" * 8
    dev = slab.device
    d_files = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    d_frames = torch.from_numpy(frames_np.view(np.int32).copy()).to(dev)
    d_segs = torch.tensor(segs, dtype=torch.int64).to(torch.int32).to(dev)
    d_luts = torch.zeros(256, dtype=torch.uint8, device=dev)
    stride = int(_lib.lib().abub_png_raw_stride(W, H))
    d_z = torch.empty((zoff + 16,), dtype=torch.uint8, device=dev)
    d_raw = torch.empty((n * stride,), dtype=torch.uint8, device=dev)
    out = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    status = torch.zeros((n,), dtype=torch.int32, device=dev)

    def go():
        _lib.check(_lib.lib().abub_png_decode_dev(d_files.data_ptr(), d_files.numel(), d_frames.data_ptr(), n, d_segs.data_ptr(), len(segs),
                                                  d_luts.data_ptr(), 0, W, H, d_z.data_ptr(), d_z.numel(), d_raw.data_ptr(), d_raw.numel(),
                                                  out.data_ptr(), out.numel(), status.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "abub_png_decode_dev")

    go()
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0, "abub_png_decode_dev refused a frame"
    for k in (0, F - 1, n - 1):
        assert torch.equal(out[k], slab[0, k % F]), "GPU PNG decode differs from the encoded frame"
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = min(ts)
    return {"frames_per_batch": n, "ms_per_batch": ms, "frames_per_s": n / ms * 1e3, "GBps_of_pixels": n * P / ms / 1e6,
            "encoded_MB_per_frame": len(blob) / n / 1e6,
            "note": "gather + inflate (a parsing and a writing wave per frame, 4 frames per CU) + unfilter on resident files; "
                    "8-bit grey PNG, compress_level 1; the ingest figure above includes reading the archive, the upload and the detect stages"}


def ingest_inclusive(args, slab, pipe, E, C, F, W, H):
    """Write the first events of the run as <tmp>/<run>.zip (8-bit grey PNGs, the 40l-19 layout), then train and detect
    from the archive with host.Run(kind="zip") + run_batched (ZipParser -> PNG decode on host threads -> pinned batches
    -> hipMemcpyAsync -> RunPipeline -> OutputWriter)."""
    import io
    import shutil
    import tempfile
    import zipfile
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image

    from autobub3hs_amd import host

    nev = min(args.ingest_events, E)
    run_id = "20200925_0"
    tmp = tempfile.mkdtemp(prefix="abub_ingest_")
    try:
        t0 = time.perf_counter()
        frames = slab[: nev * C].cpu().numpy()  # [nev*C, F, H, W]

        def enc(job):
            s, k = job
            b = io.BytesIO()
            Image.fromarray(frames[s, k]).save(b, format="PNG", compress_level=1)
            return s, k, b.getvalue()

        with ThreadPoolExecutor(min(32, usable_cores())) as ex:
            blobs = list(ex.map(enc, [(s, k) for s in range(nev * C) for k in range(F)]))
        zpath = os.path.join(tmp, run_id + ".zip")
        zbytes = 0
        with zipfile.ZipFile(zpath, "w", zipfile.ZIP_STORED) as z:
            for e in range(nev):
                z.writestr(f"{run_id}/{e}/", b"")
                z.writestr(f"{run_id}/{e}/Images/", b"")
            for s, k, data in blobs:
                z.writestr(f"{run_id}/{s // C}/Images/cam{s % C}_image{30 + k}.png", data)
                zbytes += len(data)
        t_make = time.perf_counter() - t0
        del blobs
        ncore = usable_cores()
        nthr = min(args.threads, ncore)                                          # host stages of the detect pipeline
        ndec = args.decode_threads if args.decode_threads > 0 else min(ncore, 128)  # PNG decode (AutoBubStart3.cpp:338-342
        t1 = time.perf_counter()                                                  # runs omp_get_max_threads() events at once)
        run = host.Run(kind="zip", run_folder=os.path.join(tmp, run_id))
        tr = [run.train(c, shape=(H, W)) for c in range(C)]
        t_train = time.perf_counter() - t1
        assert all(t[0] == 0 for t in tr), "training from the archive failed"
        t2 = time.perf_counter()
        stats = run.run_batched(C, tmp + "/", run_id, 30, nthreads=nthr, decode_threads=ndec)
        t_detect = time.perf_counter() - t2
        run.close()
        nrows = sum(1 for _ in open(os.path.join(tmp, f"abub3hs_{run_id}.txt")))
        return {
            "frames_per_s": nev * C * F / t_detect, "frames": nev * C * F, "events": nev,
            "source": f"zip archive on local disk, {zbytes / 1e6:.0f} MB of 8-bit grey PNG (compress_level 1), ZipParser; frames decoded "
                      "on the GPU (abub_png_decode_dev) unless ABUB_GPU_DECODE=0 (then: own PNG decoder on host threads)",
            "frames_decoded_on_gpu": int(stats["frames_gpu_decoded"]), "frames_decoded_on_host": int(stats["frames_host_decoded"]),
            "decode_threads": ndec, "host_threads": nthr, "cores_available": ncore, "seconds": {"detect_total": t_detect, "list": stats["list_s"], "decode": stats["decode_s"],
                                                "upload_gpu_host_stages": stats["gpu_s"], "gpu_decode_of_that": stats["gpudecode_s"], "write": stats["write_s"],
                                                "training_from_archive": t_train, "making_the_archive": t_make},
            "frames_per_s_decode_only": nev * C * F / max(stats["decode_s"], 1e-9),
            "batches": int(stats["batches"]), "output_rows": nrows,
            "note": "detect_total = list + decode (reading the files, or reading + host decode; overlapped with the GPU from the second "
                    "batch on) + upload + GPU decode + detect + write; "
                    "training (2 frames per event and camera, decoded separately) is outside the figure like in the resident run",
        }
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def microbench(args, torch, hip, dev, W, H):
    """SURVEY 8(d) "Config 3": slab [N][H][W] in HBM + sigma6, K2 over i = 2..N-1 with ref = i-2, D and [count][256]
    histograms written (store mode), >= 3 timed repetitions after a warm-up, HIP events on the launch stream."""
    P = W * H
    N = args.micro_frames
    free, _ = torch.cuda.mem_get_info()
    need = 2 * N * P + (64 << 20)
    if need > free * 0.9:
        N = max(64, int((free * 0.9 - (64 << 20)) // (2 * P)))
    from autobub3hs_amd import synth

    slab = synth.long_stack(N, W, H, xp="torch", device=dev)
    sg = torch.ones((1, H, W), dtype=torch.uint8, device=dev)
    s6 = hip.sigma6(sg)
    njobs = N - 2
    jobs = hip.make_jobs([(i, i - 2, 0, i - 2) for i in range(2, N)], dev)
    hist = torch.empty((njobs, 256), dtype=torch.int32, device=dev)
    D = torch.empty((njobs, H, W), dtype=torch.uint8, device=dev)

    def timeit(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a_, b_ in evs:
            a_.record()
            fn()
            b_.record()
        torch.cuda.synchronize()
        ts = sorted(a_.elapsed_time(b_) for a_, b_ in evs)
        return sum(ts) / len(ts), ts[0]

    res = {"frames": N, "jobs_per_launch": njobs, "width": W, "height": H, "sigma": 1,
           "data": "synth.long_stack: background + sum of four U{-1,0,1} per pixel and frame (about 3.5 supra-6-sigma noise "
                   "pixels per row) + a disc of radius <= 33 growing in the second half of every 64-frame block"}

    def row(ms, ms_min, nP):
        gbps = nP * P * njobs / (ms * 1e-3) / 1e9
        return {"ms_per_launch": ms, "ms_per_launch_min": ms_min, "us_per_job": ms * 1e3 / njobs,
                "compulsory_bytes_per_job": nP * P, "compulsory_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS}

    chain = (njobs, 2)
    ms, mn = timeit(lambda: hip.diff_hist(slab, s6, jobs, W, H, store=True, hist=hist, diff=D, chain=chain))
    res["store_mode"] = row(ms, mn, 2)  # read every frame once, write D once
    h_store = hist.clone()
    # size-independent checks at full size: totals, bincount(D) == histogram on sampled jobs
    assert bool((h_store.sum(1) == P).all()), "microbench: histogram totals"
    for k in range(0, njobs, max(1, njobs // 6)):
        assert torch.equal(torch.bincount(D[k].flatten().to(torch.int64), minlength=256).to(torch.int32), h_store[k])
    ms, mn = timeit(lambda: hip.diff_hist(slab, s6, jobs, W, H, store=False, hist=hist, chain=chain))
    res["trigger_only"] = row(ms, mn, 1)
    assert torch.equal(hist, h_store), "microbench: trigger-only histograms differ from store mode"
    # the dense-regime worst case: every row through the full row machine (what a frame of dense foreground costs)
    hip.k2_set_option("bound", 0)
    try:
        ms, mn = timeit(lambda: hip.diff_hist(slab, s6, jobs, W, H, store=True, hist=hist, diff=D), reps=3)
        res["store_mode_row_machine_only"] = row(ms, mn, 2)
        assert torch.equal(hist, h_store), "microbench: row-machine histograms differ"
        ms, mn = timeit(lambda: hip.diff_hist(slab, s6, jobs, W, H, store=False, hist=hist), reps=3)
        res["trigger_only_row_machine_only"] = row(ms, mn, 1)
    finally:
        hip.k2_set_option("bound", 1)
    res["nonzero_pixels_per_frame"] = float((P - h_store[:, 0].double()).mean().item())
    # for scale: what this chip's memory system gives a plain device-to-device copy of the frames into D (torch's copy
    # kernel: one read + one write per byte, no arithmetic) -- the store mode above moves the same compulsory bytes
    nb = min(njobs, D.shape[0]) * P
    ms, mn = timeit(lambda: D.view(-1)[:nb].copy_(slab.view(-1)[:nb]))
    res["plain_copy_same_bytes"] = {"ms": ms, "ms_min": mn, "read_plus_write_GBps": 2.0 * nb / (ms * 1e-3) / 1e9,
                                    "frac_of_8TBps": 2.0 * nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "store_mode_vs_copy": (2.0 * P * njobs / (res["store_mode"]["ms_per_launch"] * 1e-3)) /
                                                          (2.0 * nb / (ms * 1e-3)),
                                    "note": "tools/rowload_bench.cpp (profiles/*/rowload.jsonl) holds the hand-written flat read "
                                            "and copy kernels: 6.4 TB/s read, 4.9-5.2 TB/s read+write on this chip"}
    res["contract_algorithmic_GBps"] = {"store_mode_4P": 4.0 * P * njobs / (res["store_mode"]["ms_per_launch"] * 1e-3) / 1e9,
                                        "trigger_only_3P": 3.0 * P * njobs / (res["trigger_only"]["ms_per_launch"] * 1e-3) / 1e9}
    del slab, D
    return res


def cpu_baseline(args, np, torch, slab, mu_d, sg_d, hist_h, pipe, tss, S, F, C, masks_for=None):
    """CPU baseline = the oracle's whole detect path (AnyCamAnalysis restatement) on a bounded sample of the same
    stacks, 1 core.  It doubles as a full-size parity check of the GPU results."""
    from oracle import pyoracle as orc

    orc.build()
    done, tcpu, nstk = 0, 0.0, 0
    s_i = 0
    mu_h = mu_d.cpu().numpy()
    sg_h = sg_d.cpu().numpy()
    H_, W_ = slab.shape[2], slab.shape[3]
    mk = {c: (masks_for(W_, H_, c) if masks_for else (None, None)) for c in range(C)}
    while tcpu < args.cpu_seconds and s_i < S:
        st = slab[s_i].cpu().numpy()
        c = s_i % C
        tc = time.perf_counter()
        a = orc.Analyzer(st, mu_h[c], sg_h[c], tss[c], fid_mask=mk[c][0], bel_mask=mk[c][1])
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

    ncore = max(1, min(usable_cores(), 32, S))
    stacks_mc = [(slab[s].cpu().numpy(), s % C) for s in range(0, S, max(1, S // ncore))][:ncore]

    def _one(item):
        st, c = item
        a = orc.Analyzer(st, mu_h[c], sg_h[c], tss[c], fid_mask=mk[c][0], bel_mask=mk[c][1])  # ctypes drops the GIL inside the oracle
        a.any_cam_analysis()
        a.close()

    with ThreadPoolExecutor(len(stacks_mc)) as ex:
        tc = time.perf_counter()
        list(ex.map(_one, stacks_mc))
        tmc = time.perf_counter() - tc
    return {
        "value": done / tcpu, "unit": "frames/s", "cores": 1, "kind": "port",
        "all_cores": {"value": len(stacks_mc) * F / tmc, "cores": len(stacks_mc),
                      "sample": f"{len(stacks_mc)} stacks at once, one per thread, {tmc:.1f} s"},
        "sample": f"{nstk} stacks ({done} frames) of the same workload through the oracle's end-to-end detect "
                  f"(oracle/abub_oracle.c, {orc.CFLAGS_NOTE}, results identical to the GPU's), {tcpu:.1f} s on 1 core of {os.cpu_count()}",
    }


if __name__ == "__main__":
    main()
