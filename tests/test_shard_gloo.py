"""N>1 path on CPU: world_size-2 gloo run of the sharding / ordered-gather / max-time logic that bench.py and
a multi-GPU driver use (the per-event work is done by the oracle here; on GPUs it is the HIP pipeline)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _analyse(event):
    sys.path.insert(0, ROOT)
    from autobub3hs_amd import synth
    from oracle import pyoracle as orc

    W, H, F = 160, 96, 24
    spec = synth.random_spec(W, H, F, event, 0, margin=20)
    fr = synth.render_event(W, H, spec, event, 0)
    tr = synth.training_pairs(W, H, 6, 0, F)
    mu, sg = orc.welford(tr)
    a = orc.Analyzer(fr, mu, sg, len(tr))
    staged, state, bubbles = a.any_cam_analysis()
    a.close()
    return (event, staged, state["trig"], [tuple(b["desc"][0][k] for k in "xywh") for b in bubbles])


def _worker(rank, world, port, n_events, q):
    sys.path.insert(0, ROOT)
    from autobub3hs_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.events_for_rank(n_events, rank, world)
    dist.barrier()
    rows = [(e, _analyse(e)) for e in mine]
    t = shard.max_over_ranks(1.0 + rank)
    merged = shard.gather_rows_in_event_order(rows, dst=0)
    if rank == 0:
        q.put((t, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_partition():
    from autobub3hs_amd import shard

    for n, w in [(10, 2), (7, 4), (3, 8), (100, 8)]:
        parts = [shard.events_for_rank(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert all(all(e % w == r for e in p) for r, p in enumerate(parts))
    assert shard.global_event_ids(3, 1, 4) == [1, 5, 9]
    assert shard.gather_rows_in_event_order([(2, "c"), (0, "a"), (1, "b")]) == ["a", "b", "c"]
    assert shard.max_over_ranks(2.5) == 2.5


def test_world_size_2_gloo_matches_single_process():
    n_events, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_events, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, merged = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0  # max over ranks of (1.0, 2.0)
    single = [_analyse(e) for e in range(n_events)]
    assert merged == single
    assert [m[0] for m in merged] == list(range(n_events))
