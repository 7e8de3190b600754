"""Event sharding across ranks (one process per GPU).  Detection is independent per (event, camera) given
the per-camera model (reference AutoBubStart3.cpp:342-388), so there is NO data-path collective: events are
dealt round-robin (`event_index % world`, mirroring the reference's `schedule(static,1)`), every rank
processes its own events, and only the finished output rows / the timing scalar cross ranks.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used for exactly that: barrier, max-reduce of the
elapsed time, and an ordered gather of per-event result rows to rank 0 (the `ordered` clause :380-383)."""
import torch.distributed as dist


def events_for_rank(n_events, rank, world):
    """Indices (into the numerically sorted event list) owned by `rank`."""
    return list(range(rank, n_events, world))


def global_event_ids(events_per_rank, rank, world):
    """Weak-scaling benchmark layout: rank owns `events_per_rank` events, interleaved round-robin."""
    return [rank + k * world for k in range(events_per_rank)]


def max_over_ranks(seconds, device=None):
    import torch

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rows_in_event_order(rows, dst=0):
    """rows: list of (event_index, payload) produced by this rank.  Returns, on `dst`, the payloads of all
    ranks sorted by event index (the order the reference writes its output file); None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [p for _, p in sorted(rows, key=lambda r: r[0])]
    world = dist.get_world_size()
    out = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(rows, out, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged = [r for part in out for r in part]
    merged.sort(key=lambda r: r[0])
    return [p for _, p in merged]
