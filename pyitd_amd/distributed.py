"""Sharding of a batch of independent signals over the GPUs of one node (one process per GPU).

The ITD path has no exchange step: signals are independent units, so ranks own contiguous ranges of the
batch and run the engine on their shard with NO data-path collective.  The only communication
is the gather of the per-signal summaries (rows, stop reason, knots per level: a few KB), done with
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
A single long signal is never split: its level recursion is serial (ITD.py:431).
"""
import numpy as np

MAX_ROWS = 22


def shard_range(batch, world_size, rank):
    """Contiguous, balanced range [lo, hi) of the batch owned by `rank` (the first batch % world ranks get one more)."""
    if batch < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad shard request")
    q, r = divmod(batch, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_summary(summary):
    """Engine.summary() dict -> one int64 array [n_local, 3 + MAX_ROWS + 1] (n_rows, n_baselines, stop, knot counts)."""
    n = len(summary["n_rows"])
    out = np.empty((n, 3 + MAX_ROWS + 1), np.int64)
    out[:, 0] = summary["n_rows"]
    out[:, 1] = summary["n_baselines"]
    out[:, 2] = summary["stop"]
    out[:, 3:] = summary["knot_counts"]
    return out


def unpack_summary(arr):
    return {"n_rows": arr[:, 0].astype(np.int32), "n_baselines": arr[:, 1].astype(np.int32),
            "stop": arr[:, 2].astype(np.int32), "knot_counts": arr[:, 3:].copy()}


def gather_summaries(local_summary, batch, group=None, device=None):
    """All-gather the per-signal summaries of every rank's shard, in batch order.  Every rank gets the whole table."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    width = 3 + MAX_ROWS + 1
    cap = -(-batch // world)                      # equal-sized slots: all_gather needs one shape
    lo, hi = shard_range(batch, world, rank)
    slot = torch.full((cap, width), -1, dtype=torch.int64, device=device)
    if hi > lo:
        slot[: hi - lo] = torch.from_numpy(pack_summary(local_summary)).to(slot.device)
    parts = [torch.empty_like(slot) for _ in range(world)]
    dist.all_gather(parts, slot, group=group)
    rows = []
    for r, p in enumerate(parts):
        l, h = shard_range(batch, world, r)
        rows.append(p[: h - l].cpu().numpy())
    return unpack_summary(np.concatenate(rows, axis=0))
