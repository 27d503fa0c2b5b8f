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


WIDTH = 4 + MAX_ROWS + 1


def pack_summary(summary):
    """Engine.summary() dict -> one int64 array [n_local, WIDTH] (n_rows, n_baselines, stop, nan_levels, knot counts)."""
    n = len(summary["n_rows"])
    out = np.empty((n, WIDTH), np.int64)
    out[:, 0] = summary["n_rows"]
    out[:, 1] = summary["n_baselines"]
    out[:, 2] = summary["stop"]
    out[:, 3] = summary.get("nan_levels", -1)      # -2: that signal's input held a NaN (rejected on its rank)
    out[:, 4:] = summary["knot_counts"]
    return out


def unpack_summary(arr):
    return {"n_rows": arr[:, 0].astype(np.int32), "n_baselines": arr[:, 1].astype(np.int32),
            "stop": arr[:, 2].astype(np.int32), "nan_levels": arr[:, 3].astype(np.int32), "knot_counts": arr[:, 4:].copy()}


def gather_summaries(local_summary, batch, group=None, device=None):
    """All-gather the per-signal summaries of every rank's shard, in batch order.  Every rank gets the whole table."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    width = WIDTH
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


class ShardedBatch:
    """One rank's part of a batch of `batch` independent signals of `n` samples, sharded over `world` ranks (one process
    per GPU).  The rank owns signals [lo, hi) = shard_range(batch, world, rank), decomposes them with ITS engine in one
    launch sequence and takes part in ONE collective: the all-gather of the per-signal summaries.  Samples never leave the
    GPU that produced them (full float64 outputs stay sharded).

        sb = ShardedBatch(8192, 1 << 20, 7, world, rank, device=local_rank)
        sb.decompose(x_local_ptr, numpy.float32, x_stride, rows_local_ptr)       # pointers to the LOCAL shard
        table = sb.gather(device=torch.device("cuda", local_rank))                # every rank: all 8192 summaries

    `engine` may be injected (tests run the sharding logic against a stand-in on CPU ranks).
    """

    def __init__(self, batch, n, max_iteration, world, rank, device=0, engine=None, local_limit=None):
        self.batch, self.n, self.max_iteration = int(batch), int(n), int(max_iteration)
        self.world, self.rank = int(world), int(rank)
        self.lo, self.hi = shard_range(self.batch, self.world, self.rank)
        self.n_local = self.hi - self.lo
        # local_limit (tests, rehearsals): only the first `local_limit` signals of the rank's real range [lo, hi) are
        # decomposed — the range itself stays the real one; such an object cannot take part in the all-gather
        self.local_limit = None if local_limit is None else int(local_limit)
        if self.local_limit is not None:
            self.n_local = min(self.n_local, self.local_limit)
        if engine is None:
            from .engine import Engine
            engine = Engine(self.n, max(self.n_local, 1), device)
        self.engine = engine

    def decompose(self, x_ptr, dtype, x_stride, rows_ptr, baselines_ptr=None, stream=None):
        """Enqueue the decomposition of the local shard (no host sync, no communication)."""
        if self.n_local:
            self.engine.decompose_dev(x_ptr, dtype, self.n, self.n_local, x_stride, self.max_iteration, rows_ptr,
                                      baselines_ptr, stream)

    def local_summary(self):
        if self.n_local:
            return self.engine.summary(self.n_local)
        return {"n_rows": np.zeros(0, np.int32), "n_baselines": np.zeros(0, np.int32), "stop": np.zeros(0, np.int32),
                "nan_levels": np.zeros(0, np.int32), "knot_counts": np.zeros((0, MAX_ROWS + 1), np.int64)}

    def gather(self, group=None, device=None):
        """Synchronise with the local decomposition and all-gather the summaries: the whole batch's table, in batch order."""
        if self.local_limit is not None and self.n_local != self.hi - self.lo:
            raise ValueError("a ShardedBatch with local_limit decomposes only part of its range: no all-gather")
        return gather_summaries(self.local_summary(), self.batch, group=group, device=device)

    def scatter_from(self, root, x_root=None, group=None, out=None):
        """The batch lives on ONE rank (`root`: x_root = its [batch, n] tensor, None elsewhere): hand every rank its contiguous
        shard — the north star's "trivial batch scatter" — as ONE group of point-to-point transfers (grouped send / recv:
        ncclSend / ncclRecv over xGMI under RCCL, all of the root's links busy at once; gloo on CPU tensors in the tests).
        Returns this rank's [n_local, n] tensor (`out` if given).  Ranks that synthesise or load their own shard (the bench
        default) never call this."""
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        if world != self.world or rank != self.rank:
            raise ValueError("process group and ShardedBatch disagree on world / rank")
        if rank == root:
            if x_root is None or tuple(x_root.shape) != (self.batch, self.n):
                raise ValueError("the root passes the whole batch [%d, %d]" % (self.batch, self.n))
            mine = x_root[self.lo:self.hi]
            ops = []
            for r in range(world):
                lo, hi = shard_range(self.batch, world, r)
                if r != root and hi > lo:
                    ops.append(dist.P2POp(dist.isend, x_root[lo:hi].contiguous(), r, group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if out is None:
                return mine
            out.copy_(mine)
            return out
        if out is None:
            raise ValueError("a receiving rank passes `out`: its [n_local, n] tensor of the batch's dtype, on its device")
        if self.hi > self.lo:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, out, root, group)]):
                w.wait()
        return out
