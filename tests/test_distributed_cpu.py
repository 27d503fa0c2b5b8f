"""N>1 path on CPU: two gloo processes run pyitd_amd.distributed.ShardedBatch — each decomposes ITS shard of the batch
(through an engine stand-in backed by the CPU oracle: no GPU here) and the summaries are all-gathered in batch order."""
import os
import socket

import numpy as np
import pytest

from helpers import sines_noise


def test_shard_range_partitions_the_batch():
    from pyitd_amd.distributed import shard_range
    for batch in (0, 1, 7, 8, 9, 1024, 8192):
        for world in (1, 2, 3, 4, 8):
            got = [shard_range(batch, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == batch
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in got]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleEngine:
    """Stand-in for pyitd_amd.Engine on a CPU rank: `decompose_dev` runs the CPU oracle on the shard it is handed (a numpy
    array instead of a device pointer), `summary` reports what the real engine's summary must equal."""

    def __init__(self):
        self.res = []

    def decompose_dev(self, x, dtype, n, batch, x_stride, max_iteration, rows, baselines=None, stream=None):
        from oracle import cpu_oracle
        self.res = [cpu_oracle.itd_lean(x[j], max_iteration) for j in range(batch)]
        for j, r in enumerate(self.res):
            rows[j, : r["rows"].shape[0]] = r["rows"]

    def summary(self, k):
        from pyitd_amd.distributed import MAX_ROWS
        assert k == len(self.res)
        out = {"n_rows": np.zeros(k, np.int32), "n_baselines": np.zeros(k, np.int32), "stop": np.zeros(k, np.int32),
               "nan_levels": np.full(k, -1, np.int32), "knot_counts": np.full((k, MAX_ROWS + 1), -1, np.int64)}
        for j, r in enumerate(self.res):
            out["n_rows"][j] = r["rows"].shape[0]
            out["stop"][j] = 0 if r["stop"] == "natural" else 1
            out["knot_counts"][j, : len(r["knot_counts"])] = r["knot_counts"]
        return out


def _worker(rank, world, port, batch, n, m, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyitd_amd.distributed import ShardedBatch
        sb = ShardedBatch(batch, n, m, world, rank, engine=_OracleEngine())
        x = np.stack([sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0) for b in range(sb.lo, sb.hi)]) if sb.n_local \
            else np.zeros((0, n), np.float32)
        rows = np.zeros((sb.n_local, m + 2, n))
        sb.decompose(x, np.float32, n, rows)          # the rank's own shard, no communication
        dist.barrier()
        full = sb.gather()                            # the one collective of the path
        if rank == 0:
            q.put({k: v.tolist() for k, v in full.items()})
    finally:
        dist.destroy_process_group()


def test_two_ranks_decompose_their_shards_and_gather_in_batch_order():
    import torch.multiprocessing as mp
    batch, n, m = 5, 4096, 4      # odd batch: ranks own 3 and 2 signals
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, batch, n, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import cpu_oracle
    assert len(got["n_rows"]) == batch and got["nan_levels"] == [-1] * batch
    for b in range(batch):
        r = cpu_oracle.itd_lean(sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0), m)
        assert got["n_rows"][b] == r["rows"].shape[0]
        assert got["knot_counts"][b][: len(r["knot_counts"])] == r["knot_counts"].tolist()


def _scatter_worker(rank, world, port, batch, n, m, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyitd_amd.distributed import ShardedBatch
        sb = ShardedBatch(batch, n, m, world, rank, engine=_OracleEngine())
        root = 1                                        # not rank 0: the root's own shard is the last one
        x_root = torch.from_numpy(np.stack([sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0) for b in range(batch)])) if rank == root else None
        out = torch.full((sb.n_local, n), -7.0, dtype=torch.float32)
        mine = sb.scatter_from(root, x_root, out=out)   # one group of point-to-point transfers
        x = mine.numpy()
        rows = np.zeros((sb.n_local, m + 2, n))
        sb.decompose(x, np.float32, n, rows)
        full = sb.gather()
        if rank == 0:
            q.put({"table": {k: v.tolist() for k, v in full.items()}, "first": x[0, :8].tolist(), "lo": sb.lo})
    finally:
        dist.destroy_process_group()


def test_scatter_from_one_rank_then_shard_and_gather():
    """north star: "RCCL over xGMI only for the trivial batch scatter/gather" — the batch lives on rank 1, every rank receives
    its contiguous shard, decomposes it and the summaries are all-gathered in batch order."""
    import torch.multiprocessing as mp
    batch, n, m = 5, 2048, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, 2, port, batch, n, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import cpu_oracle
    assert got["lo"] == 0 and got["first"] == sines_noise(n, seed=0)[:8].tolist()      # rank 0 received signal 0
    t = got["table"]
    assert len(t["n_rows"]) == batch
    for b in range(batch):
        r = cpu_oracle.itd_lean(sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0), m)
        assert t["n_rows"][b] == r["rows"].shape[0]
        assert t["knot_counts"][b][: len(r["knot_counts"])] == r["knot_counts"].tolist()


def test_local_limit_keeps_the_real_range():
    from pyitd_amd.distributed import ShardedBatch
    sb = ShardedBatch(8192, 1 << 20, 7, 8, 5, engine=_OracleEngine(), local_limit=4)
    assert (sb.lo, sb.hi, sb.n_local) == (5 * 1024, 6 * 1024, 4)
    with pytest.raises(ValueError):
        sb.gather()


def test_summary_packing_keeps_nan_levels():
    from pyitd_amd.distributed import MAX_ROWS, pack_summary, unpack_summary
    s = {"n_rows": np.array([3, 1], np.int32), "n_baselines": np.array([2, 0], np.int32), "stop": np.array([1, 0], np.int32),
         "nan_levels": np.array([-1, -2], np.int32), "knot_counts": np.full((2, MAX_ROWS + 1), 7, np.int64)}
    u = unpack_summary(pack_summary(s))
    for k in s:
        assert u[k].tolist() == s[k].tolist(), k
