"""N>1 path on CPU: two gloo processes shard a batch and gather the per-signal summaries (no GPU compute:
each rank fills its shard's summary from the CPU oracle, which is what the engine's summary must equal)."""
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


def _worker(rank, world, port, batch, n, m, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cpu_oracle
        from pyitd_amd.distributed import MAX_ROWS, gather_summaries, shard_range
        lo, hi = shard_range(batch, world, rank)
        k = hi - lo
        local = {"n_rows": np.zeros(k, np.int32), "n_baselines": np.zeros(k, np.int32), "stop": np.zeros(k, np.int32),
                 "knot_counts": np.full((k, MAX_ROWS + 1), -1, np.int64)}
        for j, b in enumerate(range(lo, hi)):
            r = cpu_oracle.itd_lean(sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0), m)
            local["n_rows"][j] = r["rows"].shape[0]
            local["stop"][j] = 0 if r["stop"] == "natural" else 1
            local["knot_counts"][j, : len(r["knot_counts"])] = r["knot_counts"]
        dist.barrier()
        full = gather_summaries(local, batch)
        if rank == 0:
            q.put({k: v.tolist() for k, v in full.items()})
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_summaries_in_batch_order():
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
    for b in range(batch):
        r = cpu_oracle.itd_lean(sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0), m)
        assert got["n_rows"][b] == r["rows"].shape[0]
        assert got["knot_counts"][b][: len(r["knot_counts"])] == r["knot_counts"].tolist()
