"""BASELINE configs at their full workloads, the 8-GPU sharding emulated on one GPU, a bounded fixed-seed slice of the
parity fuzz, and the engine properties round 2 added (chunked batches, NaN rules inside the extraction kernel, helper
workspace).  Everything goes through the C ABI; the CPU oracle is the checker."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from helpers import assert_bits_equal, canon_u64, fuzz_signal, sines_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.lib()
    return cpu_oracle


def _threads():
    return max(1, min(32, os.cpu_count() or 1))


def _batch_signal(b, n):
    """signal b of the config 3 / 4 batch: noise draw b mod 16, frequencies scaled by 1 + b/8192"""
    return sines_noise(n, seed=b % 16, fscale=1.0 + b / 8192.0)


def test_config3_full_size_batch_1024_x_2p20(P, torch, oracle):
    """BASELINE configs[2]: batch of 1024 signals x 2^20 float32 samples, 8 levels, one MI355X.
    Rows and per-level knot counts bit-exact vs the oracle on the 16 distinct noise draws + 16 random others; exact
    reconstruction (rows sum back to the input) and the stop summary for all 1024, checked on the device."""
    B, n, m = 1024, 1 << 20, 7
    R = m + 2
    with ThreadPoolExecutor(_threads()) as ex:
        x_host = np.stack(list(ex.map(lambda b: _batch_signal(b, n), range(B))))
    x = torch.from_numpy(x_host).cuda()
    rows = torch.empty((B, R, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, B, 0)
    torch.cuda.synchronize()
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
    s = eng.summary(B)
    assert (s["n_rows"] == R).all() and (s["stop"] == 1).all() and (s["nan_levels"] == -1).all()
    rng = np.random.default_rng(2024)
    check = list(range(16)) + sorted(rng.choice(np.arange(16, B), 16, replace=False).tolist())
    with ThreadPoolExecutor(_threads()) as ex:
        refs = list(ex.map(lambda b: oracle.itd_lean(x_host[b], m), check))
    for b, ref in zip(check, refs):
        assert ref["rows"].shape[0] == R
        assert s["knot_counts"][b, :R].tolist() == ref["knot_counts"].tolist(), "signal %d knots per level" % b
        assert_bits_equal(rows[b].cpu().numpy(), ref["rows"], "signal %d rows" % b)
    # all 1024, on the device: rows sum back to the input; the last sample is never moved (ITD.py:112-117)
    worst = 0.0
    for c0 in range(0, B, 64):
        rec = rows[c0:c0 + 64].sum(dim=1)
        worst = max(worst, float((rec - x[c0:c0 + 64].double()).abs().max()))
    assert worst < 1e-12
    assert bool((rows[:, 0, -1] == x[:, -1].double()).all()) and bool((rows[:, 1:, -1] == 0).all())
    eng.close()


def test_config4_sharding_emulated_on_one_gpu(P, torch, oracle):
    """BASELINE configs[3]: batch of 8192 x 2^20 sharded over 8 GPUs — emulated: for rank = 0..7 the rank's ShardedBatch
    (shard_range(8192, 8, rank)) decomposes the first signals of ITS shard (reduced per-rank batch) on this GPU; the
    concatenated summary table and one signal's rows per rank must equal the oracle's."""
    from pyitd_amd.distributed import ShardedBatch, shard_range, pack_summary, unpack_summary
    G, world, n, m, per_rank = 8192, 8, 1 << 20, 7, 16      # (16 signals of every rank's real range: 128 of the 8192)
    R = m + 2
    eng = P.Engine(n, per_rank, 0)
    tables, ids = [], []
    rows = torch.empty((per_rank, R, n), dtype=torch.float64, device="cuda")
    row_checks = []
    for rank in range(world):
        lo, hi = shard_range(G, world, rank)
        assert hi - lo == 1024 and lo == rank * 1024
        sb = ShardedBatch(G, n, m, world, rank, engine=eng, local_limit=per_rank)   # the rank's REAL range of the 8192-signal batch,
        assert (sb.lo, sb.hi, sb.n_local) == (lo, hi, per_rank)                      # of which the first per_rank signals run here
        mine = list(range(sb.lo, sb.lo + sb.n_local))
        x_host = np.stack([_batch_signal(b, n) for b in mine])
        x = torch.from_numpy(x_host).cuda()
        torch.cuda.synchronize()
        sb.decompose(x.data_ptr(), np.float32, n, rows.data_ptr())
        tables.append(pack_summary(sb.local_summary()))
        ids += mine
        row_checks.append((mine[-1], x_host[-1], rows[-1].cpu().numpy()))
    table = unpack_summary(np.concatenate(tables, axis=0))       # what the all-gather delivers, in batch order
    with ThreadPoolExecutor(_threads()) as ex:
        refs = list(ex.map(lambda b: oracle.itd_lean(_batch_signal(b, n), m), ids))
    for j, (b, ref) in enumerate(zip(ids, refs)):
        assert int(table["n_rows"][j]) == ref["rows"].shape[0], "signal %d" % b
        assert ("natural", "timeout")[int(table["stop"][j])] == ref["stop"]
        assert table["knot_counts"][j, : len(ref["knot_counts"])].tolist() == ref["knot_counts"].tolist(), "signal %d" % b
    for (b, xh, got) in row_checks:
        ref = refs[ids.index(b)]
        assert_bits_equal(got[: ref["rows"].shape[0]], ref["rows"], "signal %d rows" % b)
    eng.close()


def test_fuzz_slice_single_signals(P, oracle):
    """300 fixed-seed cases of tools/fuzz_parity.py: 8 signal families, sizes around the tile boundaries, 0-11 levels."""
    rng = np.random.default_rng(12345)
    bad = []
    for c in range(300):
        kind = int(rng.integers(0, 8))
        n = int(rng.choice([3, 4, 5, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 4097, int(rng.integers(3, 70000)),
                            int(rng.integers(3, 200000))]))
        m = int(rng.integers(0, 12))
        dtype = np.float32 if rng.random() < 0.5 and kind != 7 else np.float64
        x = fuzz_signal(rng, kind, n).astype(dtype)
        if not np.all(np.isfinite(x)):
            continue
        ref = oracle.itd(x, m)
        dec = P.ITD()
        rows = dec.itd(x, max_iteration=m)
        ok = rows.shape == ref["rows"].shape and np.array_equal(canon_u64(rows), canon_u64(ref["rows"])) and \
            dec.stop_reason == ref["stop"] and \
            np.array_equal(canon_u64(dec.get_baselines()), canon_u64(ref["baselines"])) and \
            [int(v) for v in dec.knot_counts[1:] if v >= 0] == ref["knot_counts"].tolist()   # the counts ITD.py:403 prints
        if not ok:
            bad.append((c, kind, n, m, dtype.__name__))
    assert not bad, "mismatching cases (case, kind, n, m, dtype): %s" % bad[:10]


def test_fuzz_slice_batches(P, oracle):
    """50 fixed-seed random batches (1-39 signals, mixed stop behaviour, with and without the baselines buffer)."""
    rng = np.random.default_rng(54321)
    bad = []
    for c in range(50):
        B = int(rng.integers(1, 40))
        n = int(rng.choice([3, 17, 511, 512, 513, 4095, 4096, 4097, int(rng.integers(3, 40000))]))
        m = int(rng.integers(0, 9))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        x = np.stack([fuzz_signal(rng, int(rng.integers(0, 7)), n) for _ in range(B)]).astype(dtype)
        if not np.all(np.isfinite(x)):
            continue
        keep = bool(rng.random() < 0.3)
        out = P.itd_batch(x, m, keep_baselines=keep)
        for b in range(B):
            ref = oracle.itd(x[b], m)
            nr = int(out["n_rows"][b])
            ok = nr == ref["rows"].shape[0] and ("natural", "timeout")[int(out["stop"][b])] == ref["stop"] and \
                np.array_equal(canon_u64(out["rows"][b, :nr]), canon_u64(ref["rows"]))
            if ok and keep:
                nb = int(out["n_baselines"][b])
                ok = nb == ref["baselines"].shape[0] and np.array_equal(canon_u64(out["baselines"][b, :nb]), canon_u64(ref["baselines"]))
            if not ok:
                bad.append((c, B, n, m, dtype.__name__, b))
    assert not bad, "mismatching signals (case, B, n, m, dtype, signal): %s" % bad[:10]


def _decompose(P, torch, x_np, m, chunk=None, keep=False):
    B, n = x_np.shape
    xd = torch.from_numpy(x_np).cuda()
    rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bases = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda") if keep else None
    eng = P.Engine(n, B, 0)
    if chunk is not None:
        eng.set_batch_chunk(chunk)
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), x_np.dtype, n, B, n, m, rows.data_ptr(), bases.data_ptr() if keep else None, None)
    s = eng.summary(B)
    out = rows.cpu().numpy(), (bases.cpu().numpy() if keep else None), s
    eng.close()
    return out


def test_batch_with_leading_silence_follows_the_nan_rules(P, torch, oracle):
    """Half of a 64-signal batch starts with digital silence (a leading plateau: the first baseline goes NaN and the
    reference's stop test takes detect_peaks' NaN branch, ITD.py:46-51,64-68,400-404).  The extraction kernel follows those
    rules itself: same launch sequence as any other batch, bit-exact rows, baselines and knot counts."""
    n, B, m = 1 << 16, 64, 9
    rng = np.random.default_rng(5)
    x = np.stack([sines_noise(n, seed=b, fscale=1 + b / 64.0, dtype=np.float64) for b in range(B)])
    for b in range(0, B, 2):
        x[b, : int(rng.integers(2, 3000))] = 0.0          # silence at the head, lengths across tile boundaries
    x[2, :] = np.round(x[2] * 8) / 8                       # quantised as well: plateaus everywhere
    x[4, -700:] = 0.25                                     # and a trailing plateau
    for keep in (False, True):
        rows, bases, s = _decompose(P, torch, x, m, keep=keep)
        for b in range(B):
            ref = oracle.itd(x[b], m)
            nr = int(s["n_rows"][b])
            assert nr == ref["rows"].shape[0], "signal %d" % b
            assert int(s["nan_levels"][b]) == -1
            assert s["knot_counts"][b, 1:nr + 1].tolist()[: len(ref["knot_counts"])] == ref["knot_counts"].tolist()[:nr], "signal %d" % b
            assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d rows" % b)
            if keep:
                nb = int(s["n_baselines"][b])
                assert nb == ref["baselines"].shape[0]
                assert_bits_equal(bases[b, :nb], ref["baselines"], "signal %d baselines" % b)


def test_results_do_not_depend_on_the_batch_chunk(P, torch, oracle):
    n, B, m = 9000, 13, 6
    rng = np.random.default_rng(8)
    x = np.stack([fuzz_signal(rng, b % 7, n) for b in range(B)])
    base_rows, _, base_s = _decompose(P, torch, x, m, chunk=0)
    for chunk in (1, 3, 5, 13, 64):
        rows, _, s = _decompose(P, torch, x, m, chunk=chunk)
        assert s["n_rows"].tolist() == base_s["n_rows"].tolist() and s["stop"].tolist() == base_s["stop"].tolist()
        assert s["knot_counts"].tolist() == base_s["knot_counts"].tolist()
        for b in range(B):
            nr = int(s["n_rows"][b])
            assert_bits_equal(rows[b, :nr], base_rows[b, :nr], "chunk %d signal %d" % (chunk, b))
    for b in range(B):
        ref = oracle.itd(x[b], m)
        assert_bits_equal(base_rows[b, : int(base_s["n_rows"][b])], ref["rows"], "signal %d" % b)


def test_helpers_do_not_disturb_a_decomposition_in_flight(P, torch, oracle):
    """itd_detect_* / itd_baseline_extract_* between itd_decompose_* and itd_get_summary (what the header suggests for
    inspecting per-level knots) run in a workspace of their own."""
    n, m = 1 << 18, 7
    x = sines_noise(n, seed=9)
    y = sines_noise(n // 2, seed=10, dtype=np.float64)
    xd = torch.from_numpy(x).cuda()
    rows = torch.empty((m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, 1, 0)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, stream.cuda_stream)
    k = eng.detect_host(y)                                  # the engine's own stream: may overlap the decomposition
    rot, base = eng.baseline_extract_host(y)
    s = eng.summary(1)
    ref = oracle.itd_lean(x, m)
    nr = int(s["n_rows"][0])
    assert nr == ref["rows"].shape[0] and int(s["stop"][0]) == 1
    assert s["knot_counts"][0, :nr].tolist() == ref["knot_counts"].tolist()
    assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "rows after interleaved helpers")
    np.testing.assert_array_equal(k, oracle.knots(y))
    r2, b2 = oracle.itd_baseline_extract(y)
    assert_bits_equal(rot, r2, "helper rotation")
    assert_bits_equal(base, b2, "helper baseline")
    eng.close()


def test_single_level_helpers_follow_the_reference_on_nan_input(P, oracle):
    """detect_peaks / matlab_detect_peaks / detect_knots / itd_baseline_extract on NaN input: the reference's NaN branch
    (ITD.py:46-51, 64-68; numba_accelerated_itd.py:28-49), pinned by its own outputs (helpers_nan_input.npz) and, on a longer
    signal with NaNs on tile boundaries, by the oracle."""
    from helpers import load_golden
    g = load_golden("helpers_nan_input")
    for c in range(int(g["cases"])):
        x = g["x_%d" % c]
        np.testing.assert_array_equal(P.detect_peaks(x.copy()), g["valleys_%d" % c])
        np.testing.assert_array_equal(P.matlab_detect_peaks(x.copy()), g["matlab_%d" % c])
        rot, base = P.itd_baseline_extract(x.copy())
        assert_bits_equal(rot, g["rot_%d" % c], "rotation %d" % c)
        assert_bits_equal(base, g["base_%d" % c], "baseline %d" % c)
    rng = np.random.default_rng(8)
    n = 40000
    x = np.sin(np.arange(n) / 9.0) + 0.3 * rng.standard_normal(n)
    x[[0, 511, 512, 513, 1024, 20000, 20001, n - 1]] = np.nan
    keep = x.copy()
    np.testing.assert_array_equal(P.detect_peaks(x), oracle.detect_peaks(x))
    np.testing.assert_array_equal(P.matlab_detect_peaks(x), oracle.detect_peaks(x, matlab=True))
    np.testing.assert_array_equal(P.detect_knots(x), oracle.knots(x))
    rot, base = P.itd_baseline_extract(x)
    r2, b2 = oracle.itd_baseline_extract(x)
    assert_bits_equal(rot, r2, "rotation")
    assert_bits_equal(base, b2, "baseline")
    assert np.array_equal(np.isnan(x), np.isnan(keep)), "the caller's array must not be written"
    assert P.ITD().itd(x, 3).shape[0] >= 1


def test_stopped_signals_in_a_batch(P, torch, oracle):
    """Signals that stop at different levels share launches with signals that run to the timeout: stopped ones leave
    every later launch at once (before any load), the others are unaffected."""
    n, m = 50000, 10
    t = np.linspace(0, 1, n)
    x = np.stack([np.sin(2 * np.pi * 3 * t), sines_noise(n, seed=1, dtype=np.float64), t ** 2, np.sin(2 * np.pi * 40 * t) + t,
                  sines_noise(n, seed=2, dtype=np.float64), np.cos(2 * np.pi * 1.5 * t)])
    rows, bases, s = _decompose(P, torch, x, m, keep=True)
    stops = set()
    for b in range(x.shape[0]):
        ref = oracle.itd(x[b], m)
        nr, nb = int(s["n_rows"][b]), int(s["n_baselines"][b])
        stops.add((ref["stop"], nr))
        assert nr == ref["rows"].shape[0] and nb == ref["baselines"].shape[0]
        assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d" % b)
        assert_bits_equal(bases[b, :nb], ref["baselines"], "signal %d baselines" % b)
    assert len(stops) >= 3      # the batch really mixes stop levels


def _rank_main(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(rank)
        from pyitd_amd.distributed import ShardedBatch
        batch, n, m = 7, 1 << 16, 5
        sb = ShardedBatch(batch, n, m, world, rank, device=rank)
        # the batch lives on the LAST rank's GPU and is scattered from there (grouped ncclSend / ncclRecv under RCCL: the north
        # star's "trivial batch scatter"); with one rank the root keeps its own range
        root = world - 1
        x_root = torch.from_numpy(np.stack([_batch_signal(b, n) for b in range(batch)])).cuda() if rank == root else None
        x = sb.scatter_from(root, x_root, out=torch.full((sb.n_local, n), -1.0, dtype=torch.float32, device="cuda"))
        assert bool((x[0] == torch.from_numpy(_batch_signal(sb.lo, n)).cuda()).all())
        rows = torch.empty((sb.n_local, m + 2, n), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        sb.decompose(x.data_ptr(), np.float32, n, rows.data_ptr())
        # what bench.py's ranks do around the timed region, on RCCL: barrier, the all-gather of the per-rank times, the table
        dist.barrier()
        tt = torch.tensor([1.0 + rank], dtype=torch.float64, device=torch.device("cuda", rank))
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt)
        assert [float(p.item()) for p in parts] == [1.0 + r for r in range(world)]
        table = sb.gather(device=torch.device("cuda", rank))
        if rank == 0:
            q.put({k: v.tolist() for k, v in table.items()})
    finally:
        dist.destroy_process_group()


def test_two_gpu_ranks_shard_and_gather_over_rccl(torch, oracle):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU node); the 1-GPU box runs the emulation above")
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for b in range(7):
        ref = oracle.itd_lean(_batch_signal(b, 1 << 16), 5)
        assert got["n_rows"][b] == ref["rows"].shape[0]
        assert got["knot_counts"][b][: len(ref["knot_counts"])] == ref["knot_counts"].tolist()


def test_one_rank_over_rccl(torch, oracle):
    """The RCCL leg of the sharded path on the one-GPU box: a process group of ONE rank with backend "nccl" (= RCCL) runs the same
    rank function as the two-GPU test — engine on its shard, barrier, all-gather of the times and of the summary table on device
    tensors.  (Two ranks cannot share a GPU under RCCL; the two-rank logic runs over gloo in tests/test_distributed_cpu.py and in the
    --rehearse-one-gpu run of bench.py.)"""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rank_main, args=(0, 1, port, q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    for b in range(7):
        ref = oracle.itd_lean(_batch_signal(b, 1 << 16), 5)
        assert got["n_rows"][b] == ref["rows"].shape[0]
        assert got["knot_counts"][b][: len(ref["knot_counts"])] == ref["knot_counts"].tolist()


def _modes_agree(P, torch, oracle, x, m, expect_repeat=None):
    """The three level-0 modes give bit-identical results; returns the rows of the automatic mode."""
    from pyitd_amd.engine import LEVEL0_AUTO, LEVEL0_RECORDS
    x = np.ascontiguousarray(x)
    n = x.shape[0]
    ref = oracle.itd(x, m)
    out = {}
    for mode in (LEVEL0_AUTO, LEVEL0_RECORDS):
        eng = P.Engine(n, 1, 0)
        eng.set_level0_mode(mode)
        xd = torch.from_numpy(x).cuda()
        rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        bases = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), bases.data_ptr(), None)
        s = eng.summary(1)
        nr, nb = int(s["n_rows"][0]), int(s["n_baselines"][0])
        assert nr == ref["rows"].shape[0] and nb == ref["baselines"].shape[0], "mode %d" % mode
        assert [int(v) for v in s["knot_counts"][0, 1:] if v >= 0] == ref["knot_counts"].tolist(), "mode %d" % mode
        assert int(s["knot_counts"][0, 0]) == len(oracle.knots(np.asarray(x, dtype=np.float64))), "mode %d: knots of the input" % mode
        assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "mode %d rows" % mode)
        assert_bits_equal(bases[:nb].cpu().numpy(), ref["baselines"], "mode %d baselines" % mode)
        out[mode] = rows[:nr].cpu().numpy()
        eng.close()
    return out[LEVEL0_AUTO]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_fused_level0_on_smooth_signals(P, torch, oracle, dtype):
    """The fused level-0 launch takes a tile's neighbouring knots from the 128 samples either side; smoother stretches
    make it walk on through the signal (overlapping 512-sample windows).  Knot spacings from a few samples to ~3000."""
    n = 200003
    t = np.arange(n) / n
    rng = np.random.default_rng(17)
    cases = {
        "extrema every ~60 samples": np.sin(2 * np.pi * 1700 * t),
        "every ~300": np.sin(2 * np.pi * 333 * t) + 0.3 * t,
        "every ~1500": np.sin(2 * np.pi * 67 * t) * (1 + t),
        "every ~3000": np.sin(2 * np.pi * 33.3 * t + 0.4),
        "chirp from dense to sparse": np.sin(2 * np.pi * (3000 * t - 2950 * t * t)),
        "noise burst in a slow wave": np.sin(2 * np.pi * 40 * t) + np.where((t > 0.4) & (t < 0.41), 0.01 * rng.standard_normal(n), 0.0),
        "plateaus (quantised slow wave)": np.round(np.sin(2 * np.pi * 90 * t) * 20) / 20,
    }
    for name, x in cases.items():
        _modes_agree(P, torch, oracle, x.astype(dtype), 6)


def test_fused_level0_beyond_its_reach_repeats_record_driven(P, torch, oracle):
    """Knots further apart than the fused launch reaches (> ~4000 samples): itd_get_summary repeats the call through
    k_scan0 + records; the engine's following decompositions start record-driven; a forced-fused engine reports it."""
    from pyitd_amd.engine import LEVEL0_FUSED
    n = 1 << 18
    t = np.arange(n) / n
    x = np.sin(2 * np.pi * 9 * t) + 0.2 * t * t          # an extremum every ~14 500 samples
    _modes_agree(P, torch, oracle, x, 5)
    mono = np.linspace(-1, 2, n) ** 3                      # no knots at all: the whole signal is one segment
    _modes_agree(P, torch, oracle, mono, 3)
    eng = P.Engine(n, 1, 0)
    eng.set_level0_mode(LEVEL0_FUSED)
    xd = torch.from_numpy(x).cuda()
    rows = torch.empty((7, n), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), np.float64, n, 1, n, 5, rows.data_ptr(), None, None)
    with pytest.raises(P.ITDError):
        eng.summary(1)
    eng.close()


def test_fused_level0_float32_subnormals_and_huge_values(P, torch, oracle):
    """The fused launch evaluates the predicate in float32 for float32 input: differences of subnormals must not be
    flushed, and +-inf / 1e38 magnitudes behave like the reference's float64 differences of the widened values."""
    rng = np.random.default_rng(23)
    n = 70000
    tiny = (rng.standard_normal(n) * 1e-42).astype(np.float32)          # float32 subnormals (spacing 1.4e-45)
    assert np.count_nonzero(tiny) > n // 2 and np.abs(tiny).max() < 1.2e-38
    _modes_agree(P, torch, oracle, tiny, 5)
    huge = (rng.standard_normal(n) * 1e38).astype(np.float32)
    huge[np.isinf(huge)] = np.float32(3e38)
    _modes_agree(P, torch, oracle, huge, 4)
    mixed = rng.standard_normal(n).astype(np.float32)
    mixed[::7] = mixed[1::7][: len(mixed[::7])]                          # equal neighbours: plateaus of two
    _modes_agree(P, torch, oracle, mixed, 6)


def test_bench_gpus_2_rehearsal_on_one_gpu(torch):
    """`python bench.py --gpus 2` as the driver's plain form: the parent spawns two ranks which shard config 4's batch recipe,
    decompose their shard on the GPU and all-gather the summaries.  On a one-GPU box both ranks share cuda:0 and the collective
    runs over gloo (--rehearse-one-gpu); everything else is the code path of the multi-GPU run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-one-gpu", "--batch", "6",
                        "--log2n", "16", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["signals_per_gpu"] == 6 and d["config"]["samples_per_signal"] == 1 << 16
    assert d["config"]["signals_in_gathered_table"] == 12 and d["config"]["rows_all_ranks"] == [9]
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and "rehearsal" in d["config"]


def test_batch_larger_than_one_grid(P, torch, oracle):
    """More signals than gridDim.y allows (65535): the batch runs in chunks, nothing about it is special for the caller."""
    B, n, m = 70000, 200, 3
    rng = np.random.default_rng(77)
    x_host = np.cumsum(rng.standard_normal((B, n)), axis=1).astype(np.float32)
    x = torch.from_numpy(x_host).cuda()
    rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, B, 0)
    torch.cuda.synchronize()
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, None)
    s = eng.summary(B)
    check = sorted(set(list(range(0, 40)) + list(range(65500, 65600)) + list(range(B - 40, B)) + rng.choice(B, 300, replace=False).tolist()))
    got = rows[check].cpu().numpy()
    for k, b in enumerate(check):
        ref = oracle.itd(x_host[b], m)
        nr = int(s["n_rows"][b])
        assert nr == ref["rows"].shape[0] and ("natural", "timeout")[int(s["stop"][b])] == ref["stop"], "signal %d" % b
        assert_bits_equal(got[k, :nr], ref["rows"], "signal %d rows" % b)
    eng.close()


def test_engines_on_concurrent_host_threads(P, torch, oracle):
    """SURVEY 8b threading: an engine is not thread-safe, but several engines — each with its own host thread and stream —
    run concurrently on one GPU.  Four threads decompose different signals in loops (device form and host form mixed)."""
    n, m, iters = 50000, 5, 12
    rng = np.random.default_rng(99)
    sigs = [np.cumsum(rng.standard_normal(n)).astype(np.float32) + (0.3 * rng.standard_normal(n)).astype(np.float32) for _ in range(8)]
    refs = [oracle.itd(x, m) for x in sigs]
    errors = []

    def worker(k):
        try:
            eng = P.Engine(n, 1, 0)
            stream = torch.cuda.Stream()
            rows = torch.empty((m + 2, n), dtype=torch.float64, device="cuda")
            xs = [torch.from_numpy(x).cuda() for x in sigs]
            torch.cuda.synchronize()
            for it in range(iters):
                j = (k + 3 * it) % len(sigs)
                if it % 3 == 2:
                    res = eng.decompose_host(sigs[j], m, want_baselines=True)
                    got, nr = res["rows"], res["rows"].shape[0]
                else:
                    eng.decompose_dev(xs[j].data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, stream.cuda_stream)
                    nr = int(eng.summary(1)["n_rows"][0])
                    got = rows[:nr].cpu().numpy()
                if nr != refs[j]["rows"].shape[0] or not np.array_equal(canon_u64(got), canon_u64(refs[j]["rows"])):
                    errors.append((k, it, j))
            eng.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((k, repr(ex)))

    with ThreadPoolExecutor(4) as ex:
        list(ex.map(worker, range(4)))
    assert not errors, errors[:5]
