"""Device-resident entry points (torch tensors carry the HBM buffers; only raw pointers cross the C ABI):
batches, strides, the ping-pong/keep-all baseline modes, full-size property checks."""
import os

import numpy as np
import pytest

from helpers import assert_bits_equal, sines_noise

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
    return cpu_oracle


def _run(P, torch, x_np, m, keep_baselines, stride_pad=0):
    B, n = x_np.shape
    stride = n + stride_pad
    dt = torch.float32 if x_np.dtype == np.float32 else torch.float64
    xd = torch.zeros((B, stride), dtype=dt, device="cuda")
    xd[:, :n] = torch.from_numpy(x_np).cuda()
    R = m + 2
    rows = torch.full((B, R, n), float("nan"), dtype=torch.float64, device="cuda")
    bases = torch.full((B, R, n), float("nan"), dtype=torch.float64, device="cuda") if keep_baselines else None
    eng = P.Engine(n, B, 0)
    torch.cuda.synchronize()   # the engine's stream does not wait for torch's null stream
    stream = torch.cuda.current_stream().cuda_stream
    eng.decompose_dev(xd.data_ptr(), x_np.dtype, n, B, stride, m, rows.data_ptr(),
                      bases.data_ptr() if keep_baselines else None, stream)
    s = eng.summary(B)
    torch.cuda.synchronize()
    out = rows.cpu().numpy(), (bases.cpu().numpy() if keep_baselines else None), s
    eng.close()
    return out


@pytest.mark.parametrize("keep", [False, True])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_batch_matches_oracle_per_signal(P, torch, oracle, keep, dtype):
    n, B, m = 6000, 9, 6
    rng = np.random.default_rng(3)
    x = np.stack([sines_noise(n, seed=b, fscale=1 + b / 8.0, dtype=np.float64) for b in range(B)])
    x[3] = np.linspace(0, 1, n) ** 2                  # monotone: one all-zero row (ITD.py:404-416 with counter 0)
    x[4] = np.sin(np.linspace(0, 3 * np.pi, n))       # stops naturally after a couple of levels
    x[5] = rng.standard_normal(n)
    x = x.astype(dtype)
    rows, bases, s = _run(P, torch, x, m, keep, stride_pad=4)
    for b in range(B):
        ref = oracle.itd(x[b], m)
        nr = int(s["n_rows"][b])
        assert nr == ref["rows"].shape[0], "signal %d" % b
        assert ("natural", "timeout")[int(s["stop"][b])] == ref["stop"]
        assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d rows" % b)
        assert int(s["nan_levels"][b]) == -1
        if keep:
            nb = int(s["n_baselines"][b])
            assert nb == ref["baselines"].shape[0]
            assert_bits_equal(bases[b, :nb], ref["baselines"], "signal %d baselines" % b)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_unaligned_rows_and_ragged_tiles(P, torch, oracle, dtype):
    # odd element stride: signals 1.. start at addresses that are not 16-byte aligned, and N is not a multiple of the tile —
    # the bounds-checked buffer accesses of k_extract / k_scan0 take these through the same path as full aligned tiles
    n, B, m = 4099, 3, 3
    x = np.stack([sines_noise(n, seed=10 + b, dtype=np.float64) for b in range(B)]).astype(dtype)
    rows, _, s = _run(P, torch, x, m, False, stride_pad=1)
    for b in range(B):
        ref = oracle.itd(x[b], m)
        assert_bits_equal(rows[b, : int(s["n_rows"][b])], ref["rows"], "signal %d" % b)


def test_config2_full_size_properties(P, torch, oracle):
    """BASELINE configs[1]: 2^24 float32 sines+noise, 8 levels (max_iteration=7), one GPU.
    Bit-exact vs the oracle on knot counts and on a checksum of every row; exact reconstruction."""
    n, m = 1 << 24, 7
    x = sines_noise(n)
    rows, _, s = _run(P, torch, x[None], m, False)
    nr = int(s["n_rows"][0])
    ref = oracle.itd_lean(x, m)
    assert nr == ref["rows"].shape[0] == 9 and int(s["stop"][0]) == 1
    assert s["knot_counts"][0, :nr].tolist() == ref["knot_counts"].tolist()
    assert_bits_equal(rows[0, :nr], ref["rows"], "2^24 rows")
    # size-independent properties: rows sum back to the input; the last sample is never moved (ITD.py:112-117)
    recon = rows[0, :nr].sum(axis=0)
    assert np.max(np.abs(recon - x.astype(np.float64))) < 1e-12
    assert rows[0, 0, -1] == float(x[-1]) and np.all(rows[0, 1:nr, -1] == 0.0)


def test_batch_with_nan_path_signals(P, torch, oracle):
    """A batch where some signals take the NaN-faithful re-run and others the fast path."""
    n, B, m = 5000, 6, 5
    rng = np.random.default_rng(77)
    x = rng.standard_normal((B, n))
    x[1, :4] = 1.0          # leading plateau -> NaN path
    x[4, :130] = -2.0       # a long one, crossing a 64-sample group
    for keep in (False, True):
        rows, bases, s = _run(P, torch, x, m, keep)
        for b in range(B):
            ref = oracle.itd(x[b], m)
            nr = int(s["n_rows"][b])
            assert nr == ref["rows"].shape[0]
            assert int(s["nan_levels"][b]) == -1
            assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d rows (keep=%s)" % (b, keep))
            if keep:
                nb = int(s["n_baselines"][b])
                assert_bits_equal(bases[b, :nb], ref["baselines"], "signal %d baselines" % b)


def test_config3_shape_small_batch(P, torch, oracle):
    """BASELINE configs[2] shape at a size the oracle finishes in seconds: many 2^16 signals, 8 levels."""
    n, B, m = 1 << 16, 24, 7
    x = np.stack([sines_noise(n, seed=b % 16, fscale=1 + b / 8192.0) for b in range(B)])
    rows, _, s = _run(P, torch, x, m, False)
    for b in (0, 5, 17, 23):
        ref = oracle.itd_lean(x[b], m)
        nr = int(s["n_rows"][b])
        assert nr == ref["rows"].shape[0]
        assert s["knot_counts"][b, :nr].tolist() == ref["knot_counts"].tolist()
        assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d" % b)


def test_decompose_is_graph_capturable(P, torch, oracle):
    """The decompose entry points allocate nothing and never synchronise: a whole decomposition can be captured in
    a hipGraph (here through torch's capture API) and replayed on new data."""
    n, m = 50000, 5
    x_np = sines_noise(n, seed=3)
    x = torch.from_numpy(x_np).cuda()
    rows = torch.zeros((m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, 1, 0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, side.cuda_stream)  # warm-up
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None,
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for seed in (4, 5):
        y_np = sines_noise(n, seed=seed)
        x.copy_(torch.from_numpy(y_np))
        rows.zero_()
        g.replay()
        torch.cuda.synchronize()
        ref = oracle.itd(y_np, m)
        assert_bits_equal(rows[: ref["rows"].shape[0]].cpu().numpy(), ref["rows"], "graph replay seed %d" % seed)
    eng.close()


def test_itd_batch_python_api(P, torch, oracle):
    rng = np.random.default_rng(21)
    x = rng.standard_normal((5, 3000)).astype(np.float32)
    x[2] = np.linspace(0, 1, 3000, dtype=np.float32) ** 2          # monotone: a single all-zero row
    for src in (x, torch.from_numpy(x).cuda()):
        out = P.itd_batch(src, max_iteration=4, keep_baselines=True)
        rows = out["rows"].cpu().numpy() if hasattr(out["rows"], "cpu") else out["rows"]
        bases = out["baselines"].cpu().numpy() if hasattr(out["baselines"], "cpu") else out["baselines"]
        for b in range(5):
            ref = oracle.itd(x[b], 4)
            nr, nb = int(out["n_rows"][b]), int(out["n_baselines"][b])
            assert nr == ref["rows"].shape[0] and nb == ref["baselines"].shape[0]
            assert_bits_equal(rows[b, :nr], ref["rows"], "batch api rows %d" % b)
            assert_bits_equal(bases[b, :nb], ref["baselines"], "batch api baselines %d" % b)
    with pytest.raises(ValueError):
        P.itd_batch(np.zeros((2, 2)))
    bad = x.copy()
    bad[1, 7] = np.nan                      # NaN in one signal of the batch: that signal follows the reference's NaN branch
    out = P.itd_batch(bad, 3)
    for b in range(5):
        ref = oracle.itd(bad[b], 3)
        assert_bits_equal(out["rows"][b, : int(out["n_rows"][b])], ref["rows"], "batch with a NaN, signal %d" % b)


def test_kernel_timing_api(P, torch, oracle):
    # itd_set_kernel_timing / _stride / itd_get_kernel_timing: instrumented decompositions launch the extraction kernels with
    # their own events; results must not change and the tallies must count exactly the instrumented launches
    from pyitd_amd.engine import TIME_DECOMPOSE, TIME_EXTRACT, TIME_EXTRACT_FINAL, TIME_EXTRACT_L0
    n, m = 1 << 16, 5
    x = sines_noise(n, seed=2, dtype=np.float32)
    ref = oracle.itd(x, m)
    xd = torch.from_numpy(x).cuda()
    rows = torch.empty((m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, 1, 0)
    from pyitd_amd.engine import FUSE_OFF
    eng.set_fuse_mode(FUSE_OFF)       # the launch classes counted here are the level-by-level engine's (the fused levels: below)
    torch.cuda.synchronize()
    steps, stride = 6, 2
    eng.set_timing(steps, stride=stride)
    for _ in range(steps):
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
    s = eng.summary(1)
    timed = steps // stride
    ms, cnt = eng.kernel_timing(TIME_EXTRACT)
    assert cnt == timed * m and ms > 0.0                      # levels 1..m
    assert eng.kernel_timing(TIME_EXTRACT_L0)[1] == timed
    assert eng.kernel_timing(TIME_EXTRACT_FINAL)[1] == timed
    span_ms, span_cnt = eng.kernel_timing(TIME_DECOMPOSE)
    assert span_cnt == timed and span_ms >= ms                # the span contains the launches it brackets
    eng.set_timing(0)
    nr = int(s["n_rows"][0])
    assert nr == ref["rows"].shape[0]
    assert_bits_equal(rows[:nr].cpu().numpy(), ref["rows"], "rows with instrumented launches")
    eng.close()
    # with the sparse levels fused: levels 0 .. 2 as launches, one instrumented sample pass, a span over the knot side
    from pyitd_amd.engine import FUSE_ONLY, TIME_KF_APPLY, TIME_KF_KNOTS
    eng = P.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_ONLY)
    eng.set_fuse_level(3)             # (whatever PYITD_FUSE_LEVEL says: the counts below are level 3's)
    eng.set_fuse_cap(-1)              # (... and PYITD_FUSE_CAP: every level from 3 on fused)
    eng.set_timing(steps, stride=stride)
    for _ in range(steps):
        eng.decompose_dev(xd.data_ptr(), np.float32, n, 1, n, m, rows.data_ptr(), None, None)
    s = eng.summary(1)
    assert eng.kernel_timing(TIME_EXTRACT)[1] == timed * 2 and eng.kernel_timing(TIME_EXTRACT_FINAL)[1] == 0
    ms_a, cnt_a = eng.kernel_timing(TIME_KF_APPLY)
    ms_k, cnt_k = eng.kernel_timing(TIME_KF_KNOTS)
    assert cnt_a == timed and cnt_k == timed and ms_a > 0.0 and ms_k > 0.0
    eng.set_timing(0)
    assert_bits_equal(rows[: int(s["n_rows"][0])].cpu().numpy(), ref["rows"], "rows of the fused form with instrumented launches")
    eng.close()


def test_decomposition_is_graph_capturable(P, torch, oracle):
    """include/pyitd_hip.h: the decompose calls allocate nothing and synchronise nothing, so a caller can capture them in a HIP
    graph; replays on new data in the same buffers give that data's decomposition."""
    n, M = 1 << 18, 5
    eng = P.Engine(n, 1, 0)
    x = torch.zeros(n, dtype=torch.float32, device="cuda")
    rows = torch.zeros((M + 2, n), dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    x.copy_(torch.from_numpy(sines_noise(n, seed=1)))
    torch.cuda.synchronize()
    with torch.cuda.stream(s):       # once outside the capture (workspaces that are allocated at first use)
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, s.cuda_stream)
    eng.summary(1)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    for seed in (2, 3):
        xh = sines_noise(n, seed=seed)
        x.copy_(torch.from_numpy(xh))
        rows.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert_bits_equal(rows.cpu().numpy(), oracle.itd_lean(xh, M)["rows"], "graph replay, seed %d" % seed)
    eng.close()


def test_batch_streams_and_chunks_do_not_change_results(P, torch, oracle):
    """The chunks of a batch rotate over 1 .. 4 streams (itd_set_batch_streams) in any chunk size: rows, baselines and summaries
    are the same, with signals that stop at different levels (and one that holds a NaN) spread over the chunks."""
    rng = np.random.default_rng(91)
    B, n, M = 23, 6000, 6
    xs = np.stack([sines_noise(n, seed=b, fscale=1 + b / 16) for b in range(B)]).astype(np.float64)
    xs[4] = np.linspace(0, 1, n)                      # stops at once
    xs[9] = np.sin(np.arange(n) / 900.0)              # stops after a few levels
    xs[17, [0, 3000, 3001]] = np.nan                  # the NaN-input repeat runs over the same streams
    refs = [oracle.itd(xs[b], M) for b in range(B)]
    x = torch.from_numpy(xs).cuda()
    eng = P.Engine(n, B, 0)
    for streams, chunk in ((1, 0), (2, 1), (2, 5), (3, 4), (4, 2), (4, 23)):
        eng.set_batch_streams(streams)
        eng.set_batch_chunk(chunk)
        rows = torch.full((B, M + 2, n), -7.0, dtype=torch.float64, device="cuda")
        bases = torch.full((B, M + 2, n), -7.0, dtype=torch.float64, device="cuda")
        eng.decompose_dev(x.data_ptr(), np.float64, n, B, n, M, rows.data_ptr(), bases.data_ptr(), None)
        s = eng.summary(B)
        for b in range(B):
            nr, nb = int(s["n_rows"][b]), int(s["n_baselines"][b])
            what = "streams %d chunk %d signal %d" % (streams, chunk, b)
            assert nr == refs[b]["rows"].shape[0] and nb == refs[b]["baselines"].shape[0], what
            assert_bits_equal(rows[b, :nr].cpu().numpy(), refs[b]["rows"], what + " rows")
            assert_bits_equal(bases[b, :nb].cpu().numpy(), refs[b]["baselines"], what + " baselines")
    eng.close()


def test_state_sets_survive_changing_geometry(P, torch, oracle):
    """The engine keeps two sets of per-signal states / group sums; a call's last launch re-initialises the set the NEXT call will
    use (no initialising launch per call).  Results must not depend on what ran before: batches and lengths that grow and shrink
    on one engine, calls whose summary is never read, signals that stop early or hold a NaN, a graph replayed twice and plain
    calls after it."""
    rng = np.random.default_rng(4711)
    Bmax, nmax, M = 12, 70000, 6
    eng = P.Engine(nmax, Bmax, 0)

    def make(B, n, special):
        xs = np.stack([sines_noise(n, seed=int(rng.integers(0, 1 << 30)), fscale=1 + b / 8) for b in range(B)]).astype(np.float64)
        if special and B > 2:
            xs[1] = np.linspace(-1, 1, n)            # stops at once
            xs[2, n // 2] = np.nan                   # NaN-input repeat (runs on the same set again)
        return xs

    def check(xs, read_summary=True):
        B, n = xs.shape
        x = torch.from_numpy(xs).cuda()
        rows = torch.full((B, M + 2, n), -3.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()     # (the fill runs on torch's stream, the engine on its own: unordered, the fill overwrote rows now and then)
        eng.decompose_dev(x.data_ptr(), np.float64, n, B, n, M, rows.data_ptr(), None, None)
        if not read_summary:
            torch.cuda.synchronize()
            return
        s = eng.summary(B)
        for b in range(B):
            ref = oracle.itd(xs[b], M)
            nr = int(s["n_rows"][b])
            assert nr == ref["rows"].shape[0], "B %d n %d signal %d" % (B, n, b)
            assert_bits_equal(rows[b, :nr].cpu().numpy(), ref["rows"], "B %d n %d signal %d" % (B, n, b))
            kc = [int(v) for v in s["knot_counts"][b] if v >= 0]     # [0]: the signal's own knots; [j >= 1]: the stop test's counts
            want = [int(v) for v in ref["knot_counts"]]
            assert kc[1: 1 + len(want)] == want[: len(kc) - 1], "B %d n %d signal %d" % (B, n, b)

    for B, n, special, read in ((12, 70000, True, True), (1, 513, False, True), (3, 20000, True, False), (12, 9000, False, False),
                                (2, 70000, False, True), (12, 1025, True, True), (5, 40000, True, True), (1, 3, False, True),
                                (12, 70000, False, True)):
        check(make(B, n, special), read)

    # a captured call is complete in itself: replayed twice, then plain calls of another geometry
    n = 1 << 15
    x = torch.zeros((2, n), dtype=torch.float64, device="cuda")
    rows = torch.zeros((2, M + 2, n), dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.decompose_dev(x.data_ptr(), np.float64, n, 2, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    for seed in (5, 6):
        xh = np.stack([sines_noise(n, seed=seed), sines_noise(n, seed=seed + 10, fscale=3.0)]).astype(np.float64)
        x.copy_(torch.from_numpy(xh))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for b in range(2):
            assert_bits_equal(rows[b].cpu().numpy(), oracle.itd_lean(xh[b], M)["rows"], "replay seed %d signal %d" % (seed, b))
    check(make(7, 30000, True))
    check(make(12, 70000, False))
    eng.close()


def test_knot_values_on_device_buffers(P, torch, oracle):
    """itd_knot_values_f64: baseline_knot_estimation (numba_accelerated_itd.py:167-178) on device buffers — the knots as
    itd_detect_* delivers them (int32, on the device), asynchronous on the caller's stream; bk[0] and bk[m+1] stay the caller's."""
    from pyitd_amd import engine as E
    n = 50000
    xh = sines_noise(n, seed=12).astype(np.float64)
    eng = P.Engine(n, 1, 0)
    x = torch.from_numpy(xh).cuda()
    idx = torch.zeros(n + 2, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    import ctypes
    m = ctypes.c_int64(0)
    rc = eng._L.itd_detect_f64(eng._h, x.data_ptr(), n, E.DETECT_KNOTS, idx.data_ptr() + 4, ctypes.byref(m), s.cuda_stream)
    assert rc == 0
    m = int(m.value)
    want_knots = oracle.knots(xh)
    assert m == len(want_knots)
    idx[0] = 0
    idx[m + 1] = n - 1
    bk = torch.full((m + 2,), -5.0, dtype=torch.float64, device="cuda")
    eng.knot_values_dev(x.data_ptr(), n, idx.data_ptr(), m, bk.data_ptr(), s.cuda_stream)
    s.synchronize()
    e = np.concatenate([[0], want_knots, [n - 1]]).astype(np.int64)
    want = oracle.knot_values(xh, e)
    got = bk.cpu().numpy()
    assert got[0] == -5.0 and got[-1] == -5.0
    assert_bits_equal(got[1:-1], want[1:-1], "knot values on device buffers")
    eng.close()


def test_c_abi_shard_scatter_on_one_rank(P, torch):
    """itd_shard_scatter with a world of one is a stream-ordered local copy (no communicator, RCCL never loaded); with more ranks
    and no communicator it refuses instead of guessing.  (More than one rank over RCCL: test_two_gpu_ranks_shard_and_gather_over_rccl,
    where two devices are visible.)"""
    import ctypes
    from pyitd_amd import _lib
    L = _lib.load()
    B, n = 5, 4099
    x = torch.randn((B, n), dtype=torch.float32, device="cuda")
    y = torch.zeros_like(x)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    rc = L.itd_shard_scatter(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), n, B, 4, 1, 0, 0, None, ctypes.c_void_p(s.cuda_stream))
    assert rc == 0
    s.synchronize()
    assert torch.equal(x, y)
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    assert L.itd_shard_range(B, 2, 1, ctypes.byref(lo), ctypes.byref(hi)) == 0 and (lo.value, hi.value) == (3, 5)
    assert L.itd_shard_scatter(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), n, B, 4, 2, 0, 0, None, None) != 0
