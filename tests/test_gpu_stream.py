"""Block-wise (streaming) operation and the batched / multichannel single-level operators on the GPU, through the C ABI
(itd_stream_*, *_batch_f64), against
  * tests/golden/stream/*.npz — the recipe of itd.cpp:31-44 run over the REFERENCE's own operators (oracle/gen_golden.py),
  * the pinned CPU oracle (oracle/stream_oracle.py over oracle/cpu_oracle.py) on seeded signals,
  * the whole-signal operator (tier-1: the stream IS the whole-signal result wherever blocks hold a few knots).
Tier-1 (piecewise-affine, ITD.py:79-121): bit-exact.  Cubic (itd_fourier_decomposition.py:49-122): 1e-9 of the signal's scale
(t**3 by multiplication vs libm pow, the sweeps as scans), knot selection decisions exact."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, assert_bits_equal, fuzz_signal

pytestmark = pytest.mark.gpu
STREAM = os.path.join(GOLDEN, "stream")
TOL = 1e-9


def cases(prefix):
    return sorted(f[:-4] for f in os.listdir(STREAM) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="module")
def S():
    from pyitd_amd import streaming
    return streaming


@pytest.fixture(scope="module")
def so():
    from oracle import stream_oracle
    return stream_oracle


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _close(got, ref, what):
    scale = max(1.0, float(np.nanmax(np.abs(ref))))
    assert got.shape == ref.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what
    err = np.nanmax(np.abs(got - ref)) if got.size else 0.0
    assert err <= TOL * scale, "%s: max |diff| %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("name", cases("cubic_"))
def test_cubic_stream_matches_reference_operator_goldens(S, name):
    g = np.load(os.path.join(STREAM, name + ".npz"))
    got = S.blockwise(g["x"], int(g["block"]), "cubic", int(g["margin"]), bool(g["shared_knots"]))
    _close(got, g["baseline"], name)
    # "too few extrema: the block is emitted unchanged" is a decision, not arithmetic: exact wherever the golden took it
    x = np.atleast_2d(g["x"]); ref = np.atleast_2d(g["baseline"]); got2 = np.atleast_2d(got)
    L = int(g["block"])
    for c in range(x.shape[0]):
        for k in range(x.shape[1] // L):
            sl = slice(k * L, (k + 1) * L)
            if np.array_equal(ref[c, sl], x[c, sl]):
                assert np.array_equal(got2[c, sl], x[c, sl]), (name, c, k)


@pytest.mark.parametrize("name", cases("linear_"))
def test_linear_stream_matches_reference_operator_goldens_bit_for_bit(S, name):
    g = np.load(os.path.join(STREAM, name + ".npz"))
    if np.isnan(g["baseline"]).any():
        # a leading plateau: 0/0 on the first window's end segment (ITD.py:115-116) — followed as the operator computes it
        pass
    rot, base = S.blockwise(g["x"], int(g["block"]), "linear")
    assert_bits_equal(base, g["baseline"], name + " baseline")
    assert_bits_equal(rot, g["rotation"], name + " rotation")


def test_linear_stream_is_the_whole_signal_operator(S, oracle):
    """4096-sample blocks of a 2^18-sample signal: every emitted block bit-identical to the rows of ONE whole-signal
    itd_baseline_extract (ITD.py:79-121) by the oracle — end-knot means at both ends and baseline[n-1] = 0 included."""
    rng = np.random.default_rng(12)
    for kind, n, L in ((0, 1 << 18, 4096), (1, 1 << 16, 1024), (2, 1 << 15, 512), (4, 60000, 20000), (6, 4096, 64)):
        x = fuzz_signal(rng, kind, n)
        rot_w, base_w = oracle.itd_baseline_extract(x)
        rot, base = S.blockwise(x, L, "linear")
        assert_bits_equal(base, base_w, "kind %d baseline" % kind)
        assert_bits_equal(rot, rot_w, "kind %d rotation" % kind)


def test_streams_follow_the_oracle_on_seeded_signals(S, so):
    rng = np.random.default_rng(77)
    for trial in range(12):
        kind = int(rng.integers(0, 7))
        L = int(rng.choice([8, 24, 100, 512, 1000, 4096]))
        nb = int(rng.integers(1, 7))
        C = int(rng.integers(1, 4))
        x = np.stack([fuzz_signal(rng, kind if c == 0 else int(rng.integers(0, 7)), L * nb) for c in range(C)])
        if kind == 5:
            x += 1e-3 * rng.standard_normal(x.shape)     # plateaus give 0/0 in the cubic operator's reference too: keep finite
        margin = int(rng.integers(1, 12))
        shared = bool(rng.integers(0, 2))
        ref = so.oracle_blockwise_cubic(x, L, margin, shared)
        got = S.blockwise(x, L, "cubic", margin, shared)
        if np.isfinite(ref).all():
            _close(got, ref, "cubic trial %d (kind %d L %d nb %d C %d margin %d shared %d)" % (trial, kind, L, nb, C, margin, shared))
        rref, bref = so.oracle_blockwise_linear(x, L)
        rot, base = S.blockwise(x, L, "linear")
        assert_bits_equal(base, bref, "linear trial %d baseline" % trial)
        assert_bits_equal(rot, rref, "linear trial %d rotation" % trial)


def test_sparse_knots_and_short_blocks(S, so):
    """Blocks with no or very few extrema: the cubic recipe emits them unchanged (itd.cpp:170-172), the tier-1 operator runs
    on whatever knots the window holds (its own end knots if none)."""
    t = np.arange(6 * 256)
    for x in (np.linspace(0, 1, t.size) ** 2, np.sin(2 * np.pi * t / 700.0), np.zeros(t.size), np.where(t % 512 < 256, 1.0, -1.0)):
        got = S.blockwise(x, 256, "cubic", 8)
        ref = so.oracle_blockwise_cubic(x, 256, 8)
        if np.isfinite(ref).all():
            _close(got, ref, "sparse cubic")
        rot, base = S.blockwise(x, 256, "linear")
        rref, bref = so.oracle_blockwise_linear(x, 256)
        assert_bits_equal(base, bref, "sparse linear baseline")
        assert_bits_equal(rot, rref, "sparse linear rotation")


def test_device_form_is_asynchronous_and_matches(S, so):
    """itd_stream_push_f64 on device buffers: all pushes enqueued back to back on one stream, results read once at the end."""
    import torch
    rng = np.random.default_rng(3)
    L, nb, C = 4096, 9, 2
    x = np.stack([np.cumsum(rng.standard_normal(L * nb)) * 0.05 + np.sin(np.arange(L * nb) / 40.0), rng.standard_normal(L * nb)])
    xd = torch.from_numpy(x).cuda()
    for kind in ("cubic", "linear"):
        st = S.Stream(L, C, kind, margin=8, shared_knots=False)
        base = torch.zeros_like(xd)
        rot = torch.zeros_like(xd)
        stream = torch.cuda.current_stream().cuda_stream
        n = L * nb
        for k in range(nb):
            blk = xd[:, k * L:]
            out_b = base[:, max(k - 1, 0) * L:]
            out_r = rot[:, max(k - 1, 0) * L:]
            em = st.push_dev(blk.data_ptr(), n, out_b.data_ptr(), n, out_r.data_ptr(), n, stream)
            assert em == (k >= 1)
        assert st.flush_dev(base[:, (nb - 1) * L:].data_ptr(), n, rot[:, (nb - 1) * L:].data_ptr(), n, stream)
        assert st.status() == 0
        got_b, got_r = base.cpu().numpy(), rot.cpu().numpy()
        if kind == "cubic":
            ref = so.oracle_blockwise_cubic(x, L, 8, False)
            _close(got_b, ref, "device cubic")
            _close(got_r, x - ref, "device cubic rotation")
        else:
            rref, bref = so.oracle_blockwise_linear(x, L)
            assert_bits_equal(got_b, bref, "device linear baseline")
            assert_bits_equal(got_r, rref, "device linear rotation")
        assert st.blocks_held == 0
        st.close()


def test_stream_reuse_after_flush_and_errors(S, so):
    from pyitd_amd import ITDError
    rng = np.random.default_rng(8)
    st = S.BlockwiseLinear(512)
    x = rng.standard_normal(512 * 3)
    for rep in range(2):              # flush empties the stream: the second run starts afresh
        outs = []
        for k in range(3):
            r = st.push(x[k * 512:(k + 1) * 512])
            assert (r is None) == (k == 0)
            if r is not None:
                outs.append(r[1])
        outs.append(st.flush()[1])
        assert_bits_equal(np.concatenate(outs), so.oracle_blockwise_linear(x, 512)[1], "rep %d" % rep)
    assert st.flush() is None
    with pytest.raises(ValueError):
        st.push(np.zeros(100))
    bad = x[:512].copy()
    bad[7] = np.nan
    with pytest.raises(ITDError):
        st.push(bad)
        st.push(bad)
    st.reset()
    assert st.status() == 0
    with pytest.raises(ValueError):
        S.Stream(4, 1)
    with pytest.raises(ValueError):
        S.Stream(64, 2, "linear", shared_knots=True)
    st.close()


# ---- retained extrema along channels, batched rows ---------------------------------------------------------------------
@pytest.mark.parametrize("name", cases("channels_"))
def test_retained_extrema_along_channels_match_reference_goldens(name):
    from pyitd_amd.batch import itd_baseline_extract_fast_channels
    g = np.load(os.path.join(STREAM, name + ".npz"))
    got = itd_baseline_extract_fast_channels(g["x"], g["extrema"], int(g["idx"]))
    _close(got, g["baselines"], name)


def test_cubic_batch_with_own_knots_per_signal(oracle):
    from pyitd_amd.batch import itd_baseline_extract_fast_channels
    rng = np.random.default_rng(21)
    x = np.stack([fuzz_signal(rng, k, 5000) for k in (0, 1, 4, 3)] + [np.linspace(0, 1, 5000)])
    got = itd_baseline_extract_fast_channels(x, None, 0)
    for c in range(x.shape[0]):
        e, idx = oracle.extrema_cpp(x[c])
        if idx < 2:
            assert not got[c].any()            # itd.cpp:170-172: untouched
            continue
        _close(got[c], oracle.itd_baseline_extract_fast(x[c], e, idx), "own knots, channel %d" % c)


def test_batched_tier1_extraction_of_image_rows_bit_exact(oracle):
    """10 240 rows of 512 samples (one sweep stage of siftED2D.ipynb cell 1's workload) through ONE batched call: every row's
    rotation / baseline bit-identical to the oracle's itd_baseline_extract, knot counts exact; rows holding a NaN follow
    detect_peaks' NaN branch through the single-signal path."""
    from pyitd_amd.batch import count_knots_batch, detect_knots_batch, itd_baseline_extract_batch
    rng = np.random.default_rng(2024)
    B, n = 10240, 512
    x = rng.integers(0, 256, (B, n)).astype(np.float64)
    x[5] = np.linspace(0, 1, n)                 # no knots
    x[6, :300] = 0.0                            # leading plateau: NaN on the first segment, as computed
    x[7, 100] = np.nan                          # NaN input
    rot, base, counts = itd_baseline_extract_batch(x, want_counts=True)
    check = list(range(0, 16)) + list(rng.integers(16, B, 240))
    for b in check:
        with np.errstate(all="ignore"):
            r, bs, kn, _ = oracle.itd_baseline_extract(x[b], want_knots=True)
        assert_bits_equal(base[b], bs, "row %d baseline" % b)
        assert_bits_equal(rot[b], r, "row %d rotation" % b)
        assert counts[b] == len(kn), b
    # reconstruction for all rows on the host: rotation + baseline = x wherever finite
    ok = np.isfinite(base)
    assert np.max(np.abs((rot + base - x)[ok])) < 1e-9
    sub = x[:64]
    cnt = count_knots_batch(sub)
    lists = detect_knots_batch(sub)
    for b in range(64):
        k = oracle.knots(sub[b])
        assert cnt[b] == len(k), b
        np.testing.assert_array_equal(lists[b], k)
    cnt3 = count_knots_batch(sub[:8], mode=3)
    for b in (0, 1, 2, 3, 4, 5, 6):
        assert cnt3[b] == oracle.extrema_cpp(sub[b])[1], b
