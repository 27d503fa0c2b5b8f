"""Pin the block-wise oracle (oracle/stream_oracle.py over oracle/cpu_oracle.py's operators) to tests/golden/stream/*.npz —
the same recipe (itd.cpp:31-44) run by oracle/gen_golden.py over the REFERENCE's own itd_baseline_extract_fast
(itd_fourier_decomposition.py:49-122) and itd_baseline_extract (ITD.py:79-121) — and check the properties the recipe has."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, assert_bits_equal, fuzz_signal
from oracle import cpu_oracle, stream_oracle

STREAM = os.path.join(GOLDEN, "stream")


def cases(prefix):
    return sorted(f[:-4] for f in os.listdir(STREAM) if f.startswith(prefix) and f.endswith(".npz"))


@pytest.mark.parametrize("name", cases("cubic_"))
def test_cubic_stream_matches_reference_operator_bit_for_bit(name):
    g = np.load(os.path.join(STREAM, name + ".npz"))
    got = stream_oracle.oracle_blockwise_cubic(g["x"], int(g["block"]), int(g["margin"]), bool(g["shared_knots"]))
    assert_bits_equal(got, g["baseline"], name)


@pytest.mark.parametrize("name", cases("linear_"))
def test_linear_stream_matches_reference_operator_bit_for_bit(name):
    g = np.load(os.path.join(STREAM, name + ".npz"))
    rot, base = stream_oracle.oracle_blockwise_linear(g["x"], int(g["block"]))
    assert_bits_equal(base, g["baseline"], name + " baseline")
    assert_bits_equal(rot, g["rotation"], name + " rotation")


@pytest.mark.parametrize("name", cases("channels_"))
def test_retained_extrema_along_channels(name):
    g = np.load(os.path.join(STREAM, name + ".npz"))
    got = stream_oracle.oracle_extract_fast_channels(g["x"], g["extrema"], int(g["idx"]))
    assert_bits_equal(got, g["baselines"], name)


def test_window_geometry():
    L = 10
    assert stream_oracle.windows(1, L) == [(0, 10, 0, 10)]
    assert stream_oracle.windows(2, L) == [(0, 20, 0, 10), (0, 20, 10, 20)]
    assert stream_oracle.windows(4, L) == [(0, 20, 0, 10), (0, 30, 10, 20), (10, 30, 10, 20), (20, 20, 10, 20)]
    k = np.array([3, 9, 10, 14, 19, 20, 26, 27, 29])
    # emitted part [10, 20): first knot at or behind 10 is k[2], first knot behind 19 is k[5]
    np.testing.assert_array_equal(stream_oracle.select_knots(k, 10, 20, 1), k[1:8])
    np.testing.assert_array_equal(stream_oracle.select_knots(k, 10, 20, 8), k)


def test_linear_stream_equals_the_whole_signal():
    """ITD.py:79-121 is local: wherever every block holds a few knots the stream is the whole-signal result, bit for bit —
    including baseline[n-1] = 0 at the very end and the end-knot means at both ends (the first / last window ends there)."""
    rng = np.random.default_rng(5)
    for kind, n, L in ((0, 6000, 500), (1, 8192, 1024), (2, 4096, 256), (4, 9000, 3000), (6, 2048, 64)):
        x = fuzz_signal(rng, kind, n)
        rot_w, base_w = cpu_oracle.itd_baseline_extract(x)
        rot, base = stream_oracle.oracle_blockwise_linear(x, L)
        assert_bits_equal(base, base_w, "kind %d baseline" % kind)
        assert_bits_equal(rot, rot_w, "kind %d rotation" % kind)


def test_cubic_stream_approaches_the_whole_signal_with_the_margin():
    """A spline's dependence on far knots decays (~0.27 per knot): with a wide margin the stream equals the whole-signal
    operator away from the stream's two ends; the recipe's literal margin of 1 only has to be finite."""
    rng = np.random.default_rng(9)
    L, nb = 1024, 8
    x = np.cumsum(rng.standard_normal(L * nb)) * 0.05 + np.sin(np.arange(L * nb) / 40.0)
    e, idx = cpu_oracle.extrema_cpp(x)
    whole = cpu_oracle.itd_baseline_extract_fast(x, e, idx)
    inner = slice(L, (nb - 1) * L)
    scale = np.max(np.abs(x))
    for margin, tol in ((40, 1e-9), (8, 1e-2), (1, None)):
        got = stream_oracle.oracle_blockwise_cubic(x, L, margin)
        assert np.all(np.isfinite(got))
        if tol is not None:
            assert np.max(np.abs(got[inner] - whole[inner])) < tol * scale, margin


def test_too_few_extrema_leaves_the_block_unchanged():
    x = np.linspace(0.0, 1.0, 3 * 256) ** 2          # monotone: no extrema at all (itd.cpp:170-172)
    got = stream_oracle.oracle_blockwise_cubic(x, 256, 8)
    assert_bits_equal(got, x, "monotone stream")
