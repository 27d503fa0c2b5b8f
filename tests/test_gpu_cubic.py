"""The cubic-spline baseline variant with externally supplied knots on the GPU (itd_cubic.hpp) against the
reference-generated goldens (tests/golden/cubic) and the CPU oracle.  Knot indices: exact.  Floats: within 1e-9 of the
signal's scale (the reference's numpy form evaluates t**3 with libm pow, numba multiplies, and the knot recurrences run
as scans here: see include/pyitd_hip.h)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from test_oracle_cubic import cubic_cases

pytestmark = pytest.mark.gpu
CUBIC = os.path.join(GOLDEN, "cubic")
TOL = 1e-9


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _close(got, ref, scale, what):
    assert got.shape == ref.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what
    err = np.nanmax(np.abs(got - ref)) if got.size else 0.0
    assert err <= TOL * scale, "%s: max |diff| %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("name", cubic_cases())
def test_extract_fast_matches_reference_goldens(P, name):
    g = np.load(os.path.join(CUBIC, name + ".npz"))
    base = P.itd_baseline_extract_fast(g["I"], g["extrema"], int(g["idx"]))
    _close(base, g["baseline"], max(1.0, float(np.max(np.abs(g["baseline"])))), name)


@pytest.mark.parametrize("name", [c for c in cubic_cases() if "detect" in c])
def test_detect_mode_reproduces_the_golden_knots_and_baseline(P, name):
    g = np.load(os.path.join(CUBIC, name + ".npz"))
    base, knots = P.itd_baseline_extract_cubic(g["I"], want_knots=True)
    idx = int(g["idx"])
    np.testing.assert_array_equal(knots, g["extrema"][:idx])       # int64, exact
    _close(base, g["baseline"], max(1.0, float(np.max(np.abs(g["baseline"])))), name)


def test_find_extrema_matches_oracle(P, oracle):
    for f, sr, n in ((7.0, 8000, 4000), (440.0, 8000, 4000), (3990.0, 8000, 4000), (97.0, 48000, 100000), (12000.0, 48000, 70001)):
        s = P.generate_sine_wave(f, sr, n / sr)
        e, idx = P.find_extrema(s)
        e2, idx2 = oracle.find_extrema(s)
        assert idx == idx2, (f, sr)
        np.testing.assert_array_equal(e, e2)


def test_large_signal_many_knots_vs_oracle(P, oracle):
    """2^21 samples with ~4e5 knots: the recurrences span hundreds of workgroups (reduce / carries / apply)."""
    rng = np.random.default_rng(31)
    n = 1 << 21
    x = np.cumsum(rng.standard_normal(n)) * 0.01 + np.sin(np.arange(n) / 50.0)
    e, idx = oracle.extrema_cpp(x)
    ref = oracle.itd_baseline_extract_fast(x, e, idx)
    base, knots = P.itd_baseline_extract_cubic(x, want_knots=True)
    assert len(knots) == idx > 300000
    np.testing.assert_array_equal(knots, e[:idx])
    _close(base, ref, float(np.max(np.abs(x))), "2^21 detect")
    # external knots: every 7th detected knot, the reference's own zero-terminated convention
    sub = np.concatenate([e[:idx:7], [0]]).astype(np.int64)
    ref2 = oracle.itd_baseline_extract_fast(x, sub, len(sub) - 1)
    got2 = P.itd_baseline_extract_fast(x, sub, len(sub) - 1)
    _close(got2, ref2, float(np.max(np.abs(x))), "2^21 external knots")


def test_bad_knot_lists_are_rejected(P):
    x = np.sin(np.arange(3000.0) / 9)
    with pytest.raises(P.ITDError):
        P.itd_baseline_extract_fast(x, np.array([0, 10, 10, 50, 0]), 4)        # not strictly increasing
    with pytest.raises(IndexError):
        P.itd_baseline_extract_fast(x, np.array([0, 10, 20, 5000, 0]), 4)      # outside the signal (the reference's IndexError)
    with pytest.raises(P.ITDError):
        P.itd_baseline_extract_fast(x, np.array([0, 10, 20, -5, 0]), 4)        # negative: python would wrap, the ABI refuses
    with pytest.raises(P.ITDError):
        P.itd_baseline_extract_fast(x, np.array([0, 0]), 1)                    # idx < 2
    flat = np.linspace(0, 1, 500)                                              # no knots: itd.cpp:170 leaves the buffer alone
    out, kn = P.itd_baseline_extract_cubic(flat, want_knots=True)
    assert len(kn) == 0 and np.array_equal(out, flat)


def test_sine_wrapper_bands_sum_back(P, oracle):
    """itd_sine_wrapper (itd_fourier_decomposition.py:33-47) at a small sample rate: the bands sum back to the signal, and
    the first band equals the oracle's extraction with the same knots."""
    sr, n = 600, 1801       # chosen so that every sine's extrapolated last knot lies inside the signal (elsewhere the reference
    rng = np.random.default_rng(3)     # itself fails: IndexError in numpy, an out-of-bounds read under numba)
    sig = np.sin(2 * np.pi * 40 * np.arange(n) / sr) + 0.3 * rng.standard_normal(n)
    bands = P.itd_sine_wrapper(sig, sr)
    assert len(bands) == len(np.arange(2, sr // 2 - 1, 96))
    assert np.max(np.abs(np.sum(bands, axis=0) - sig)) < 1e-9
    with pytest.raises(IndexError):       # the reference's own failure mode, mirrored
        P.itd_sine_wrapper(sig[:1000], 500)
    f1 = np.arange(2, sr // 2 - 1, 96)[::-1][1]
    e, idx = oracle.find_extrema(P.generate_sine_wave(f1, sr, n / sr))
    ref = sig - oracle.itd_baseline_extract_fast(sig, e, idx)
    _close(bands[0], ref, float(np.max(np.abs(sig))), "first band")
