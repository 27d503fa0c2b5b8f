"""Pin oracle/spline_oracle.py (the FITPACK flavour of the baseline: numba_accelerated_itd.py:182-211, MEITD.py:303-338,
siftED2D.ipynb cell 1) to the vectors oracle/gen_golden.py produced from the reference's own functions with this image's scipy
(the reference's third-party dependency, interpolate.splrep)."""
import os

import numpy as np
import pytest
import scipy

from helpers import GOLDEN, assert_bits_equal
from oracle import spline_oracle

SPLINE = os.path.join(GOLDEN, "spline")


def row_cases():
    return sorted(f[:-4] for f in os.listdir(SPLINE) if f.startswith("row_"))


@pytest.mark.parametrize("name", row_cases())
def test_rows_match_reference(name):
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    assert str(g["scipy_version"]) == scipy.__version__, "the goldens were generated with another scipy: regenerate"
    assert_bits_equal(spline_oracle.baseline(g["x"], 10), g["baseline"], name)
    if "meitd_baseline" in g:
        b = spline_oracle.baseline(g["x"], 0)
        assert_bits_equal(b, g["meitd_baseline"], name + " (MEITD form)")
        assert_bits_equal(g["x"] - b, g["meitd_rotation"], name + " rotation")


def test_crossways_matches_reference():
    g = np.load(os.path.join(SPLINE, "image48x64.npz"))
    assert_bits_equal(spline_oracle.crossways(g["image"]), g["crossways"], "crossways")


def meitd_cases():
    return sorted(f[:-4] for f in os.listdir(SPLINE) if f.startswith("meitd_"))


@pytest.mark.parametrize("name", meitd_cases())
def test_meitd_driver_logic_matches_reference(name, monkeypatch):
    """pyitd_amd/meitd.py's control logic (MEITD.py:344-549) with its two GPU operators replaced by the oracle's: the
    selection decisions, the entropy and the outputs must equal the reference's run (no GPU involved here; the -m gpu twin
    of this test runs the real operators)."""
    from oracle import cpu_oracle
    import pyitd_amd.meitd as mm

    def extract(x, device=0):
        x = np.asarray(x, dtype=np.float64)
        if cpu_oracle.knots(x).size < 2:
            raise TypeError("m > k must hold")
        b = spline_oracle.baseline(x, 0)
        return x - b, b

    monkeypatch.setattr(mm, "itd_baseline_extract_spline", extract)
    monkeypatch.setattr(mm, "_num_extrema", lambda x, device=0: int(cpu_oracle.knots(np.asarray(x, dtype=np.float64)).size))

    def extract_and_count(x, device=0):       # the fused "extraction + count of its baseline" call of the GPU path
        r, b = extract(x)
        return r, b, int(cpu_oracle.knots(b).size)

    monkeypatch.setattr(mm, "_extract_and_count", extract_and_count)
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    assert mm.weighted_permutation_entropy(g["x"], order=3, normalize=True) == float(g["wpe"])
    hi, lo, res = mm.MEITD(g["x"].copy())
    assert hi.shape == g["high"].shape and lo.shape == g["low"].shape
    assert_bits_equal(hi, g["high"], name + " high")
    assert_bits_equal(lo, g["low"], name + " low")
    assert_bits_equal(res, g["residual"], name + " residual")
    assert_bits_equal(mm.XITD(g["x"].copy()), g["xitd"], name + " XITD")
