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
def test_entropy_oracle_matches_reference(name):
    """oracle/meitd_oracle.py against the reference's own weighted_permutation_entropy(x, 3, True); the six-bin form the GPU
    operator returns (itd_wpe3_f64) gives the same number through pyitd_amd.meitd's host formula, bit for bit."""
    from oracle import meitd_oracle
    import pyitd_amd.meitd as mm
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    assert meitd_oracle.weighted_permutation_entropy(g["x"], order=3, normalize=True) == float(g["wpe"])
    w, c = meitd_oracle.bins(g["x"])
    assert c.sum() == g["x"].size - 2
    assert mm._entropy_from_bins(w, c, 3, True) == float(g["wpe"])


@pytest.mark.parametrize("name", meitd_cases())
def test_meitd_driver_logic_matches_reference(name, monkeypatch):
    """pyitd_amd/meitd.py's control logic (MEITD.py:344-549) with its device rows and GPU operators replaced by numpy rows and
    the oracle's operators: the selection decisions and the outputs must equal the reference's run bit for bit — including the
    extractions the driver does NOT repeat (the loop of retrieve_proper_rotation whose result upstream discards, MEITD.py:359-368).
    No GPU involved here; the -m gpu twin of this test runs the real operators."""
    from oracle import meitd_oracle
    import pyitd_amd.meitd as mm
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    works = []

    def work_for(n, device=0, solver="auto"):
        works.append(meitd_oracle.CpuWork(n))
        return works[-1]

    monkeypatch.setattr(mm, "_work_for", work_for)
    hi, lo, res = mm.MEITD(g["x"].copy())
    assert hi.shape == g["high"].shape and lo.shape == g["low"].shape
    assert_bits_equal(hi, g["high"], name + " high")
    assert_bits_equal(lo, g["low"], name + " low")
    assert_bits_equal(res, g["residual"], name + " residual")
    assert works[-1].calls["extract"] < 60          # (upstream: up to 292 on these signals)
    assert sorted(works[-1].free_rows) == list(range(6))
    assert_bits_equal(mm.XITD(g["x"].copy()), g["xitd"], name + " XITD")
