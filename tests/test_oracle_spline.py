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
