"""Pin the CPU oracle (oracle/itd_oracle.c) to the reference: every golden vector under
tests/golden/ was produced by the reference's own functions (oracle/gen_golden.py)."""
import numpy as np
import pytest

from conftest import golden_cases
from helpers import assert_bits_equal, load_golden, sha
from oracle import cpu_oracle


@pytest.mark.parametrize("name", golden_cases())
def test_driver_matches_reference(name):
    g = load_golden(name)
    res = cpu_oracle.itd(g["x"], int(g["max_iteration"]))
    assert res["rows"].shape[0] == int(g["n_rows"])
    assert res["stop"] == str(g["stop"])
    assert tuple(res["baselines"].shape) == tuple(g["baselines_shape"])
    # bit-exact on the whole result, through the canonical-NaN hash the generator recorded
    assert sha(res["rows"]) == str(g["rows_sha256"])
    assert sha(res["baselines"]) == str(g["baselines_sha256"])
    if "rows" in g:
        assert_bits_equal(res["rows"], g["rows"], name + " rows")
        assert_bits_equal(res["baselines"], g["baselines"], name + " baselines")
    else:
        idx = g["sample_idx"]
        assert_bits_equal(res["rows"][:, idx], g["rows_sample"], name + " rows sample")


@pytest.mark.parametrize("name", golden_cases())
def test_knots_and_extract_match_reference(name):
    g = load_golden(name)
    if not bool(g["finite"]):
        pytest.skip("per-level knots are recorded for finite cases only")
    x64 = np.asarray(g["x"], dtype=np.float64)
    res = cpu_oracle.itd(g["x"], int(g["max_iteration"]))
    levels = [x64] + [res["baselines"][j] for j in range(res["baselines"].shape[0])]
    assert len(levels) == len(g["knot_counts"])
    for j, xl in enumerate(levels):
        k = cpu_oracle.knots(xl)
        assert k.dtype == np.int64
        np.testing.assert_array_equal(k, g["knots_L%d" % j], err_msg="%s level %d" % (name, j))
    rot, base = cpu_oracle.itd_baseline_extract(x64)
    assert sha(rot) == str(g["extract_rot_sha256"])
    assert sha(base) == str(g["extract_base_sha256"])
    # detect_peaks / matlab_detect_peaks twins: union of the two = the knot set
    a = cpu_oracle.detect_peaks(x64)
    b = cpu_oracle.detect_peaks(-x64)
    np.testing.assert_array_equal(np.union1d(a, b), g["knots_L0"])
    np.testing.assert_array_equal(cpu_oracle.detect_peaks(x64, matlab=True), b)


@pytest.mark.parametrize("name", golden_cases())
def test_lean_driver_equals_full_driver(name):
    g = load_golden(name)
    m = int(g["max_iteration"])
    full = cpu_oracle.itd(g["x"], m)
    lean = cpu_oracle.itd_lean(g["x"], m, want_knots=True)
    assert lean["stop"] == full["stop"]
    assert_bits_equal(lean["rows"], full["rows"], name)
    if bool(g["finite"]):
        for j in range(min(len(lean["knots"]), len(g["knot_counts"]))):
            np.testing.assert_array_equal(lean["knots"][j], g["knots_L%d" % j])


def test_known_answer_radio_clip():
    """PyITD.ipynb cell 3 recorded output: 9 rows, natural stop, reconstruction error 0.0."""
    import math
    g = load_golden("radio8000_m11")
    res = cpu_oracle.itd(g["x"], 11)
    assert res["rows"].shape == (9, 8000) and res["stop"] == "natural"
    cols = [math.fsum(res["rows"][:, i]) for i in range(8000)]
    assert abs(float(np.sum(g["x"])) - math.fsum(cols)) < 1e-12
    # the per-pass extrema counts the reference prints (ITD.py:403); SURVEY 8a row a6 (vi)
    assert res["knot_counts"].tolist() == [1721, 522, 187, 63, 15, 7, 3, 3, 1]


def test_rejects_bad_arguments():
    with pytest.raises(ValueError):
        cpu_oracle.itd(np.zeros(2), 3)
    with pytest.raises(ValueError):
        cpu_oracle.itd(np.zeros(100), 21)  # row 22 would not fit (ITD.py:384,421)
    with pytest.raises(ValueError):
        cpu_oracle.detect_peaks(np.zeros(2))


def _finite_cases():
    return [n for n in golden_cases() if bool(load_golden(n)["finite"])]


@pytest.mark.parametrize("name", _finite_cases())
def test_numpy_restatement_matches_reference(name):
    """oracle/numpy_itd.py (bench.py's numpy CPU leg) against the reference's golden vectors, bit for bit."""
    from oracle import numpy_itd
    g = load_golden(name)
    try:
        res = numpy_itd.itd(g["x"], int(g["max_iteration"]))
    except ValueError as e:
        if "NaN" in str(e):
            pytest.skip("a baseline goes NaN: outside this leg's domain (the C oracle restates the NaN branch)")
        raise
    assert res["stop"] == str(g["stop"]) and res["rows"].shape[0] == int(g["n_rows"])
    assert sha(res["rows"]) == str(g["rows_sha256"])


@pytest.mark.parametrize("name", [n for n in _finite_cases() if load_golden(n)["x"].shape[0] <= 8000])
def test_numba_restatement_matches_reference(name):
    """oracle/numba_itd.py (bench.py's numba CPU leg; plain Python here when numba is absent) on the small goldens."""
    from oracle import numba_itd
    g = load_golden(name)
    if not np.all(np.isfinite(cpu_oracle.itd(g["x"], int(g["max_iteration"]))["rows"])):
        pytest.skip("a baseline goes NaN: outside this leg's domain")
    res = numba_itd.itd(g["x"], int(g["max_iteration"]))
    assert res["stop"] == str(g["stop"]) and res["rows"].shape[0] == int(g["n_rows"])
    assert sha(res["rows"]) == str(g["rows_sha256"])


def test_oracle_single_level_functions_on_nan_input():
    """detect_peaks / matlab_detect_peaks / itd_baseline_extract on signals that hold NaNs (their NaN branch and the in-place
    NaN -> +inf mutation): the oracle against the reference's own outputs (tests/golden/helpers_nan_input.npz, oracle/gen_golden.py)."""
    from helpers import assert_bits_equal, load_golden
    from oracle import cpu_oracle
    g = load_golden("helpers_nan_input")
    for c in range(int(g["cases"])):
        x = g["x_%d" % c]
        np.testing.assert_array_equal(cpu_oracle.detect_peaks(x), g["valleys_%d" % c])
        np.testing.assert_array_equal(cpu_oracle.detect_peaks(x, matlab=True), g["matlab_%d" % c])
        rot, base = cpu_oracle.itd_baseline_extract(x)
        assert_bits_equal(rot, g["rot_%d" % c], "rotation %d" % c)
        assert_bits_equal(base, g["base_%d" % c], "baseline %d" % c)
