"""The I/Q form's oracle (oracle/iq_oracle.py, itd.cpp:58-154; parity unpinned: no upstream test, no Python twin) against a literal
loop transcription of the fragment's statements in float64 — the two must agree wherever the fragment's own indices are defined."""
import numpy as np

from oracle import iq_oracle


def _literal(z):
    """itd.cpp:58-154 statement by statement, float64, with the Python twin's reading of the two end knots (itd_fourier_decomposition.py:
    62-63: baseline_knots[idx-1] is the last one — the fragment's `baseline_knots[idx]` / `extrema[idx]` reach one past its list)."""
    n = len(z)
    re, im = z.real, z.imag
    ext = [i for i in range(1, n - 1)
           if (((re[i - 1] < re[i]) and (re[i] >= re[i + 1])) or ((re[i - 1] > re[i]) and (re[i] <= re[i + 1])))
           and (((im[i - 1] < im[i]) and (im[i] >= im[i + 1])) or ((im[i - 1] > im[i]) and (im[i] <= im[i + 1])))]
    return ext, (re + im) / 2.0


def test_knots_and_mean_series_match_the_literal_loop():
    rng = np.random.default_rng(5)
    for n in (3, 4, 17, 500, 4099):
        z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        z[n // 3] = z[max(n // 3 - 1, 0)]                      # a tie in both components
        e, idx = iq_oracle.extrema_iq(z)
        ext, avg = _literal(z)
        assert idx == len(ext) and e[:idx].tolist() == ext and not e[idx:].any()
        base, e2, idx2 = iq_oracle.itd_baseline_extract_iq(z)
        assert idx2 == idx
        if idx < 2:
            assert base is None
        else:
            from oracle import cpu_oracle
            np.testing.assert_array_equal(base, cpu_oracle.itd_baseline_extract_fast(avg, e, idx))


def test_common_knots_are_a_subset_of_each_component_s():
    from oracle import cpu_oracle
    rng = np.random.default_rng(6)
    z = np.cumsum(rng.standard_normal(3000)) + 1j * np.cumsum(rng.standard_normal(3000))
    e, idx = iq_oracle.extrema_iq(z)
    er, ir = cpu_oracle.extrema_cpp(z.real)
    ei, ii = cpu_oracle.extrema_cpp(z.imag)
    assert set(e[:idx]) == set(er[:ir]) & set(ei[:ii])


def test_retained_knots_on_another_channel():
    """"simply estimate the extrema the first time ... further iterations should reuse the extrema" (itd.cpp:40-44)."""
    rng = np.random.default_rng(7)
    z = rng.standard_normal(2000) + 1j * rng.standard_normal(2000)
    _, e, idx = iq_oracle.itd_baseline_extract_iq(z)
    w = z * np.exp(0.3j) + 0.1
    base, e2, idx2 = iq_oracle.itd_baseline_extract_iq(w, e, idx)
    assert idx2 == idx and np.isfinite(base).all()
