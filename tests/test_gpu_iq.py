"""The common-baseline form of the cubic operator on complex (I/Q) data on the GPU (itd_baseline_extract_iq_*, itd.cpp:58-154)
against oracle/iq_oracle.py: knot indices exact, the baseline within 1e-9 of the signal's scale (the sweeps run as scans: DESIGN.md
section 7).  Parity of the recipe itself is unpinned upstream (no test, no Python twin: oracle/iq_oracle.py's header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


def _signals():
    rng = np.random.default_rng(11)
    t = np.arange(200000) / 48000.0
    yield "tone + noise", np.exp(2j * np.pi * 440 * t) + 0.2 * (rng.standard_normal(t.size) + 1j * rng.standard_normal(t.size))
    yield "random walk", np.cumsum(rng.standard_normal(70001)) + 1j * np.cumsum(rng.standard_normal(70001))
    yield "white, ragged", rng.standard_normal(5003) + 1j * rng.standard_normal(5003)
    yield "short", rng.standard_normal(40) + 1j * rng.standard_normal(40)
    z = rng.standard_normal(3000) + 1j * rng.standard_normal(3000)
    z[1000:1040] = z[1000]                                    # a plateau in both components
    yield "plateau", z


def test_iq_baseline_matches_the_oracle(P):
    from oracle import iq_oracle
    for name, z in _signals():
        ref, e, idx = iq_oracle.itd_baseline_extract_iq(z)
        got, kn, gi = P.itd_baseline_extract_iq(z, want_knots=True)
        assert gi == idx and kn.tolist() == e[:idx].tolist(), name
        assert got.dtype == np.float64 and got.shape == (len(z),)
        scale = np.max(np.abs(z))
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * scale, err_msg=name)


def test_iq_with_retained_knots_and_the_degenerate_cases(P):
    from oracle import iq_oracle
    rng = np.random.default_rng(12)
    z = rng.standard_normal(30000) + 1j * rng.standard_normal(30000)
    _, e, idx = iq_oracle.itd_baseline_extract_iq(z)
    w = z * np.exp(0.7j) - 0.25
    ref, _, _ = iq_oracle.itd_baseline_extract_iq(w, e, idx)
    got = P.itd_baseline_extract_iq(w, e, idx)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.max(np.abs(w)))
    # fewer than 2 common knots: the file leaves the caller's buffer alone — the drop-in returns the mean series unchanged
    mono = np.arange(100.0) + 1j * np.arange(100.0) ** 2
    out, kn, gi = P.itd_baseline_extract_iq(mono, want_knots=True)
    assert gi == 0 and kn.size == 0
    np.testing.assert_array_equal(out, (mono.real + mono.imag) / 2.0)
    with pytest.raises(ValueError):
        P.itd_baseline_extract_iq(np.zeros(2, dtype=complex))
    from pyitd_amd import ITDError
    with pytest.raises(ITDError):
        P.itd_baseline_extract_iq(z, np.array([5, 3, 9, 0]), 3)          # not increasing


def test_iq_at_full_size(P):
    """2^22 complex samples: knots exact against numpy, the baseline against the oracle on a strided sample."""
    from oracle import iq_oracle
    rng = np.random.default_rng(13)
    n = 1 << 22
    z = (np.sin(np.arange(n) * 0.01) + 0.3 * rng.standard_normal(n)) + 1j * (np.cos(np.arange(n) * 0.013) + 0.3 * rng.standard_normal(n))
    ref, e, idx = iq_oracle.itd_baseline_extract_iq(z)
    got, kn, gi = P.itd_baseline_extract_iq(z, want_knots=True)
    assert gi == idx and np.array_equal(kn, e[:idx])
    np.testing.assert_allclose(got[::97], ref[::97], rtol=0, atol=1e-9 * np.max(np.abs(z)))
