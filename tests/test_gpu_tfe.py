"""Instantaneous amplitude / phase / frequency of proper rotations (pyitd_amd/csrc/itd_tfe.hpp).  The reference only describes
this step (README.md:13-21, 41-55), so there is no parity target: the kernels are held to an independent numpy statement of
the same definitions (Frei & Osorio 2007, single-wave analysis) and to signals with known answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


def numpy_tfe(x):
    """The definitions, stated independently: half waves between sign changes x[i] -> x[i+1] (i in 1..n-2, strict signs),
    amplitude = max |x| of the half wave, phase by quadrant, frequency = forward phase difference mod 2 pi / 2 pi."""
    n = len(x)
    i = np.arange(1, n - 1)
    zc = i[((x[1:-1] > 0) & (0 > x[2:])) | ((x[1:-1] < 0) & (0 < x[2:]))]
    hw = np.searchsorted(zc, np.arange(n), side="left")          # crossings strictly in front of the sample
    A = np.zeros(len(zc) + 1)
    np.maximum.at(A, hw, np.abs(x))
    amp = A[hw]
    slope = np.empty(n)
    slope[:-1] = x[1:] - x[:-1]
    slope[-1] = x[-1] - x[-2]
    with np.errstate(divide="ignore", invalid="ignore"):
        asn = np.arcsin(np.clip(x / amp, -1, 1))
    ph = np.where(x >= 0, np.where(slope >= 0, asn, np.pi - asn), np.where(slope < 0, np.pi - asn, 2 * np.pi + asn))
    ph = np.where(amp > 0, ph, 0.0)
    dp = np.empty(n)
    dp[:-1] = ph[1:] - ph[:-1]
    dp[-1] = ph[-1] - ph[-2]
    dp = np.where(dp < 0, dp + 2 * np.pi, dp)
    return amp, ph, dp / (2 * np.pi)


def test_matches_the_numpy_statement(P):
    rng = np.random.default_rng(11)
    n = 100003
    t = np.arange(n)
    for x in (np.sin(2 * np.pi * 0.013 * t + 0.3) * (1 + 0.5 * np.sin(2 * np.pi * 0.0004 * t)),
              np.sin(2 * np.pi * (0.002 * t + 4e-8 * t * t)),
              rng.standard_normal(n)):
        a, p, f = P.instantaneous(x)
        ra, rp, rf = numpy_tfe(x)
        assert np.array_equal(a, ra)                          # maxima of the same samples: exact
        assert np.max(np.abs(p - rp)) < 1e-12 and np.max(np.abs(f - rf)) < 1e-12


def test_pure_tone_has_constant_amplitude_and_frequency(P):
    n, f0, amp0 = 1 << 18, 0.01, 2.0
    x = amp0 * np.sin(2 * np.pi * f0 * np.arange(n) + 0.2)
    a, p, f = P.instantaneous(x)
    inner = slice(200, n - 200)
    assert np.all(np.abs(a[inner] - amp0) < 2e-3 * amp0)      # the sampled maximum of each half wave
    assert abs(np.median(f[inner]) - f0) < 1e-4
    assert abs(np.mean(f[inner]) - f0) < 2e-4                 # one wave per 1/f0 samples on average
    assert np.all((p >= 0) & (p <= 2 * np.pi + 1e-12))


def test_rotations_of_a_decomposition(P):
    """End to end: decompose a two-tone signal, analyse its first rotations; the dominant instantaneous frequency of each
    rotation is that of the component it carries."""
    n = 1 << 16
    t = np.arange(n)
    x = np.sin(2 * np.pi * 0.05 * t) + 0.5 * np.sin(2 * np.pi * 0.004 * t + 1.0)
    rows = P.ITD().itd(x, max_iteration=4)
    a0, _, f0 = P.instantaneous(rows[0])
    a1, _, f1 = P.instantaneous(rows[1])
    assert abs(np.median(f0[500:-500]) - 0.05) < 2e-3
    assert abs(np.median(f1[2000:-2000]) - 0.004) < 1e-3
    assert abs(np.median(a0[500:-500]) - 1.0) < 0.05
