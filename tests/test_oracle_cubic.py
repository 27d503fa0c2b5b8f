"""Pin the cubic-variant restatement (oracle/itd_oracle.c: oracle_find_extrema, oracle_extrema_cpp,
oracle_itd_baseline_extract_fast) to the reference: tests/golden/cubic/*.npz were produced by the reference's own
itd_baseline_extract_fast / find_extrema (itd_fourier_decomposition.py:17-31, :49-122 — the Python twin of
itd.cpp:156-239) through oracle/gen_golden.py."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, assert_bits_equal
from oracle import cpu_oracle

CUBIC = os.path.join(GOLDEN, "cubic")


def cubic_cases():
    return sorted(f[:-4] for f in os.listdir(CUBIC) if f.endswith(".npz"))


@pytest.mark.parametrize("name", cubic_cases())
def test_extract_fast_matches_reference_bit_for_bit(name):
    g = np.load(os.path.join(CUBIC, name + ".npz"))
    base = cpu_oracle.itd_baseline_extract_fast(g["I"], g["extrema"], int(g["idx"]))
    # same glibc pow as numpy's float64 ** int on this image: bit equality holds here; the GPU tests use a tolerance
    assert_bits_equal(base, g["baseline"], name)


@pytest.mark.parametrize("name", [c for c in cubic_cases() if "detect" in c])
def test_cpp_knot_predicate_reproduces_the_golden_knots(name):
    g = np.load(os.path.join(CUBIC, name + ".npz"))
    e, idx = cpu_oracle.extrema_cpp(g["I"])
    assert idx == int(g["idx"])
    np.testing.assert_array_equal(e[: idx + 1], g["extrema"])


def test_find_extrema_restatement():
    # the golden knot lists of the sine cases come from the reference's find_extrema on numpy-generated sines
    for name in [c for c in cubic_cases() if c.startswith("cubic_radio4000_sine")]:
        g = np.load(os.path.join(CUBIC, name + ".npz"))
        f = float(name[len("cubic_radio4000_sine"):])
        sr, n = 8000, g["I"].shape[0]
        s = np.sin(2 * np.pi * f * np.arange(0, n / sr, 1 / sr))      # generate_sine_wave, itd_fourier_decomposition.py:11-14
        e, idx = cpu_oracle.find_extrema(s)
        assert idx == int(g["idx"]), name
        np.testing.assert_array_equal(e[: idx + 1], g["extrema"])
