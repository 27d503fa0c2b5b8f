"""The restatement of FITPACK's curfit in pyitd_amd/csrc/itd_fitpack.hpp (what the GPU runs, one instance per signal), built for
the host with g++ and held to scipy.interpolate.splrep on this image — the third-party routine the reference calls
(numba_accelerated_itd.py:84): knot vectors identical, coefficients bit for bit, for the interpolating call (s = 0, the
reference's) and for the smoothing branches (explicit s).  A check of the restatement, not a fallback: pyitd_amd never loads
this build."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy import interpolate

from oracle import cpu_oracle, spline_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fitpack") / "libfitpack_host.so")
    subprocess.run(["g++", "-O2", "-std=c++14", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                    os.path.join(ROOT, "tests", "c_client", "fitpack_host.cpp")], check=True, capture_output=True)
    L = ctypes.CDLL(so)
    P = ctypes.c_void_p
    L.fitpack_host_splrep.argtypes = [P, P, ctypes.c_int, ctypes.c_double, P, P, P, P]
    L.fitpack_host_interp.argtypes = [P, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_double, P]
    return L


def _data(rng, trial):
    m = int(rng.integers(12, 400))
    x = np.sort(rng.choice(np.arange(0, m * 4), m, replace=False)).astype(np.float64)
    kind = trial % 4
    if kind == 0:
        y = rng.standard_normal(m) * rng.choice([0.1, 1, 10, 100])
    elif kind == 1:
        y = np.cumsum(rng.standard_normal(m)) * 5
    elif kind == 2:
        y = 100 * np.sin(x / rng.uniform(3, 50)) + rng.standard_normal(m) * rng.choice([0.01, 1, 5])
    else:
        y = rng.integers(0, 256, m).astype(float)
    return x, y


@pytest.mark.parametrize("smoothing", [False, True])
def test_curfit_restatement_equals_scipy_splrep(host, smoothing):
    import warnings
    rng = np.random.default_rng(0)
    for trial in range(120):
        x, y = _data(rng, trial)
        m = len(x)
        s = float(m - np.sqrt(2 * m)) if smoothing else 0.0
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0, c0, _ = interpolate.splrep(x, y, k=3, s=s)
        t1, c1 = np.zeros(m + 4), np.zeros(m + 4)
        n, fp = ctypes.c_int(0), ctypes.c_double(0)
        host.fitpack_host_splrep(x.ctypes.data, y.ctypes.data, m, s, t1.ctypes.data, c1.ctypes.data, ctypes.byref(n), ctypes.byref(fp))
        assert n.value == len(t0), trial
        assert np.array_equal(t0, t1[: n.value]), trial
        assert np.array_equal(c0, c1[: n.value]), trial          # bit for bit


def test_gpu_form_equals_the_reference_baseline(host):
    """interp_fit + spline_eval (implicit knots from the int32 knot list, strided arrays) against scipy's coefficients and the
    oracle's baseline (which is pinned to the reference by tests/golden/spline), including the equi_spaced quirk."""
    rng = np.random.default_rng(1)
    checked = 0
    for trial in range(150):
        n = int(rng.choice([64, 100, 512, 513, 2000, 777]))
        kind = trial % 5
        if kind == 0:
            x = rng.integers(0, 256, n).astype(float)
        elif kind == 1:
            x = np.cumsum(rng.standard_normal(n))
        elif kind == 2:
            x = np.sin(np.arange(n) / rng.uniform(2, 30)) + 0.1 * rng.standard_normal(n)
        elif kind == 3:
            x = np.round(rng.standard_normal(n) * 3)
        else:
            x = ((-1.0) ** np.arange(n)) * (1 + rng.random(n))       # every sample an extremum: equi_spaced
        kn = cpu_oracle.knots(x)
        if kn.size < 10:
            continue
        e = np.concatenate(([0], kn, [n - 1])).astype(np.int64)
        S = spline_oracle.knot_values(x, e)
        t, c, _ = interpolate.splrep(e, S, k=3)
        xd = np.diff(e)
        ref = spline_oracle.baseline(x)
        e32, m = e.astype(np.int32), len(e)
        c1, ev = np.zeros(m), np.zeros(n)
        host.fitpack_host_interp(e32.ctypes.data, S.ctypes.data, m, c1.ctypes.data, n, int(np.all(xd == xd[0])), float(xd[0]),
                                 ev.ctypes.data)
        assert np.array_equal(c[:m], c1), trial
        assert np.array_equal(ref.view(np.uint64), ev.view(np.uint64)), trial
        checked += 1
    assert checked > 100
