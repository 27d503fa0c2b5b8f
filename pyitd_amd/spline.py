"""Host-side mirror of the reference's FITPACK flavour of the baseline and of its 2-D / ensemble consumers, backed by the HIP
engine (pyitd_amd/csrc/itd_spline.hpp, itd_fitpack.hpp).  Same names, arguments and returns as

  numba_accelerated_itd.py   itd_baseline_extract_modified(x) :182-211
  siftED2D.ipynb cell 1      itd_baseline_extract(x), mad(arr), crossways_itd_baseline_extract(data),
                             retrieve_statistical_image_component(data), totalextract2d(data)
  MEITD.py                   itd_baseline_extract(data) -> (rotation, baseline) :303-338

Every numeric step of the operator (knots, knot values, spline coefficients, evaluation, the transposes and the mean of the
crossways sweep) runs on the GPU; the ensemble's noise comes from numpy's global generator exactly as upstream
(numpy.random.normal in the same order and shapes), so a seeded run reproduces the reference's.
"""
import time

import numpy

from .itd import _engine_for


SOLVERS = {"auto": 0, "serial": 1, "parallel": 2}


def _eng(n, device, solver="auto"):
    """solver: "serial" = FITPACK's own sweep, one GPU thread per signal (bit-level against scipy's splrep); "parallel" = the same
    interpolating not-a-knot spline from its second derivatives, parallel in the knots (equal to rounding); "auto" = parallel for
    calls of few long signals (include/pyitd_hip.h: itd_set_spline_solver)."""
    eng = _engine_for(max(int(n), 4096), device)
    eng.set_spline_solver(SOLVERS[solver])
    return eng


def itd_baseline_extract_modified(x, device=0, solver="auto"):
    """numba_accelerated_itd.py:182-211 — the cubic-spline baseline (float64[N]); x itself when fewer than 10 extrema."""
    x = numpy.asarray(x, dtype=numpy.float64)
    base, _, knots = _eng(len(x), device, solver).spline_extract_host(x[None, :], 10)
    return x if knots[0] < 10 else base[0]


def itd_baseline_extract_spline(data, device=0, solver="auto"):
    """MEITD.py:303-338 — (rotation, baseline), no early-out (scipy's splrep needs at least 2 extrema: TypeError like upstream)."""
    x = numpy.asarray(data, dtype=numpy.float64)
    base, rot, knots = _eng(len(x), device, solver).spline_extract_host(x[None, :], 0, want_rotation=True)
    if knots[0] < 2:
        raise TypeError("m > k must hold")        # what scipy.interpolate.splrep raises for fewer than 4 data sites
    return rot[0], base[0]


def itd_baseline_extract_rows(x, min_extrema=10, device=0, solver="auto"):
    """The batched form the reference only has as numba.prange over rows: x[B, N] -> baselines[B, N]."""
    x = numpy.asarray(x, dtype=numpy.float64)
    return _eng(x.shape[1], device, solver).spline_extract_host(x, min_extrema)[0]


def mad(arr):
    """siftED2D.ipynb cell 1 — median absolute deviation (host numpy, as upstream)."""
    med = numpy.median(arr)
    return numpy.median(numpy.abs(arr - med))


def crossways_itd_baseline_extract(data, device=0):
    """siftED2D.ipynb cell 1 — rows, then columns of that; columns, then rows of that; the mean (float64[H, W])."""
    data = numpy.asarray(data, dtype=numpy.float64)
    return _eng(max(data.shape), device).crossways_host(data[None], 10)[0]


def retrieve_statistical_image_component(data, device=0):
    """siftED2D.ipynb cell 1 — 20 ensemble members (10 noise draws and their mirror images) through the crossways sweep in
    ONE engine call (20 x (H + W) x 2 independent signals per sweep stage), averaged pairwise and overall."""
    data = numpy.asarray(data, dtype=numpy.float64)
    m = mad(data)
    iterations = 20
    n = numpy.zeros((iterations, data.shape[0], data.shape[1]), dtype=numpy.float64)
    for each in range(iterations // 2):
        v = numpy.random.normal(0, m, data.shape)
        n[each * 2, :, :] = v + data
        n[each * 2 + 1, :, :] = (v * -1) + data
    n = _eng(max(data.shape), device).crossways_host(n, 10)
    b = (n[0::2] + n[1::2]) / 2.0
    x = numpy.zeros(data.shape, dtype=numpy.float64)
    for each in range(iterations // 2):          # the reference accumulates in this order
        x += b[each]
    return x / (iterations // 2 * 1.0)


def totalextract2d(data, device=0, verbose=True):
    """siftED2D.ipynb cell 1 — [higherpass, lowpass]; prints the wall time like upstream (its recorded 10.1457 s is the only
    timing the reference publishes)."""
    t = time.time()
    data = numpy.asarray(data).astype(dtype=numpy.float64)
    lowpass = retrieve_statistical_image_component(data, device)
    higherpass = data - lowpass
    if verbose:
        print(time.time() - t)
    return numpy.asarray([higherpass, lowpass])
