"""CPU restatement of the reference's FITPACK flavour of the baseline — TEST INFRASTRUCTURE ONLY (never imported by pyitd_amd).

  itd_baseline_extract_modified(x)      numba_accelerated_itd.py:182-211   (returns x itself when fewer than 10 extrema, :188-190)
  = itd_baseline_extract(x)             siftED2D.ipynb cell 1               (the same function, baseline only)
  itd_baseline_extract(data)            MEITD.py:303-338                    (no early-out; returns (rotation, baseline))
  crossways_itd_baseline_extract(data)  siftED2D.ipynb cell 1
  retrieve_statistical_image_component  siftED2D.ipynb cell 1

The spline itself is the reference's own third-party call, scipy.interpolate.splrep(x, y, k=3) (numba_accelerated_itd.py:84;
SciPy's FITPACK curfit, SciPy 1.15.3 on this image and on the GPU box; with no weights its default is s = 0: the
interpolating not-a-knot cubic spline), called here exactly as the reference calls it; the evaluation restates numba_splev
(numba_accelerated_itd.py:89-164), including its equi_spaced interval formula.  Pinned by tests/golden/spline/*.npz, which
oracle/gen_golden.py produces from the reference's own functions."""
import numpy as np
from scipy import interpolate

from . import cpu_oracle


def numba_splev(z, coeff):
    """numba_accelerated_itd.py:89-164, vectorised over the (increasing) arguments z = 0 .. n-1."""
    t, c, k, equi_spaced, dx = coeff
    n = t.size
    k1 = k + 1
    nk1 = n - k1
    z = np.asarray(z, dtype=np.float64)
    if equi_spaced:
        l = ((z - t[0]) / dx).astype(np.int64) + k           # int(): truncation towards zero; arguments are >= t0
        l = np.minimum(np.maximum(l, k1), nk1)
    else:
        # the stateful search (:118-125) on increasing arguments ends at the first l >= k1 with arg < t[l], capped at nk1
        l = np.minimum(np.searchsorted(t, z, side="right"), nk1)
        l = np.maximum(l, k1)
    h = np.zeros((z.size, 20))
    h[:, 0] = 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        for j in range(k):
            hh = h.copy()
            h[:, 0] = 0.0
            alive = np.ones(z.size, dtype=bool)              # the reference breaks out of the ll loop at a zero-length span
            for ll in range(j + 1):
                li = l + ll
                lj = li - j - 1
                same = t[li] == t[lj]
                go = alive & ~same
                f = np.where(go, hh[:, ll] / np.where(same, 1.0, t[li] - t[lj]), 0.0)
                h[:, ll] = np.where(go, h[:, ll] + f * (t[li] - z), h[:, ll])
                h[:, ll + 1] = np.where(go, f * (z - t[lj]), np.where(alive & same, 0.0, h[:, ll + 1]))
                alive = alive & ~same
    sp = np.zeros(z.size)
    for j in range(k1):
        sp += c[l - k1 + j] * h[:, j]
    return sp


def knot_values(x, e):
    """baseline knots (numba_accelerated_itd.py:196-206): odd-reflected ends, the interior by baseline_knot_estimation"""
    bk = cpu_oracle.knot_values(x, e)                          # interior: the same formula (:167-178 = ITD.py:106-110)
    p0 = 2 * x[0] - x[1]                                        # numpy.pad(x, 1, 'reflect', reflect_type='odd')
    pn = 2 * x[-1] - x[-2]
    bk[0] = (p0 + x[0]) / 2.0                                   # numpy.mean(padded[:2])
    bk[-1] = (x[-1] + pn) / 2.0                                 # numpy.mean(padded[-2:])
    return bk


def baseline(x, min_extrema=10):
    """itd_baseline_extract_modified (min_extrema = 10) / MEITD's itd_baseline_extract (min_extrema = 0): the baseline."""
    x = np.asarray(x, dtype=np.float64)
    kn = cpu_oracle.knots(x)                                   # matlab_detect_peaks(x) U matlab_detect_peaks(-x): the same set
    if kn.size < min_extrema:
        return x
    e = np.concatenate(([0], kn, [x.size - 1])).astype(np.int64)
    S = knot_values(x, e)
    xd = np.diff(e)
    t, c, k = interpolate.splrep(e, S, k=3)
    return numba_splev(np.arange(x.size, dtype=np.float64), (t, c, k, bool(np.all(xd == xd[0])), xd[0]))


def crossways(data, min_extrema=10):
    """crossways_itd_baseline_extract, siftED2D.ipynb cell 1"""
    data = np.asarray(data, dtype=np.float64)
    lengthwise = np.stack([baseline(r, min_extrema) for r in data])
    crosswise = np.stack([baseline(data[:, j], min_extrema) for j in range(data.shape[1])], axis=1)
    crosswise = np.stack([baseline(r, min_extrema) for r in crosswise])
    lengthwise = np.stack([baseline(lengthwise[:, j], min_extrema) for j in range(data.shape[1])], axis=1)
    return (lengthwise + crosswise) / 2.0
