"""njit restatement of the reference's numba path (ITD_numba.py:15-136) — TEST INFRASTRUCTURE / CPU-BASELINE LEG ONLY.

The upstream file cannot be imported anywhere (ITD_numba.py:57 uses an unimported name at import time) and
never travels to the GPU box.  This is the same algorithm, written as the scalar loops numba compiles: when numba is
importable the functions are njit-compiled (`AVAILABLE = True`) and bench.py times them as the "numba (restatement)" leg;
without numba they run as plain Python, which the CPU tests use at small sizes to hold them to the golden vectors.
Finite data only (no NaN branch), like oracle/numpy_itd.py.
"""
import numpy as np

try:
    import numba
    AVAILABLE = True
    _jit = numba.njit(cache=False)
except Exception:           # numba is not installed in this image
    numba = None
    AVAILABLE = False

    def _jit(f):
        return f


@_jit
def _knots(x, e):
    """interior knots of x into e[1..m], returns m (detect_peaks(x) U detect_peaks(-x), ITD_numba.py:15-54, 61-75)."""
    n = x.shape[0]
    m = 0
    d0 = x[1] - x[0]
    for i in range(1, n - 1):
        d1 = x[i + 1] - x[i]
        if (d1 > 0.0 and d0 <= 0.0) or (d1 < 0.0 and d0 >= 0.0):
            m += 1
            e[m] = i
        d0 = d1
    e[0] = 0
    e[m + 1] = n - 1
    return m


@_jit
def _extract(x, rot, base, e, bk):
    """one extraction (ITD_numba.py:56-98); returns m"""
    n = x.shape[0]
    m = _knots(x, e)
    bk[0] = (x[0] + x[1]) / 2.0
    bk[m + 1] = (x[n - 2] + x[n - 1]) / 2.0
    for k in range(1, m + 1):
        frac = (e[k] - e[k - 1]) / (e[k + 1] - e[k - 1])
        bk[k] = 0.5 * (x[e[k - 1]] + frac * (x[e[k + 1]] - x[e[k - 1]])) + 0.5 * x[e[k]]
    for i in range(n):
        base[i] = 0.0
    for k in range(m + 1):
        xe = x[e[k]]
        slope = (bk[k + 1] - bk[k]) / (x[e[k + 1]] - xe)
        for i in range(e[k], e[k + 1]):
            base[i] = bk[k] + slope * (x[i] - xe)
    for i in range(n):
        rot[i] = x[i] - base[i]
    return m


@_jit
def _count(x):
    n = x.shape[0]
    c = 0
    d0 = x[1] - x[0]
    for i in range(1, n - 1):
        d1 = x[i + 1] - x[i]
        if (d1 > 0.0 and d0 <= 0.0) or (d1 < 0.0 and d0 >= 0.0):
            c += 1
        d0 = d1
    return c


@_jit
def _itd(x, max_iteration, rows):
    """driver (ITD_numba.py:100-136 = ITD.py:384-432); returns (n_rows, stop) with stop 0 natural / 1 timeout"""
    n = x.shape[0]
    e = np.zeros(n + 2, dtype=np.int64)
    bk = np.zeros(n + 2)
    rot = np.zeros(n)
    base = np.zeros(n)
    cur = x.copy()
    prev = np.zeros(n)
    _extract(cur, rot, base, e, bk)
    counter = 0
    while True:
        num_extrema = _count(base)
        if num_extrema < 2:
            rows[counter, :] = prev
            return counter + 1, 0
        if counter > max_iteration:
            rows[counter, :] = rot + base
            return counter + 1, 1
        rows[counter, :] = rot
        prev[:] = base
        cur[:] = base
        _extract(cur, rot, base, e, bk)
        counter += 1


def itd(data, max_iteration=11):
    x = np.ascontiguousarray(data, dtype=np.float64)
    rows = np.zeros((max_iteration + 2, x.shape[0]))
    n_rows, stop = _itd(x, int(max_iteration), rows)
    return {"rows": rows[:n_rows], "stop": "natural" if stop == 0 else "timeout"}
