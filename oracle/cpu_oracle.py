"""ctypes bindings for oracle/itd_oracle.c — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference ITD path (see the header of
itd_oracle.c for the reference file:line of every function).  It may be imported only
by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by pyitd_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libitd_oracle.so")
MAX_ROWS = 22

_lib = None


def build(force=False):
    src = os.path.join(_HERE, "itd_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32 = ctypes.c_int64, ctypes.c_int32
        P = ctypes.c_void_p
        L.oracle_detect_peaks.restype = i64
        L.oracle_detect_peaks.argtypes = [P, i64, P]
        L.oracle_matlab_detect_peaks.restype = i64
        L.oracle_matlab_detect_peaks.argtypes = [P, i64, P]
        L.oracle_knots.restype = i64
        L.oracle_knots.argtypes = [P, i64, P]
        L.oracle_knot_values.restype = None
        L.oracle_knot_values.argtypes = [P, i64, P, i64, P]
        L.oracle_itd_baseline_extract.restype = ctypes.c_int
        L.oracle_itd_baseline_extract.argtypes = [P, i64, P, P, P, P, P]
        for name in ("oracle_itd", "oracle_itd_f32"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [P, i64, i32, P, P, P, P, P, P, P]
        for name in ("oracle_itd_lean", "oracle_itd_lean_f32"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [P, i64, i32, P, P, P, P, P, P]
        for name in ("oracle_find_extrema", "oracle_extrema_cpp"):
            f = getattr(L, name)
            f.restype = i64
            f.argtypes = [P, i64, P]
        L.oracle_itd_baseline_extract_fast.restype = ctypes.c_int
        L.oracle_itd_baseline_extract_fast.argtypes = [P, i64, P, i64, P, P, P]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def detect_peaks(x, matlab=False):
    """detect_peaks (ITD.py:33-76) / matlab_detect_peaks (numba_accelerated_itd.py:17-59).
    Works on a private copy (the reference mutates NaN->inf in place)."""
    x = np.array(x, dtype=np.float64, copy=True)
    idx = np.empty(x.shape[0], dtype=np.int64)
    f = lib().oracle_matlab_detect_peaks if matlab else lib().oracle_detect_peaks
    c = f(_p(x), x.shape[0], _p(idx))
    if c < 0:
        raise ValueError("detect_peaks needs at least 3 samples")
    return idx[:c].copy()


def knots(x):
    """Interior knot indices of x (ITD.py:87-98)."""
    x = np.array(x, dtype=np.float64, copy=True)
    idx = np.empty(x.shape[0], dtype=np.int64)
    c = lib().oracle_knots(_p(x), x.shape[0], _p(idx))
    if c < 0:
        raise ValueError("knots needs at least 3 samples")
    return idx[:c].copy()


def knot_values(x, e):
    """Knot values B_k for the extended knot list e = [0, ..., n-1] (ITD.py:100-110)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    e = np.ascontiguousarray(e, dtype=np.int64)
    bk = np.empty(e.shape[0], dtype=np.float64)
    lib().oracle_knot_values(_p(x), x.shape[0], _p(e), e.shape[0] - 2, _p(bk))
    return bk


def itd_baseline_extract(x, want_knots=False):
    """(rotation, baseline) of one extraction (ITD.py:79-121)."""
    x = np.array(x, dtype=np.float64, copy=True)
    n = x.shape[0]
    rot = np.empty(n)
    base = np.empty(n)
    kn = np.empty(n, dtype=np.int64)
    bk = np.empty(n + 2)
    m = ctypes.c_int64(0)
    rc = lib().oracle_itd_baseline_extract(_p(x), n, _p(rot), _p(base), _p(kn), ctypes.byref(m), _p(bk))
    if rc:
        raise ValueError("oracle_itd_baseline_extract failed: %d" % rc)
    if want_knots:
        return rot, base, kn[: m.value].copy(), bk[: m.value + 2].copy()
    return rot, base


def itd(x, max_iteration=11):
    """Full driver (ITD.py:384-432).  Returns dict(rows, baselines, stop, knot_counts)."""
    x = np.ascontiguousarray(x)
    n = x.shape[0]
    rows = np.empty((MAX_ROWS, n))
    bases = np.empty((MAX_ROWS, n))
    n_rows, n_b, stop, n_c = (ctypes.c_int32(0) for _ in range(4))
    counts = np.zeros(MAX_ROWS + 1, dtype=np.int64)
    if x.dtype == np.float32:
        f = lib().oracle_itd_f32
    else:
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = lib().oracle_itd
    rc = f(_p(x), n, int(max_iteration), _p(rows), _p(bases), ctypes.byref(n_rows), ctypes.byref(n_b),
           ctypes.byref(stop), _p(counts), ctypes.byref(n_c))
    if rc:
        raise ValueError("oracle_itd failed: %d" % rc)
    return {
        "rows": rows[: n_rows.value].copy(),
        "baselines": bases[: n_b.value].copy(),
        "stop": "natural" if stop.value == 0 else "timeout",
        "knot_counts": counts[: n_c.value].copy(),
    }


def itd_lean(x, max_iteration, want_knots=False, rows_out=None):
    """Lean driver for big N: rows [(max_iteration+2), n] and optional per-level knot lists."""
    x = np.ascontiguousarray(x)
    n = x.shape[0]
    R = max_iteration + 2
    rows = rows_out if rows_out is not None else np.empty((R, n))
    kn = np.empty((R, n), dtype=np.int64) if want_knots else None
    km = np.zeros(R, dtype=np.int64)
    last = np.empty(n)
    n_rows, stop = ctypes.c_int32(0), ctypes.c_int32(0)
    if x.dtype == np.float32:
        f = lib().oracle_itd_lean_f32
    else:
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = lib().oracle_itd_lean
    rc = f(_p(x), n, int(max_iteration), _p(rows), ctypes.byref(n_rows), ctypes.byref(stop), _p(kn), _p(km),
           _p(last))
    if rc:
        raise ValueError("oracle_itd_lean failed: %d" % rc)
    out = {"rows": rows[: n_rows.value], "stop": "natural" if stop.value == 0 else "timeout",
           "last_baseline": last}
    # the number of extractions run equals the number of rows returned (both stop rules)
    if want_knots:
        out["knots"] = [kn[j, : km[j]].copy() for j in range(n_rows.value)]
    out["knot_counts"] = km[: n_rows.value].copy()
    return out


# ---- cubic-spline baseline variant (itd_fourier_decomposition.py:17-31, :49-122; itd.cpp:156-239) ----------------------
def find_extrema(signal):
    """find_extrema (itd_fourier_decomposition.py:17-31): (extrema int64[n], idx)."""
    s = np.ascontiguousarray(signal, dtype=np.float64)
    e = np.empty(s.shape[0], dtype=np.int64)
    idx = lib().oracle_find_extrema(_p(s), s.shape[0], _p(e))
    if idx < 0:
        raise ValueError("find_extrema needs at least 2 samples")
    return e, int(idx)


def extrema_cpp(x):
    """The knot predicate of itd.cpp:161-168 (compute_extrema = true): (extrema int64[n] zero padded, idx)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    e = np.empty(x.shape[0], dtype=np.int64)
    idx = lib().oracle_extrema_cpp(_p(x), x.shape[0], _p(e))
    if idx < 0:
        raise ValueError("needs at least 3 samples")
    return e, int(idx)


def itd_baseline_extract_fast(I, extrema_input, idx, want_coef=False):
    """itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122): baseline float64[n]."""
    I = np.ascontiguousarray(I, dtype=np.float64)
    e = np.ascontiguousarray(extrema_input, dtype=np.int64)
    n = I.shape[0]
    if e.shape[0] < idx + 1:
        raise ValueError("extrema_input needs idx+1 entries")
    base = np.empty(n)
    K = np.empty(n) if want_coef else None
    b = np.empty(n) if want_coef else None
    rc = lib().oracle_itd_baseline_extract_fast(_p(I), n, _p(e), int(idx), _p(base), _p(K), _p(b))
    if rc:
        raise ValueError("oracle_itd_baseline_extract_fast failed: %d" % rc)
    return (base, K, b) if want_coef else base
