"""TEST INFRASTRUCTURE (oracle/): the common-baseline form of the cubic operator on complex (I/Q) data, `itd_baseline_extract_iq`,
/root/reference/itd.cpp:58-154, restated in numpy.  Only tests/, __graft_entry__.smoke() and bench.py's CPU legs may use this package.

PARITY UNPINNED: itd.cpp is a non-compilable float32 fragment (undeclared types and sizes, duplicate definitions), nothing upstream calls
or tests the I/Q form, and it has no Python twin (the 1-D form has: itd_fourier_decomposition.py:49-122, which oracle/itd_oracle.c
restates and the reference's own outputs pin).  What is stated here, line by line:
  * knots (itd.cpp:74-83): samples 1 .. length-2 at which BOTH components satisfy
        (d[i-1] < d[i] and d[i] >= d[i+1]) or (d[i-1] > d[i] and d[i] <= d[i+1]);
  * the scalar series (itd.cpp:96-103): avg = (I + Q) / 2 at every knot — float64 here, float32 in the fragment;
  * knot values, the tridiagonal sweep and the evaluation (itd.cpp:89-153) are, statement for statement, those of the 1-D form
    (itd.cpp:176-238) with I replaced by avg — i.e. the pinned operator `itd_baseline_extract_fast(avg, extrema, idx)` with the
    file's convention extrema[idx] = 0 (its static array at first call);
  * fewer than 2 knots: the baseline buffer is left alone (itd.cpp:85-87)."""
import numpy as np

from . import cpu_oracle


def extrema_iq(z):
    """(extrema int64[n] zero padded, idx): the knots of itd.cpp:74-83."""
    z = np.asarray(z, dtype=np.complex128)
    n = z.shape[0]

    def flags(d):
        a, b, c = d[:-2], d[1:-1], d[2:]
        return ((a < b) & (b >= c)) | ((a > b) & (b <= c))

    k = np.flatnonzero(flags(z.real) & flags(z.imag)) + 1
    e = np.zeros(n, dtype=np.int64)
    e[: k.size] = k
    return e, int(k.size)


def itd_baseline_extract_iq(z, extrema=None, idx=None):
    """(baseline float64[n] or None when fewer than 2 knots, extrema, idx)."""
    z = np.asarray(z, dtype=np.complex128)
    if extrema is None:
        extrema, idx = extrema_iq(z)
    if idx < 2:
        return None, extrema, idx
    avg = (z.real + z.imag) / 2.0
    return cpu_oracle.itd_baseline_extract_fast(avg, extrema, idx), extrema, idx
