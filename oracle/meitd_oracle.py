"""TEST INFRASTRUCTURE — CPU restatement of MEITD.py's entropy (weighted_permutation_entropy, MEITD.py:79-128) and a CPU stand-in
for the device rows pyitd_amd/meitd.py works on.  Only tests/ and tools' CPU legs use this module; the product path never does.

Pinned by tests/golden/spline/meitd_*.npz: `wpe` there is the reference's own weighted_permutation_entropy(x, 3, True), and
high / low / residual / xitd its own MEITD / XITD outputs (oracle/gen_golden.py)."""
from math import factorial

import numpy

from . import cpu_oracle, spline_oracle

HASHES = (5, 7, 11, 15, 19, 21)        # numpy.unique's order of (argsort * [1, 3, 9]).sum(1) over the permutations of (0, 1, 2)


def _embed(x, order=3, delay=1):
    """MEITD.py:48-70"""
    N = len(x)
    Y = numpy.empty((order, N - (order - 1) * delay))
    for i in range(order):
        Y[i] = x[i * delay:i * delay + Y.shape[1]]
    return Y.T


def weighted_permutation_entropy(time_series, order=3, normalize=False, sort_kind="quicksort"):
    """MEITD.py:79-128, expression by expression.  sort_kind: the reference asks numpy for "quicksort"; on rows of 3 values that is
    numpy's insertion sort (ties keep their index order) on every build — on rows of 4 or more an AVX-512 build of numpy sorts with
    x86-simd-sort's networks, whose order of tied values is its own: "stable" is the build-independent reading (tests of orders != 3)."""
    x = numpy.array(time_series)
    hashmult = numpy.power(order, numpy.arange(order))
    sorted_idx = _embed(x, order=order).argsort(kind=sort_kind)
    windows = numpy.lib.stride_tricks.sliding_window_view(x, order)      # = util_rolling_window(x, order), MEITD.py:73-76
    weights = numpy.var(windows, 1)
    hashval = (numpy.multiply(sorted_idx, hashmult)).sum(1)
    counts = []
    for h in numpy.unique(hashval):
        w = weights[numpy.where(hashval == h)[0]]
        counts.append(numpy.cumsum(w)[-1] if w.size else 0.0)            # the reference adds them one by one, in index order
    wc = numpy.array(counts)
    p = numpy.true_divide(wc, wc.sum())
    pe = -numpy.multiply(p, numpy.log2(p)).sum()
    if normalize:
        pe /= numpy.log2(factorial(order))
    return pe


def bins(x):
    """(weights[6], windows[6]) of the order-3 patterns, in HASHES' order: what itd_wpe3_f64 returns (sums one by one, in index
    order, like the reference's cumsum)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    sorted_idx = _embed(x, order=3).argsort(kind="quicksort")
    hashval = (sorted_idx * numpy.array([1, 3, 9])).sum(1)
    weights = numpy.var(numpy.lib.stride_tricks.sliding_window_view(x, 3), 1)
    w, c = numpy.zeros(6), numpy.zeros(6, numpy.int64)
    for b, h in enumerate(HASHES):
        sel = weights[hashval == h]
        c[b] = sel.size
        w[b] = numpy.cumsum(sel)[-1] if sel.size else 0.0
    return w, c


class CpuWork:
    """pyitd_amd.meitd._Work over numpy rows and the CPU oracle's operators (row handles are row numbers)."""

    ROWS_KEPT = 22

    def __init__(self, n):
        self.n = n
        self.rows = numpy.zeros((6 + 2 * self.ROWS_KEPT, n))
        self.high0, self.low0 = 6, 6 + self.ROWS_KEPT
        self.calls = {"extract": 0, "count": 0, "probe": 0}
        self.reset()

    def reset(self):
        self.free_rows = list(range(6))

    def kept(self, first, k):
        return first + k

    def take(self):
        return self.free_rows.pop()

    def give(self, p):
        self.free_rows.append(p)

    def upload(self, x, dst):
        self.rows[dst] = x

    def download(self, src, rows=1):
        return self.rows[src:src + rows].copy()

    def assign(self, dst, src):
        self.rows[dst] = self.rows[src]

    def zero(self, dst):
        self.rows[dst] = 0.0

    def count(self, src):
        self.calls["count"] += 1
        return int(cpu_oracle.knots(self.rows[src]).size)

    def entropy(self, src):
        return weighted_permutation_entropy(self.rows[src], order=3, normalize=True)

    def probe(self, src):
        self.calls["probe"] += 1
        return numpy.mean(self.entropy(src)), int(cpu_oracle.knots(self.rows[src]).size)

    def subtract(self, a, b, out):
        self.rows[out] = self.rows[a] - self.rows[b]

    def extract(self, src, base, rot=None, want_baseline_count=False):
        self.calls["extract"] += 1
        x = self.rows[src].copy()
        if cpu_oracle.knots(x).size < 2:
            raise TypeError("m > k must hold")
        b = spline_oracle.baseline(x, 0)
        self.rows[base] = b
        if rot is not None:
            self.rows[rot] = x - b
        return int(cpu_oracle.knots(b).size) if want_baseline_count else None
