"""Host-side mirror of MEITD.py's selection drivers, on top of the GPU operators:

  weighted_permutation_entropy(time_series, order=3, normalize=False)   MEITD.py:79-128   (orders 2 .. 5)
  retrieve_proper_rotation(x, WPEMAX)                                    MEITD.py:344-368
  determine_if_first_is_proper_rotation(x, WPEMAX)                       MEITD.py:371-392
  MEITD(data, max_iteration=40, WPEMAX=0.6) -> (high, low, residual)     MEITD.py:395-534
  XITD(data)                                                             MEITD.py:536-549

The drivers are control logic around three operators, all of which run on the GPU on arrays that STAY on the GPU for the whole
loop (include/pyitd_hip.h, "MEITD's operators on device-resident signals"): the cubic-spline baseline extraction (MEITD.py:303-338
-> itd_baseline_extract_spline2_f64), the extrema count (matlab_detect_peaks(x).size + matlab_detect_peaks(-x).size ->
itd_count_knots_f64) and the entropy's pass over the samples (itd_wpe3_f64: six weighted sums, taken in the reference's order).
Per pass a few scalars come back — the counts, the six sums — and the threshold tests are drawn from them here; the signal is
uploaded once and the components are downloaded once.

Upstream quirks kept: `max_iteration` is never used by MEITD (:395); XITD passes its WPEMAX estimate in that position (:541), so
MEITD runs with WPEMAX = 0.6 there (the estimate itself, :538-540, has no consumer and is not computed); nothing is printed.
One piece of upstream's work is NOT repeated because nothing can observe it: when the entropy test of retrieve_proper_rotation
fails, its loop (:359-366) keeps extracting down to fewer than 6 extrema and then returns its INPUT (:368) — on the golden signals
that is 209-249 of the ~290 extractions of a MEITD call.
"""
import threading
from math import factorial

import numpy

from .engine import DeviceBuffer
from .spline import _eng


def _entropy_from_bins(weights, windows, order=3, normalize=False):
    """MEITD.py:119-128 from the patterns' weighted counts (in numpy.unique's order; a pattern without windows is absent).  The same
    numpy operations on the same (short) arrays as upstream: the result's bits depend on numpy's summation order and its log2."""
    wc = weights[windows > 0]
    p = numpy.true_divide(wc, wc.sum())
    pe = -numpy.multiply(p, numpy.log2(p)).sum()
    if normalize:
        pe /= numpy.log2(factorial(order))
    return pe


def weighted_permutation_entropy(time_series, order=3, normalize=False, device=0):
    """MEITD.py:79-128: permutation patterns of the embedded series, each window weighted by its variance.  The pass over the samples
    runs on the GPU (order 3, the only one MEITD.py itself uses: itd_wpe3_f64; orders 2, 4, 5: itd_wpe_f64); the entropy is drawn
    here from the patterns' weighted counts.  Orders above 5 are refused (order^order hash values: the operator is not built for
    them)."""
    x = numpy.ascontiguousarray(numpy.array(time_series), dtype=numpy.float64)
    if x.ndim != 1 or not 2 <= int(order) <= 5 or len(x) < int(order):
        raise ValueError("weighted_permutation_entropy: a 1-D series of at least `order` samples, order 2 .. 5")
    buf = DeviceBuffer(x.nbytes, device)
    try:
        buf.upload(x)
        eng = _eng(len(x), device)
        w, c = eng.wpe3_dev(buf.ptr, len(x)) if int(order) == 3 else eng.wpe_dev(buf.ptr, len(x), int(order))
    finally:
        buf.free()
    return _entropy_from_bins(w, c, int(order), normalize)


_ROWS_KEPT = 22          # MEITD returns once more than 20 components are kept (:414-415): 21 rows at most, in either list


class _Work:
    """The device arrays of one MEITD run — rows of n float64 in one allocation: four working rows that change roles by
    renaming (no copies), the two lists of kept rotations."""

    def __init__(self, n, device, solver="auto"):
        self.n, self.device, self.solver = n, device, solver
        self.eng = _eng(n, device, solver)
        self.buf = DeviceBuffer((6 + 2 * _ROWS_KEPT) * n * 8, device)
        self.free_rows = [self.row(i) for i in range(6)]
        self.high0, self.low0 = self.row(6), self.row(6 + _ROWS_KEPT)
        # the whole loop as one launch: where an extraction of this signal is the one-workgroup parallel-in-knots operator anyway
        self.one_launch = 3 <= n <= _ONE_LAUNCH_MAX and (solver == "parallel" or (solver == "auto" and n >= 1024))
        self.last = {}

    def row(self, i):
        return self.buf.ptr + i * self.n * 8

    def kept(self, first, k):
        return first + k * self.n * 8

    def take(self):
        return self.free_rows.pop()

    def give(self, p):
        self.free_rows.append(p)

    # ---- the operators -------------------------------------------------------------------------------------------------------
    def upload(self, x, dst):
        self.eng.copy(dst, x.ctypes.data, x.nbytes, 1, wait=True)

    def download(self, src, rows=1):
        out = numpy.empty((rows, self.n))
        self.eng.copy(out.ctypes.data, src, out.nbytes, 0, wait=True)
        return out

    def assign(self, dst, src):
        self.eng.copy(dst, src, self.n * 8, 2)

    def zero(self, dst):
        self.eng.copy(dst, None, self.n * 8, 3)

    def probe(self, src):
        """(normalised entropy, extrema count) of a row: MEITD.py:346-351 / :373-378."""
        w, c, count = self.eng.wpe3_dev(src, self.n, want_knots=True)
        return numpy.mean(_entropy_from_bins(w, c, 3, True)), count

    def count(self, src):
        return self.eng.count_knots_dev(src, self.n)

    def entropy(self, src):
        """weighted_permutation_entropy(row, order=3, normalize=True)"""
        return _entropy_from_bins(*self.eng.wpe3_dev(src, self.n), 3, True)

    def subtract(self, a, b, out):
        self.eng.subtract_dev(a, b, out, self.n)

    def reset(self):
        """every working row is free again (a call that raised may have left some taken)"""
        self.free_rows = [self.row(i) for i in range(6)]

    def extract(self, src, base, rot=None, want_baseline_count=False):
        """itd_baseline_extract (MEITD.py:303-338): src -> (rot, base); with the extrema count of the produced baseline."""
        r = self.eng.spline_extract_dev(src, self.n, base, rot, 0, want_baseline_knots=want_baseline_count)
        knots = r[0] if want_baseline_count else r
        if knots < 2:
            raise TypeError("m > k must hold")        # what scipy.interpolate.splrep raises for fewer than 4 data sites
        return r[1] if want_baseline_count else None


_work = {}
_CACHE_MAX_BYTES = 64 << 20        # device rows kept between calls per device (50 rows x n x 8 B): larger ones are freed when the call returns
_lock = threading.RLock()          # the module-level API keeps one set of device rows per device: calls are serialised


def _work_for(n, device, solver="auto"):
    key = int(device)
    w = _work.get(key)
    if w is None or w.n != n or w.solver != solver or w.eng is not _eng(n, device, solver):
        if w is not None:
            w.buf.free()
        w = _work[key] = _Work(n, device, solver)
    w.eng.set_spline_solver({"auto": 0, "serial": 1, "parallel": 2}[solver])
    w.reset()
    return w


def release(device=None):
    """Free the device rows this module keeps between calls (all devices, or one)."""
    with _lock:
        for key in [k for k in _work if device is None or k == int(device)]:
            _work.pop(key).buf.free()


def _done(w):
    """after a call: rows of a long signal do not stay allocated (a 2^24-sample call holds 6.7 GB)"""
    if isinstance(w, _Work) and (6 + 2 * _ROWS_KEPT) * w.n * 8 > _CACHE_MAX_BYTES:
        release(w.device)


def _proper(wpe, WPEMAX):
    return wpe < WPEMAX and not wpe < 0.2             # MEITD.py:364 / :387


def _retrieve(wk, rot, WPEMAX):
    """retrieve_proper_rotation on the row `rot`: returns (row holding the result, proper)."""
    wpe, count = wk.probe(rot)
    if count > 5 and _proper(wpe, WPEMAX):            # the first extraction is the answer (:360-365)
        base, out = wk.take(), wk.take()
        wk.extract(rot, base, out)
        wk.give(base)
        wk.give(rot)
        return out, 1
    return rot, 0                                     # "I can't retrieve a proper rotation" / the loop that returns its input


def _determine(wk, src, rot, base, WPEMAX, probed=None):
    """determine_if_first_is_proper_rotation(src) into the rows rot / base (base None: not wanted).  Returns proper."""
    wpe, count = probed if probed is not None else wk.probe(src)
    if count < 5:
        wk.assign(rot, src)
        if base is not None:
            wk.zero(base)
        return 0
    tmp = base if base is not None else wk.take()
    wk.extract(src, tmp, rot)
    if base is None:
        wk.give(tmp)
    return 1 if _proper(wpe, WPEMAX) else 0


def retrieve_proper_rotation(x, WPEMAX, device=0, solver="auto"):
    """MEITD.py:344-368 — the first extraction's rotation if the entropy of the INPUT passes the test, else the input."""
    x = numpy.ascontiguousarray(numpy.asarray(x).astype(dtype=numpy.float64))
    with _lock:
        wk = _work_for(len(x), device, solver)
        rot = wk.take()
        wk.upload(x, rot)
        rot, proper = _retrieve(wk, rot, WPEMAX)
        out = wk.download(rot)[0] if proper else x
        wk.give(rot)
        _done(wk)
    return out, proper


def determine_if_first_is_proper_rotation(x, WPEMAX, device=0, solver="auto"):
    """MEITD.py:371-392 — one extraction; proper if the input's entropy lies in [0.2, WPEMAX)."""
    x = numpy.ascontiguousarray(numpy.asarray(x).astype(dtype=numpy.float64))
    with _lock:
        wk = _work_for(len(x), device, solver)
        src, rot, base = wk.take(), wk.take(), wk.take()
        wk.upload(x, src)
        proper = _determine(wk, src, rot, base, WPEMAX)
        r, b = wk.download(rot)[0], wk.download(base)[0]
        for p in (src, rot, base):
            wk.give(p)
        _done(wk)
    return r, b, proper


_ONE_LAUNCH_MAX = 8192      # itd_meitd_small_f64: one workgroup holds the signal's run


def _proper_rows(log, WPEMAX):
    """the threshold test of every logged probe, from entropies re-drawn with the reference's numpy expression (_entropy_from_bins)"""
    full = (log["c"] > 0).all(axis=1)
    wpe = numpy.empty(len(log))
    if full.any():                    # (all six patterns present: the same numpy operations row by row, in one call)
        w = log["w"][full]
        p = numpy.true_divide(w, w.sum(axis=1)[:, None])
        pe = -numpy.multiply(p, numpy.log2(p)).sum(axis=1)
        pe /= numpy.log2(factorial(3))
        wpe[full] = pe
    for i in numpy.flatnonzero(~full):
        wpe[i] = _entropy_from_bins(log["w"][i], log["c"][i], 3, True)
    return (wpe < WPEMAX) & ~(wpe < 0.2)


def _meitd_one_launch(wk, data, WPEMAX):
    """The loop as ONE launch (include/pyitd_hip.h: itd_meitd_small_f64; csrc/itd_meitd.hpp).  Returns "host" when the device did not
    deliver — a NaN, an extraction scipy would refuse, a threshold test that numpy's log2 and the device's draw differently: the
    host-driven loop below then runs the call and reproduces the reference's behaviour —, else what _meitd returns."""
    wk.upload(data, wk.row(5))
    res, log = wk.eng.meitd_small_dev(wk.buf.ptr, wk.n, WPEMAX)
    status = int(res[0])
    wk.last = {"one_launch": True, "status": status, "probes": int(res[4]), "extractions": int(res[5]), "turns": int(res[6]),
               "us": {k: int(v) / 100.0 for k, v in zip(("probes", "extractions", "counts", "copies", "probe_sums", "probe_entropy", "", "", "x_knots", "x_values", "x_rows", "x_forward",
                                                         "x_backward", "x_eval"), res[8:22]) if k}}
    if status >= 2:
        return "host"
    if len(log):
        wd = log["wpe"]
        if not numpy.array_equal(_proper_rows(log, WPEMAX), (wd < WPEMAX) & ~(wd < 0.2)):
            wk.last["status"] = -1
            return "host"
    if status == 1:
        return None
    x = wk.row(int(res[3]))
    wk.free_rows.remove(x)
    return int(res[1]), int(res[2]), x


def _meitd(wk, data, WPEMAX):
    """MEITD.py:395-534 on device rows.  Returns (n_high, n_low, row of the residual) or None for the early return of :411-413."""
    if getattr(wk, "one_launch", False):
        r = _meitd_one_launch(wk, data, WPEMAX)
        if not isinstance(r, str):
            return r
        wk.reset()
    x, rotation, baseline = wk.take(), wk.take(), wk.take()
    wk.upload(data, x)
    n_high = n_low = 0
    probed = wk.probe(x)
    proper = _determine(wk, x, rotation, baseline, WPEMAX, probed)
    changed, on_signal, digs = False, True, 1           # xchanged, HILO == 1, soft_reset
    count = probed[1]
    if count < 4:
        for p in (x, rotation, baseline):
            wk.give(p)
        return None
    while count > 5:
        if n_high + n_low > 20:
            break
        if proper == 0:      # not proper yet, but decomposable: go down its own baselines
            rotation, proper = _retrieve(wk, rotation, WPEMAX)
        if proper == 1:
            if on_signal:
                wk.assign(wk.kept(wk.high0, n_high), rotation)
                n_high += 1
            else:
                wk.assign(wk.kept(wk.low0, n_low), rotation)
                n_low += 1
            digs = 0
            wk.subtract(x, rotation, x)
            changed = True
        if changed and on_signal:
            count = wk.count(x)
            if count < 5:
                continue
            wk.extract(x, baseline)
            proper = _determine(wk, baseline, rotation, None, WPEMAX)
            changed, on_signal = False, False
            continue
        elif on_signal:
            proper = _determine(wk, baseline, rotation, None, WPEMAX)
            on_signal = False
            continue
        if changed and not on_signal:
            probed = wk.probe(x)
            count = probed[1]
            if count < 5:
                continue
            proper = _determine(wk, x, rotation, baseline, WPEMAX, probed)
            changed, on_signal = False, True
            continue
        if not changed and not on_signal:
            if digs == 0:
                wk.extract(x, baseline, rotation)
                digs = 1
            count = wk.count(baseline)
            if count < 5:
                continue
            for _ in range(digs):
                deeper = wk.take()
                count = wk.extract(baseline, deeper, rotation, want_baseline_count=True)
                wk.give(baseline)
                baseline = deeper
                if count < 5:
                    break
            digs += 1
            continue
    wk.give(rotation)
    wk.give(baseline)
    return n_high, n_low, x


def MEITD(data, max_iteration=40, WPEMAX=0.6, device=0, solver="auto"):
    """MEITD.py:395-534 — alternate between peeling a proper rotation off the signal (the "high" list) and off its baseline
    (the "low" list); dig further down the baselines when neither succeeds.  Returns (high[h, N], low[l, N], residual[N]).

    solver: how the spline baselines are solved (pyitd_amd.spline._eng).  "auto" (default) takes the parallel-in-knots form for signals
    of >= 1024 samples: baselines and entropies then equal scipy's / the reference's to rounding (~1e-14 of the signal's scale), not bit
    for bit — and the driver's decisions are hard thresholds on them (extrema counts of nearly flat baselines against 5, entropies
    against 0.2 / WPEMAX): a signal that sits on a threshold may select different components than MEITD.py.  "serial" (FITPACK's own
    sweep, bit-level against scipy) removes that difference at one GPU thread per extraction.  The module keeps one set of device rows
    per device between calls (freed above 64 MB; `release()`); calls are serialised by a lock."""
    data = numpy.ascontiguousarray(numpy.asarray(data).astype(dtype=numpy.float64))
    with _lock:
        wk = _work_for(len(data), device, solver)
        r = _meitd(wk, data, WPEMAX)
        if r is None:
            zero = numpy.zeros(len(data))
            return zero, zero, data
        n_high, n_low, x = r
        high = wk.download(wk.high0, n_high) if n_high else numpy.zeros((0, len(data)))
        low = wk.download(wk.low0, n_low) if n_low else numpy.zeros((0, len(data)))
        residual = wk.download(x)[0]
        wk.give(x)
        _done(wk)
    return high, low, residual


def XITD(data, device=0, solver="auto"):
    """MEITD.py:536-549 — MEITD's components and residual, ordered by their entropy (taken on the device rows).  solver: see MEITD."""
    data = numpy.ascontiguousarray(numpy.asarray(data).astype(dtype=numpy.float64))
    with _lock:
        wk = _work_for(len(data), device, solver)
        r = _meitd(wk, data, 0.6)                        # upstream passes its WPEMAX estimate where max_iteration goes (:541)
        if r is None:
            zero = numpy.zeros(len(data))
            rotations = numpy.vstack((numpy.vstack((zero, zero)), data))
            ent = [weighted_permutation_entropy(rotations[i, :], order=3, normalize=True, device=device) for i in range(3)]
            return rotations[numpy.argsort(ent), :]
        n_high, n_low, x = r
        rows = [wk.kept(wk.high0, k) for k in range(n_high)] + [wk.kept(wk.low0, k) for k in range(n_low)] + [x]
        ent = [wk.entropy(p) for p in rows]
        out = numpy.empty((len(rows), wk.n))
        for k, i in enumerate(numpy.argsort(ent)):
            out[k] = wk.download(rows[i])[0]
        wk.give(x)
        _done(wk)
    return out
