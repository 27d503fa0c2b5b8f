"""Host-side mirror of MEITD.py's selection drivers, on top of the GPU operators:

  weighted_permutation_entropy(time_series, order=3, normalize=False)   MEITD.py:79-128   (host numpy, as upstream)
  retrieve_proper_rotation(x, WPEMAX)                                    MEITD.py:344-368
  determine_if_first_is_proper_rotation(x, WPEMAX)                       MEITD.py:371-392
  MEITD(data, max_iteration=40, WPEMAX=0.6) -> (high, low, residual)     MEITD.py:395-534
  XITD(data)                                                             MEITD.py:536-549

The drivers are control logic around two operators, both of which run on the GPU: the cubic-spline baseline extraction
(MEITD.py:303-338 -> pyitd_amd.itd_baseline_extract_spline) and the extrema count (matlab_detect_peaks(x).size +
matlab_detect_peaks(-x).size -> the engine's knot scan).  The entropy is a few numpy calls on the host upstream and stays
that: its sums are taken in the reference's order so that the threshold tests decide identically.
Upstream quirks kept: `max_iteration` is never used by MEITD (:395); XITD passes its WPEMAX estimate in that position
(:541), so MEITD runs with WPEMAX = 0.6 there; nothing is printed.
"""
from math import factorial

import numpy

from .itd import _engine_for
from .spline import itd_baseline_extract_spline


def _embed(x, order=3, delay=1):
    """MEITD.py:48-70"""
    N = len(x)
    Y = numpy.empty((order, N - (order - 1) * delay))
    for i in range(order):
        Y[i] = x[i * delay:i * delay + Y.shape[1]]
    return Y.T


def weighted_permutation_entropy(time_series, order=3, normalize=False):
    """MEITD.py:79-128: permutation patterns of the embedded series, each window weighted by its variance."""
    x = numpy.array(time_series)
    hashmult = numpy.power(order, numpy.arange(order))
    sorted_idx = _embed(x, order=order).argsort(kind="quicksort")
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


def _num_extrema(x, device):
    """matlab_detect_peaks(x).size + matlab_detect_peaks(-x).size: the knot count of x — counted on the GPU, four bytes come
    back (no index list is built or copied)."""
    return int(_engine_for(len(x), device).count_knots_host(x)[0])


def _extract_and_count(x, device):
    """itd_baseline_extract(x) (MEITD.py:303-338) and the extrema count of its baseline in ONE engine call: the loops of
    MEITD.py:362-363 and :497-505 ask for that count right after every extraction."""
    from .spline import _eng
    x = numpy.asarray(x, dtype=numpy.float64)
    base, rot, knots, bk = _eng(len(x), device).spline_extract_host(x[None, :], 0, want_rotation=True, want_baseline_knots=True)
    if knots[0] < 2:
        raise TypeError("m > k must hold")        # what scipy.interpolate.splrep raises for fewer than 4 data sites
    return rot[0], base[0], int(bk[0])


def retrieve_proper_rotation(x, WPEMAX, device=0):
    """MEITD.py:344-368 — keep extracting from the baseline until fewer than 6 extrema remain; the entropy of the INPUT decides."""
    x = numpy.asarray(x).astype(dtype=numpy.float64)
    wpe = numpy.mean(weighted_permutation_entropy(x, order=3, normalize=True))
    accept = wpe < WPEMAX and not wpe < 0.2
    count = _num_extrema(x, device)
    if count < 5:
        return x, 0
    rotation = numpy.zeros(len(x))
    baseline = x.copy()
    while count > 5:
        rotation, baseline, count = _extract_and_count(baseline, device)
        if accept:
            return rotation, 1
    return x, 0


def determine_if_first_is_proper_rotation(x, WPEMAX, device=0):
    """MEITD.py:371-392 — one extraction; proper if the input's entropy lies in [0.2, WPEMAX)."""
    x = numpy.asarray(x).astype(dtype=numpy.float64)
    wpe = numpy.mean(weighted_permutation_entropy(x, order=3, normalize=True))
    if _num_extrema(x, device) < 5:
        return x, numpy.zeros(len(x)), 0
    rotation, baseline = itd_baseline_extract_spline(x, device)
    return rotation, baseline, 1 if (wpe < WPEMAX and not wpe < 0.2) else 0


def MEITD(data, max_iteration=40, WPEMAX=0.6, device=0):
    """MEITD.py:395-534 — alternate between peeling a proper rotation off the signal (the "high" list) and off its baseline
    (the "low" list); dig further down the baselines when neither succeeds.  Returns (high[h, N], low[l, N], residual[N])."""
    x = numpy.asarray(data).astype(dtype=numpy.float64)
    n = len(x)
    high, low = numpy.zeros((44, n)), numpy.zeros((44, n))
    n_high = n_low = 0
    rotation, baseline = numpy.zeros(n), numpy.zeros(n)
    rotation[:], baseline[:], proper = determine_if_first_is_proper_rotation(x, WPEMAX, device)
    changed, on_signal, digs = False, True, 1           # xchanged, HILO == 1, soft_reset
    count = _num_extrema(x, device)
    if count < 4:
        zero = numpy.zeros(n)
        return zero, zero, x
    while count > 5:
        if n_high + n_low > 20:
            return high[:n_high], low[:n_low], x[:]
        if proper == 0:      # not proper yet, but decomposable: go down its own baselines
            rotation[:], proper = retrieve_proper_rotation(rotation[:], WPEMAX, device)
        if proper == 1:
            if on_signal:
                high[n_high] = rotation
                n_high += 1
            else:
                low[n_low] = rotation
                n_low += 1
            digs = 0
            x = x - rotation
            changed = True
        if changed and on_signal:
            count = _num_extrema(x, device)
            if count < 5:
                continue
            _, baseline[:] = itd_baseline_extract_spline(x, device)
            rotation[:], _, proper = determine_if_first_is_proper_rotation(baseline[:], WPEMAX, device)
            changed, on_signal = False, False
            continue
        elif on_signal:
            rotation[:], _, proper = determine_if_first_is_proper_rotation(baseline[:], WPEMAX, device)
            on_signal = False
            continue
        if changed and not on_signal:
            count = _num_extrema(x, device)
            if count < 5:
                continue
            rotation[:], baseline[:], proper = determine_if_first_is_proper_rotation(x, WPEMAX, device)
            changed, on_signal = False, True
            continue
        if not changed and not on_signal:
            if digs == 0:
                rotation[:], baseline[:] = itd_baseline_extract_spline(x, device)
                digs = 1
            count = _num_extrema(baseline, device)
            if count < 5:
                continue
            for _ in range(digs):
                rotation[:], baseline[:], count = _extract_and_count(baseline[:], device)
                if count < 5:
                    break
            digs += 1
            continue
    return high[:n_high], low[:n_low], x[:]


def XITD(data, device=0):
    """MEITD.py:536-549 — MEITD's components and residual, ordered by their entropy."""
    data = numpy.asarray(data).astype(dtype=numpy.float64)
    m_ = data.mean(axis=0)
    sd_ = data.std(axis=0, ddof=0)
    with numpy.errstate(all="ignore"):
        wpemax = numpy.log(abs(20 * numpy.log10(abs(numpy.where(sd_ == 0, 0, m_ / sd_)))))
    high, low, residual = MEITD(data, wpemax, device=device)     # upstream passes it where max_iteration goes (:541)
    rotations = numpy.vstack((high, low))
    rotations = numpy.vstack((rotations, residual))
    ent = [weighted_permutation_entropy(rotations[i, :], order=3, normalize=True) for i in range(rotations.shape[0])]
    return rotations[numpy.argsort(ent), :]
