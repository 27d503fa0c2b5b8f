"""Row-wise / channel-wise forms of the single-level operators (numpy in -> numpy out) over the asynchronous batched entry
points of the C ABI (itd_baseline_extract_batch_f64, itd_detect_batch_f64, itd_baseline_extract_cubic_batch_f64).

The reference applies its operators row by row under numba.prange (siftED2D.ipynb cell 1) and re-uses retained extrema along
channels (itd.cpp:40-44); here a whole batch is one launch sequence and no knot count crosses PCIe in between.  Device
memory comes from the C ABI's own allocator (engine.DeviceBuffer): no torch needed.
"""
import numpy

from .engine import DETECT_KNOTS, DeviceBuffer
from .itd import _engine_for


def _rows(x):
    x = numpy.ascontiguousarray(x, dtype=numpy.float64)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array [signals, samples]")
    return x


def itd_baseline_extract_batch(x, device=0, want_counts=False):
    """itd_baseline_extract (ITD.py:79-121) of every row of x[B, n]: (rotation[B, n], baseline[B, n]).  Rows that hold a NaN
    are re-run one by one through the single-signal operator, which follows detect_peaks' NaN branch (ITD.py:46-51)."""
    x = _rows(x)
    B, n = x.shape
    eng = _engine_for(n, device)
    d_x, d_rot, d_base, d_info = (DeviceBuffer(x.nbytes, device), DeviceBuffer(x.nbytes, device), DeviceBuffer(x.nbytes, device),
                                  DeviceBuffer(4 * B, device))
    try:
        d_x.upload(x)
        eng.extract_batch_dev(d_x.ptr, n, B, n, d_rot.ptr, n, d_base.ptr, n, d_info.ptr)
        info = d_info.download(numpy.empty(B, numpy.int32))      # itd_dev_copy synchronises
        rot, base = d_rot.download(numpy.empty_like(x)), d_base.download(numpy.empty_like(x))
    finally:
        for b in (d_x, d_rot, d_base, d_info):
            b.free()
    counts = numpy.where(info < 0, -1 - info, info).astype(numpy.int64)
    for b in numpy.flatnonzero(info < 0):
        r, bs, kn, _ = eng.baseline_extract_host(x[b], want_knots=True)
        rot[b], base[b], counts[b] = r, bs, len(kn)
    return (rot, base, counts) if want_counts else (rot, base)


def count_knots_batch(x, mode=DETECT_KNOTS, device=0):
    """Number of knots of every row of x[B, n] under predicate `mode` (engine.DETECT_*, 3 = itd.cpp:161-168, 4 = sign
    changes); no index list is built or copied.  Rows that hold a NaN are counted by the single-signal operator."""
    x = _rows(x)
    B, n = x.shape
    eng = _engine_for(n, device)
    d_x, d_info = DeviceBuffer(x.nbytes, device), DeviceBuffer(4 * B, device)
    try:
        d_x.upload(x)
        eng.detect_batch_dev(d_x.ptr, n, B, n, mode, None, 0, d_info.ptr)
        info = d_info.download(numpy.empty(B, numpy.int32))
    finally:
        d_x.free()
        d_info.free()
    out = info.astype(numpy.int64)
    for b in numpy.flatnonzero(info < 0):
        out[b] = len(eng.detect_host(x[b], mode)) if mode <= 2 else -1 - info[b]
    return out


def detect_knots_batch(x, mode=DETECT_KNOTS, device=0):
    """The ordered knot lists of every row of x[B, n] (list of int64 arrays)."""
    x = _rows(x)
    B, n = x.shape
    eng = _engine_for(n, device)
    stride = max(n - 2, 1)
    d_x, d_idx, d_info = DeviceBuffer(x.nbytes, device), DeviceBuffer(4 * B * stride, device), DeviceBuffer(4 * B, device)
    try:
        d_x.upload(x)
        eng.detect_batch_dev(d_x.ptr, n, B, n, mode, d_idx.ptr, stride, d_info.ptr)
        info = d_info.download(numpy.empty(B, numpy.int32))
        idx = d_idx.download(numpy.empty((B, stride), numpy.int32))
    finally:
        for b in (d_x, d_idx, d_info):
            b.free()
    out = []
    for b in range(B):
        if info[b] < 0 and mode <= 2:
            out.append(eng.detect_host(x[b], mode))
        else:
            m = info[b] if info[b] >= 0 else -1 - info[b]
            out.append(idx[b, :m].astype(numpy.int64))
    return out


def itd_baseline_extract_fast_channels(x, extrema_input, idx, device=0):
    """itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122) of every channel of x[C, n] on ONE retained knot list
    (itd.cpp:40-44: "simply estimate the extrema the first time ... retain the extrema ... reuse the extrema but evaluate and
    produce the baseline on new data").  extrema_input: idx + 1 entries; None: every channel's own extrema (itd.cpp:159-169).
    Returns baselines[C, n]; channels left without a spline (fewer than 2 knots) come back as zeros, like the reference's
    freshly allocated result."""
    x = _rows(x)
    C, n = x.shape
    eng = _engine_for(n, device)
    d_x, d_base, d_info = DeviceBuffer(x.nbytes, device), DeviceBuffer(x.nbytes, device), DeviceBuffer(4 * C, device)
    d_e = None
    try:
        d_x.upload(x)
        d_base.upload(numpy.zeros_like(x))
        if extrema_input is not None:
            e = numpy.ascontiguousarray(extrema_input, dtype=numpy.int64)
            if e.shape[0] < idx + 1:
                raise ValueError("extrema_input needs idx+1 entries")
            if e[: idx + 1].min() < 0 or e[: idx + 1].max() >= n:
                raise IndexError("extrema outside the signal")      # what the reference's indexing would raise
            e32 = e[: idx + 1].astype(numpy.int32)
            d_e = DeviceBuffer(e32.nbytes, device)
            d_e.upload(e32)
            eng.cubic_batch_dev(d_x.ptr, n, C, n, d_e.ptr, 0, int(idx), d_base.ptr, n, d_info.ptr)
        else:
            eng.cubic_batch_dev(d_x.ptr, n, C, n, None, 0, 0, d_base.ptr, n, d_info.ptr)
        info = d_info.download(numpy.empty(C, numpy.int32))
        if (info == -1).any():
            raise ValueError("extrema_input must be strictly increasing")
        if (info == -2).any():
            raise ValueError("NaN in the signal (the cubic operator has no NaN branch)")
        return d_base.download(numpy.empty_like(x))
    finally:
        for b in (d_x, d_base, d_info, d_e):
            if b is not None:
                b.free()
