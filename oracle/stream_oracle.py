"""Block-wise (streaming) operation of the single-level baseline operators — TEST INFRASTRUCTURE ONLY (checker for
pyitd_amd's itd_stream_* entry points; never imported by the product).

The reference states the recipe in a comment only (itd.cpp:31-38, nothing implements it):

    :32  "use a circular buffer with modulous tracking to rotate the samples."
    :33  "re-assess extrema in the entire buffer every iteration"
    :34  "use from the last extrema in the first buffer to the first extrema in the last buffer"
    :35  "set the first and last baseline knots manually to said values"
    :36  "update the j array"
    :37  "compute only the baseline[i] array for the inner third of the buffer overall"
    :38  "rotate buffers, rinse and repeat"

and, for several channels / repeated passes (itd.cpp:40-44): "estimate the extrema the first time you process your data
with this method, and retain the extrema.  further iterations should reuse the extrema but evaluate and produce the
baseline on new data" — here: the knots of channel 0's window are reused for every channel's window (`shared_knots`).

This file states that recipe ONCE, as plain Python over operators that are passed in:
  * with oracle/cpu_oracle.py's operators it is the CPU oracle the `-m gpu` tests hold the device stream to;
  * oracle/gen_golden.py calls the same functions with the REFERENCE's own operators (itd_baseline_extract_fast,
    itd_fourier_decomposition.py:49-122; itd_baseline_extract, ITD.py:79-121) to write tests/golden/stream/*.npz, which
    tests/test_oracle_stream.py holds the oracle form to bit for bit.

Window geometry (three blocks of L samples, blocks numbered from 0; the block emitted is the middle one):
    block 0        window = blocks 0, 1       emitted part [0, L)     (no predecessor: the window starts with it)
    block j        window = blocks j-1 .. j+1 emitted part [L, 2L)
    last block     window = the last two      emitted part [L, 2L)    (no successor; one block in all: [0, L) of itself)

Cubic operator (kind "cubic", itd_baseline_extract_fast with externally chosen knots):
    knots   = itd.cpp:161-168's predicate over the whole window (:33)
    a, b    = first knot at or behind the emitted part's first sample / behind its last
    sel     = knots[max(a - margin, 0) : min(b + margin + 2, m)]
              margin extrema either side of the emitted part (:34 is margin = 1), and two more behind it: the operator never
              computes the value of its second-to-last knot (K[idx-1] = 0, itd_fourier_decomposition.py:61: range(1, idx-1)),
              so that knot is kept outside the emitted part
    fewer than 4 selected knots: the block is emitted unchanged (itd.cpp:170-172 "break early")
    baseline = itd_baseline_extract_fast(window, sel, len(sel) - 1)[emitted part] — the operator pins the first and the last
              knot value to the data there (:83 = itd.cpp:35)

Tier-1 operator (kind "linear", itd_baseline_extract of ITD.py:79-121 on the window): rotation and baseline of the emitted
part.  The operator is local — a sample's baseline depends on the two knots at or before it and the two behind it — so
wherever a window holds those four knots the emitted values are bit-identical to the whole-signal operator's
(tests/test_oracle_stream.py::test_linear_stream_equals_the_whole_signal).
"""
import numpy as np


def windows(n_blocks, L):
    """[(window first sample, window length, lo, hi)] for the blocks 0 .. n_blocks-1 of a stream of n_blocks * L samples."""
    out = []
    for j in range(n_blocks):
        if n_blocks == 1:
            out.append((0, L, 0, L))
        elif j == 0:
            out.append((0, 2 * L, 0, L))
        elif j == n_blocks - 1:
            out.append(((j - 1) * L, 2 * L, L, 2 * L))
        else:
            out.append(((j - 1) * L, 3 * L, L, 2 * L))
    return out


def select_knots(knots, lo, hi, margin):
    """The knots the spline is built on for the emitted part [lo, hi) of a window (see the module docstring)."""
    m = len(knots)
    a = int(np.searchsorted(knots, lo, side="left"))
    b = int(np.searchsorted(knots, hi, side="left"))
    return knots[max(a - margin, 0): min(b + margin + 2, m)]


def cubic_window(extract_fast, extrema_cpp, W, lo, hi, margin, knots=None):
    """Baseline of W[lo:hi].  `knots`: retained extrema of another channel's window (itd.cpp:40-44), else W's own."""
    if knots is None:
        e, m = extrema_cpp(W)
        knots = np.asarray(e[:m], dtype=np.int64)
    sel = select_knots(knots, lo, hi, margin)
    if len(sel) < 4:
        return np.array(W[lo:hi], dtype=np.float64)
    base = extract_fast(W, np.ascontiguousarray(sel, dtype=np.int64), len(sel) - 1)
    return np.asarray(base)[lo:hi].copy()


def blockwise_cubic(extract_fast, extrema_cpp, x, L, margin, shared_knots=False):
    """x[C, n_blocks * L] (or one channel [n]) -> baseline of the same shape, block by block."""
    x2 = np.atleast_2d(np.asarray(x, dtype=np.float64))
    C, n = x2.shape
    assert n % L == 0 and n >= L
    out = np.empty_like(x2)
    for j, (w0, wl, lo, hi) in enumerate(windows(n // L, L)):
        knots = None
        for c in range(C):
            W = np.ascontiguousarray(x2[c, w0:w0 + wl])
            if shared_knots and c == 0:
                e, m = extrema_cpp(W)
                knots = np.asarray(e[:m], dtype=np.int64)
            out[c, j * L:(j + 1) * L] = cubic_window(extract_fast, extrema_cpp, W, lo, hi, margin,
                                                     knots if shared_knots else None)
    return out if np.ndim(x) == 2 else out[0]


def blockwise_linear(extract, x, L):
    """x[C, n_blocks * L] (or [n]) -> (rotation, baseline), block by block with ITD.py:79-121 on every window."""
    x2 = np.atleast_2d(np.asarray(x, dtype=np.float64))
    C, n = x2.shape
    assert n % L == 0 and n >= L
    rot, base = np.empty_like(x2), np.empty_like(x2)
    for j, (w0, wl, lo, hi) in enumerate(windows(n // L, L)):
        for c in range(C):
            r, b = extract(np.array(x2[c, w0:w0 + wl], dtype=np.float64))   # a private copy: the reference mutates NaN -> +inf
            rot[c, j * L:(j + 1) * L] = np.asarray(r)[lo:hi]
            base[c, j * L:(j + 1) * L] = np.asarray(b)[lo:hi]
    if np.ndim(x) == 2:
        return rot, base
    return rot[0], base[0]


# ---- the oracle form: the recipe over oracle/cpu_oracle.py's pinned operators ---------------------------------------------
def oracle_blockwise_cubic(x, L, margin=8, shared_knots=False):
    from . import cpu_oracle
    return blockwise_cubic(cpu_oracle.itd_baseline_extract_fast, cpu_oracle.extrema_cpp, x, L, margin, shared_knots)


def oracle_blockwise_linear(x, L):
    from . import cpu_oracle
    return blockwise_linear(cpu_oracle.itd_baseline_extract, x, L)


def oracle_extract_fast_channels(x, extrema, idx):
    """The retained-extrema call of itd.cpp:40-44 without blocks: one knot list, every channel of x[C, n]."""
    from . import cpu_oracle
    x2 = np.atleast_2d(np.asarray(x, dtype=np.float64))
    return np.stack([cpu_oracle.itd_baseline_extract_fast(x2[c], extrema, idx) for c in range(x2.shape[0])])
