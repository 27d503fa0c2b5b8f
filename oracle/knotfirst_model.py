"""CPU model of the "knot-first" form of the sparse levels — TEST INFRASTRUCTURE ONLY (never imported by pyitd_amd).

Idea.  One extraction (ITD.py:79-121) maps every sample through an affine function of ITSELF,
    baseline[i] = B_k + S_k * (x[i] - x[e_k])          for i in [e_k, e_k+1),
so between two consecutive knots the baseline is a monotone image of the (monotone) input and the NEXT level's knots can only
sit at this level's knots — apart from positions where exact ties in the data let rounding make or break a plateau.  Hence the
whole level recursion ITD.py:384-432 can run on the knot list alone:
  * a candidate carries the level's values at (e-1, e, e+1);
  * B_k, S_k come from the knots' values and positions (ITD.py:100-116), the three values move to the next level through the
    affine maps of the segments they lie in, the knot predicate (ITD.py:59 on x and -x) on the new triple decides survival;
  * positions next to an exact tie of the level the recursion starts from (x[i] == x[i+1]) and sample n-2 (its right neighbour
    is forced to 0: baseline[n-1] is never written, ITD.py:112-117) stay candidates for ever ("sticky"), and so does every
    candidate whose triple shows an exact tie at some level.
The samples then need ONE pass over all fused levels (every sample through its segment's map at each level, rotation rows
written, nothing else read or written), which also re-derives every level's knots from the actual samples: any difference
from the knot side's list means the shortcut missed a knot and the result is discarded (the engine repeats level by level).

This file states that algorithm in numpy, exactly as the GPU runs it, so that tests can hold (a) the idea to the pinned
oracle (rows and per-level knot lists bit for bit, or an honest "verification failed"), and (b) the GPU's intermediate tables
to this model.
"""
import numpy as np


class NeedFallback(Exception):
    pass


def _predicate(yl, yc, yr):
    with np.errstate(invalid="ignore"):
        dp, dn = yc - yl, yr - yc
        return ((dn > 0) & (dp <= 0)) | ((dn < 0) & (dp >= 0))


NEAR = 2.0 ** -20


def near(a, b):
    """Two neighbouring values that rounding may make (or has made) equal within the levels to come: |a - b| <= 2^-20 max(|a|, |b|)."""
    with np.errstate(invalid="ignore"):
        return np.abs(a - b) <= NEAR * np.maximum(np.abs(a), np.abs(b))


def level_tables(n, pos, xc, ends):
    """Extended knot list, values, B and S of one level (ITD.py:93-116).  pos / xc: interior knots; ends = x[0], x[1], x[n-2], x[n-1]."""
    m = len(pos)
    e = np.concatenate(([0], pos, [n - 1])).astype(np.int64)
    X = np.concatenate(([ends[0]], xc, [ends[3]]))
    B = np.empty(m + 2)
    B[0] = (ends[0] + ends[1]) / 2.0
    B[m + 1] = (ends[2] + ends[3]) / 2.0
    if m:
        k = np.arange(1, m + 1)
        frac = (e[k] - e[k - 1]).astype(np.float64) / (e[k + 1] - e[k - 1]).astype(np.float64)
        B[k] = 0.5 * (X[k - 1] + frac * (X[k + 1] - X[k - 1])) + 0.5 * X[k]
    with np.errstate(all="ignore"):
        S = (B[1:] - B[:-1]) / (X[1:] - X[:-1])           # segments 0 .. m
    return e, X, B, S


def apply_map(n, e, X, B, S, i, xi):
    """baseline at positions i (values xi) through this level's segments; baseline[n-1] = 0."""
    k = np.searchsorted(e[1:-1], i, side="right")         # knots at or before i
    with np.errstate(all="ignore"):
        b = B[k] + S[k] * (xi - X[k])
    return np.where(i == n - 1, 0.0, b)


def knot_side(x_level, x0, n_extract, knots_fn):
    """The recursion on the knot list: `n_extract` extractions starting from the level whose input x_level is given in full.
    x0: the original signal (unused since round 4: the sticky candidates come from the exact ties of x_level).  Returns the per-level tables (dict: pos, e, X, B, S, ends) and the knot list of
    the last pending baseline.  Raises NeedFallback on non-finite knot data / too many ties."""
    n = len(x_level)
    # near ties of the level the recursion starts from — NOT the exact ties of the original signal: quantised input grows ties of its
    # own at the first levels (adjacent knots: B_k = x[k-1]/4 + x[k]/2 + x[k+1]/4 is exact on a grid and two neighbours can
    # coincide), a level may break a tie by one ulp and the next one restore it, and neighbours a few thousand ulps apart can
    # collapse a few levels on (differences shrink geometrically from level to level)
    x64 = np.asarray(x_level, dtype=np.float64)
    ties = np.flatnonzero(near(x64[:-1], x64[1:]))
    sticky = np.unique(np.concatenate((ties, ties + 1, [n - 2])))
    sticky = sticky[(sticky >= 1) & (sticky <= n - 2)]
    if len(sticky) > n // 16:      # (the GPU's limit is a capacity: a workgroup of 64 tiles holds 1024 candidates, knots included)
        raise NeedFallback("too many exact ties in the input (%d)" % len(ties))
    pos = np.asarray(knots_fn(x_level), dtype=np.int64)
    cand = np.unique(np.concatenate((pos, sticky)))
    is_knot = np.isin(cand, pos)
    tri = np.stack([x_level[cand - 1], x_level[cand], x_level[cand + 1]], axis=1)
    ends = np.array([x_level[0], x_level[1], x_level[n - 2], x_level[n - 1]])
    sticky_set = set(sticky.tolist())
    levels = []
    for _ in range(n_extract):
        kp, kx = cand[is_knot], tri[is_knot, 1]
        if not np.all(np.isfinite(tri)) or not np.all(np.isfinite(ends)):
            raise NeedFallback("non-finite knot data")
        e, X, B, S = level_tables(n, kp, kx, ends)
        if not (np.all(np.isfinite(B)) and np.all(np.isfinite(S))):
            raise NeedFallback("non-finite knot values / slopes")
        levels.append({"pos": kp, "e": e, "X": X, "B": B, "S": S, "ends": ends.copy()})
        # the three samples of every candidate, and the four end samples, through this level's maps
        idx = np.stack([cand - 1, cand, cand + 1], axis=1)
        new = apply_map(n, e, X, B, S, idx.ravel(), tri.ravel()).reshape(-1, 3)
        ends = apply_map(n, e, X, B, S, np.array([0, 1, n - 2, n - 1]), ends)
        flag = _predicate(new[:, 0], new[:, 1], new[:, 2])
        tie = near(new[:, 0], new[:, 1]) | near(new[:, 1], new[:, 2])
        sticky_set.update(cand[tie].tolist())
        keep = flag | np.isin(cand, np.fromiter(sticky_set, dtype=np.int64, count=len(sticky_set)))
        cand, tri, is_knot = cand[keep], new[keep], flag[keep]
    return levels, cand[is_knot]


def sample_pass(x_level, levels, next_knots):
    """Every sample through the fused levels; returns the list of baselines and the per-level knot lists re-derived from the
    samples themselves (the verification)."""
    n = len(x_level)
    i = np.arange(n)
    cur = np.asarray(x_level, dtype=np.float64)
    bases, found = [], []
    for L in levels:
        b = apply_map(n, L["e"], L["X"], L["B"], L["S"], i, cur)
        bases.append(b)
        f = np.zeros(n, bool)
        f[1:-1] = _predicate(b[:-2], b[1:-1], b[2:])
        found.append(np.flatnonzero(f))
        cur = b
    return bases, found
