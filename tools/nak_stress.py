"""parallel-in-knots solver against the serial (FITPACK) one on knot spacings chosen to stress the warm-up bound"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import spline

def zigzag(n, pos, rng, amp):
    """a signal whose extrema sit at `pos` (alternating), linear in between, with amplitudes amp[k]"""
    x = np.zeros(n)
    p = np.concatenate(([0], pos, [n - 1]))
    v = np.concatenate(([0.0], amp * (-1.0) ** np.arange(len(pos)), [0.0]))
    return np.interp(np.arange(n), p, v)

rng = np.random.default_rng(5)
worst = 0.0
for case in range(60):
    n = int(rng.choice([3000, 8000, 8192, 20000, 100000]))
    kind = case % 4
    if kind == 0:      # geometric spacings shrinking then growing, repeated
        sp = []
        while sum(sp) < n - 10:
            top = int(rng.integers(6, 13))
            run = [2 ** k for k in range(top, -1, -1)] + [2 ** k for k in range(0, top + 1)]
            sp += run
        sp = np.array(sp)
    elif kind == 1:    # random spacings over five orders of magnitude
        sp = np.maximum(1, (10.0 ** rng.uniform(0, 3.5, n // 8)).astype(int))
    elif kind == 2:    # long stretch, then dense knots
        sp = np.concatenate(([n // 3], np.ones(200, int), [n // 4], np.ones(300, int) * 2, [max(1, n // 5)]))
    else:              # dense everywhere
        sp = rng.integers(1, 4, n)
    pos = np.cumsum(sp)
    pos = pos[pos < n - 2]
    amp = 10.0 ** rng.uniform(-3, 3, len(pos)) if kind != 3 else rng.uniform(0.5, 1.5, len(pos))
    x = zigzag(n, pos, rng, amp) + (1e-9 * rng.standard_normal(n) if kind == 2 else 0.0)
    try:
        rp, bp = spline.itd_baseline_extract_spline(x, solver="parallel")
        rs, bs = spline.itd_baseline_extract_spline(x, solver="serial")
    except TypeError:
        continue
    scale = np.abs(bs).max()
    err = np.abs(bp - bs).max() / scale
    worst = max(worst, err)
    print("case %2d kind %d n %6d knots %6d  max |parallel - serial| / scale = %.2e" % (case, kind, n, len(pos), err))
print("worst %.2e" % worst)
