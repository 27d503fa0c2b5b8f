"""numpy restatement of the reference ITD path — TEST INFRASTRUCTURE / CPU-BASELINE LEG ONLY (never imported by pyitd_amd).

A vectorised form of ITD.py:33-121 + :384-432 that performs, element for element, the same IEEE binary64 operations in the
same association order as the reference's loops, so its results are bit-identical (tests/test_oracle_golden.py holds it
to the golden vectors):

  knot predicate   dx = x[1:] - x[:-1];  valley: dx[i] > 0 & dx[i-1] <= 0, peak: the same on -dx      (ITD.py:44-59, 87-88)
  knot list        e = [0, flagged indices, n-1]                                                       (ITD.py:93-98)
  knot values      B_k = 0.5*(x[e_{k-1}] + ((e_k-e_{k-1})/(e_{k+1}-e_{k-1}))*(x[e_{k+1}]-x[e_{k-1}])) + 0.5*x[e_k]   (:100-110)
  baseline map     segment id of sample i = number of knots at or before i (inclusive prefix count of the flags);
                   baseline[i] = B_k + ((B_{k+1}-B_k)/(x[e_{k+1}]-x[e_k]))*(x[i]-x[e_k]),  baseline[n-1] = 0    (:112-117)
  driver           stop rules and row packing of ITD.itd                                               (:384-432)

NaN handling: the reference's detect_peaks NaN branch (ITD.py:46-51, 64-68) is NOT restated here; inputs whose baselines
go NaN are outside this leg's domain (the C oracle covers them) and raise.
"""
import numpy as np

MAX_ROWS = 22


def knot_flags(x):
    """bool[n]: interior knots of x = detect_peaks(x) U detect_peaks(-x) (ITD.py:59, 87-98; first/last never, :70-73)."""
    dx = x[1:] - x[:-1]
    f = np.zeros(x.shape[0], dtype=bool)
    vil, vix = dx[1:], dx[:-1]
    f[1:-1] = ((vil > 0) & (vix <= 0)) | ((vil < 0) & (vix >= 0))
    return f


def baseline_extract(x):
    """(rotation, baseline, m) of one extraction (ITD.py:79-121)."""
    n = x.shape[0]
    if np.isnan(x).any():
        raise ValueError("numpy restatement: NaN input (the reference's NaN branch is restated by the C oracle only)")
    f = knot_flags(x)
    e = np.concatenate(([0], np.flatnonzero(f), [n - 1])).astype(np.int64)
    m = e.shape[0] - 2
    xe = x[e]
    bk = np.empty(m + 2)
    bk[0] = (x[0] + x[1]) / 2.0
    bk[-1] = (x[-2] + x[-1]) / 2.0
    if m:
        frac = (e[1:-1] - e[:-2]) / (e[2:] - e[:-2])            # int64 true division, like the reference
        bk[1:-1] = 0.5 * (xe[:-2] + frac * (xe[2:] - xe[:-2])) + 0.5 * xe[1:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        slope = (bk[1:] - bk[:-1]) / (xe[1:] - xe[:-1])         # one per segment (0/0 at flat end segments: kept)
        seg = np.cumsum(f)                                      # knots at or before the sample = its segment
        base = bk[seg] + slope[np.minimum(seg, m)] * (x - xe[seg])
    base[n - 1] = 0.0                                           # never written by the reference's half-open slices
    return x - base, base, m


def itd(data, max_iteration=11):
    """The driver (ITD.py:384-432): dict(rows, stop, knot_counts) like oracle.cpu_oracle.itd_lean."""
    x = np.asarray(data, dtype=np.float64)
    n = x.shape[0]
    if n < 3 or max_iteration < 0 or max_iteration > MAX_ROWS - 2:
        raise ValueError("bad arguments")
    rows = np.zeros((max_iteration + 2, n))
    prev = np.zeros(n)                       # baselines[counter-1] (python index -1 = the untouched zero row at 0)
    rot, base, m0 = baseline_extract(x)
    counts = [m0]
    counter = 0
    while True:
        if np.isnan(base).any():
            raise ValueError("numpy restatement: a baseline went NaN (domain of the C oracle)")
        num_extrema = int(knot_flags(base).sum())               # ITD.py:400-402
        if num_extrema < 2:                                     # :404-416
            rows[counter] = prev
            counter += 1
            stop = "natural"
            break
        if counter > max_iteration:                             # :418-426
            rows[counter] = rot + base
            counter += 1
            stop = "timeout"
            break
        rows[counter] = rot                                     # :428-432
        prev = base
        counts.append(num_extrema)
        rot, base, _ = baseline_extract(base)
        counter += 1
    return {"rows": rows[:counter], "stop": stop, "knot_counts": np.asarray(counts[:counter], dtype=np.int64)}
