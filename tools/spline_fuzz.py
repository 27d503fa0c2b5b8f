"""The FITPACK flavour of the baseline (pyitd_amd.spline: itd_baseline_extract_rows with the serial and the parallel-in-knots solver,
crossways_itd_baseline_extract) against the scipy-backed oracle (oracle/spline_oracle.py: splrep called as the reference calls it) on random
images / batches of rows: the serial solver to 1e-10 of the scale (bit-level against scipy on nearly every value), the parallel-in-knots
solver to its 1e-9.
usage: python tools/spline_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal
from oracle import spline_oracle as SO
from pyitd_amd import spline as S

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()


def close(got, ref, what, tol=1e-10):
    scale = max(1.0, float(np.nanmax(np.abs(ref))))
    assert got.shape == ref.shape, what + ": shape"
    assert np.array_equal(np.isnan(got), np.isnan(ref)), what + ": NaN pattern"
    err = float(np.nanmax(np.abs(got - ref))) if got.size else 0.0
    assert err <= tol * scale, "%s: max |diff| %.3e of scale %.3e" % (what, err, scale)


for case in range(cases):
    rows = int(rng.choice([1, 2, 7, 64, 300]))
    n = int(rng.choice([16, 50, 128, 512, 513, 1000, 3000]))
    kinds = [int(rng.integers(0, 5)) for _ in range(rows)]
    x = np.stack([fuzz_signal(rng, k, n) for k in kinds]).astype(np.float64)
    x += 1e-6 * rng.standard_normal(x.shape)                      # (exact plateaus give scipy duplicate sites: the reference raises there)
    min_extrema = int(rng.choice([0, 2, 10]))
    what = "case %d (%d rows x %d, min_extrema %d)" % (case, rows, n, min_extrema)
    try:
        with np.errstate(all="ignore"):
            ref = np.stack([SO.baseline(r, max(min_extrema, 2)) for r in x])
            for solver in ("serial", "parallel"):
                got = S.itd_baseline_extract_rows(x, max(min_extrema, 2), solver=solver)
                # (the moment form on a row of three knots 1 and 1507 samples apart — an interpolant that overshoots a hundredfold —
                #  differs from FITPACK by 1.0e-10 of the scale, with round 4's solver constants and divisions just the same: its bound is 1e-9)
                close(got, ref, what + " rows, " + solver, 1e-10 if solver == "serial" else 1e-9)
            if case % 10 == 0 and rows >= 7 and n <= 512:
                img = x[:min(rows, 64), :min(n, 128)]
                close(S.crossways_itd_baseline_extract(img), SO.crossways(img), what + " crossways")
    except AssertionError as ex:
        bad += 1
        print("MISMATCH " + str(ex)[:240])
    except Exception as ex:
        bad += 1
        print("ERROR %s: %s %s" % (what, type(ex).__name__, str(ex)[:200]))
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
