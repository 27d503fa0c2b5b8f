"""The single-level operators of the numpy API (pyitd_amd.itd: detect_peaks, matlab_detect_peaks, detect_knots, baseline_knot_estimation,
itd_baseline_extract, find_extrema, itd_baseline_extract_fast / _cubic) against the pinned CPU oracle on random signals of random lengths —
the open-ended form of the seeded cases in tests/test_gpu_parity.py and tests/test_gpu_cubic.py.  Indices and the tier-1 floats bit for bit,
the cubic operator to 1e-9 of the scale.  usage: python tools/ops_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal, assert_bits_equal
from oracle import cpu_oracle as O
import importlib
A = importlib.import_module("pyitd_amd.itd")

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(cases):
    kind = int(rng.integers(0, 8))
    n = int(rng.choice([3, 4, 5, 17, 64, 100, 511, 512, 513, 1000, 4096, 4097, 8191, 20000, 65536, 100001]))
    x = fuzz_signal(rng, kind, n)
    if case % 7 == 3:
        x = x.astype(np.float32)
    what = "case %d (family %d, n %d, %s)" % (case, kind, n, x.dtype)
    try:
        with np.errstate(all="ignore"):
            xs = np.asarray(x, dtype=np.float64)
            assert np.array_equal(A.detect_peaks(x), O.detect_peaks(xs)), "detect_peaks"
            assert np.array_equal(A.matlab_detect_peaks(x), O.detect_peaks(xs, matlab=True)), "matlab_detect_peaks"
            e = O.knots(xs)
            assert np.array_equal(A.detect_knots(x), e), "detect_knots"
            if n >= 3:
                rot, base = A.itd_baseline_extract(x)
                rr, bb = O.itd_baseline_extract(xs)
                assert_bits_equal(base, bb, "baseline")
                assert_bits_equal(rot, rr, "rotation")
            e1, i1 = O.find_extrema(xs)
            e2, i2 = A.find_extrema(xs)
            assert i1 == i2 and np.array_equal(e1, e2), "find_extrema"
            if np.isfinite(xs).all() and n >= 8:
                ec, idx = O.extrema_cpp(xs)
                if idx >= 4:
                    ref = O.itd_baseline_extract_fast(xs, ec, idx)
                    got = A.itd_baseline_extract_fast(xs, ec, idx)
                    if np.isfinite(ref).all():
                        scale = max(1.0, float(np.abs(ref).max()))
                        assert got.shape == ref.shape and np.isfinite(got).all() and np.abs(got - ref).max() <= 1e-9 * scale, "cubic: %.3e of scale %.3e" % (np.abs(got - ref).max(), scale)
    except AssertionError as ex:
        bad += 1
        print("MISMATCH " + what + ": " + str(ex)[:200])
    except Exception as ex:
        bad += 1
        print("ERROR " + what + ": %s %s" % (type(ex).__name__, str(ex)[:200]))
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
