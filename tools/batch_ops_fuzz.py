"""The batched single-level operators (pyitd_amd.batch: itd_baseline_extract_batch, count_knots_batch, detect_knots_batch,
itd_baseline_extract_fast_channels) against the CPU oracle on random batches (1 .. 3000 rows of 3 .. 5000 samples, NaNs, plateaus, knot-free
rows): rows bit for bit, counts and lists exact, the cubic channels to 1e-9.  usage: python tools/batch_ops_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal, assert_bits_equal
from oracle import cpu_oracle as O
from pyitd_amd.batch import count_knots_batch, detect_knots_batch, itd_baseline_extract_batch, itd_baseline_extract_fast_channels

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(cases):
    B = int(rng.choice([1, 2, 5, 64, 300, 3000]))
    n = int(rng.choice([3, 8, 100, 511, 512, 513, 1000, 5000]))
    if B * n > 3_000_000:
        B = 300
    x = np.stack([fuzz_signal(rng, int(rng.integers(0, 8)), n) for _ in range(B)]).astype(np.float64)
    if case % 3 == 0:
        x[int(rng.integers(0, B)), int(rng.integers(0, n))] = np.nan
    if case % 4 == 1:
        x[int(rng.integers(0, B))] = np.linspace(0, 1, n)
    what = "case %d (%d x %d)" % (case, B, n)
    try:
        with np.errstate(all="ignore"):
            rot, base, counts = itd_baseline_extract_batch(x, want_counts=True)
            check = range(B) if B <= 64 else rng.integers(0, B, 48)
            for b in check:
                r, bs, kn, _ = O.itd_baseline_extract(x[b], want_knots=True)
                assert_bits_equal(base[b], bs, what + " row %d baseline" % b)
                assert_bits_equal(rot[b], r, what + " row %d rotation" % b)
                assert counts[b] == len(kn), what + " row %d count" % b
            sub = x[:min(B, 64)]
            if not np.isnan(sub).any():
                cnt = count_knots_batch(sub)
                lists = detect_knots_batch(sub)
                for b in range(sub.shape[0]):
                    k = O.knots(sub[b])
                    assert cnt[b] == len(k) and np.array_equal(lists[b], k), what + " row %d knots" % b
            if n >= 100 and np.isfinite(x).all() and B <= 64:
                got = itd_baseline_extract_fast_channels(x, None, 0)
                for c in range(B):
                    e, idx = O.extrema_cpp(x[c])
                    if idx < 2:
                        assert not got[c].any(), what + " channel %d untouched" % c
                        continue
                    ref = O.itd_baseline_extract_fast(x[c], e, idx)
                    if np.isfinite(ref).all():
                        scale = max(1.0, float(np.abs(ref).max()))
                        assert np.abs(got[c] - ref).max() <= 1e-9 * scale, what + " cubic channel %d" % c
    except AssertionError as ex:
        bad += 1
        print("MISMATCH " + str(ex)[:240])
    except Exception as ex:
        bad += 1
        print("ERROR %s: %s %s" % (what, type(ex).__name__, str(ex)[:200]))
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
