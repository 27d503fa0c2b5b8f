"""MEITD's operators on device rows, one by one: us per call (entropy probe, extrema count, spline extraction, copies)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import meitd
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spline")
x = np.load(os.path.join(G, "meitd_walk.npz"))["x"]
wk = meitd._work_for(len(x), 0)
a, b, c = wk.take(), wk.take(), wk.take()
wk.upload(x, a)
def t(f, reps=200):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
print("probe (wpe + count)   %.1f us" % t(lambda: wk.probe(a)))
print("entropy (wpe only)    %.1f us" % t(lambda: wk.entropy(a)))
print("count                 %.1f us" % t(lambda: wk.count(a)))
print("extract               %.1f us" % t(lambda: wk.extract(a, b, c)))
print("extract + count       %.1f us" % t(lambda: wk.extract(a, b, c, want_baseline_count=True)))
print("assign (enqueue)      %.1f us" % t(lambda: wk.assign(b, a)))
print("subtract (enqueue)    %.1f us" % t(lambda: wk.subtract(a, c, b)))
print("download 1 row        %.1f us" % t(lambda: wk.download(a)))
print("_entropy_from_bins    %.1f us" % t(lambda: meitd._entropy_from_bins(np.ones(6), np.ones(6, np.int64), 3, True)))
