"""The FITPACK flavour of the baseline on ONE long signal (numba_accelerated_itd.py:182-211 is a 1-D operator): the serial
bit-level sweep (one GPU lane), the parallel moment form (itd_nak.hpp) and scipy's splrep-based CPU oracle, 2^17 and 2^20 samples."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import spline
from oracle import spline_oracle

for log2n in (12, 17, 20):
    n = 1 << log2n
    x = np.cumsum(np.random.default_rng(6).standard_normal(n))
    t0 = time.perf_counter(); ref = spline_oracle.baseline(x, 10); t_cpu = time.perf_counter() - t0
    res = {}
    for solver in ("parallel", "serial"):
        if solver == "serial" and log2n > 17:
            continue
        spline.itd_baseline_extract_modified(x, solver=solver)
        t0 = time.perf_counter(); got = spline.itd_baseline_extract_modified(x, solver=solver); dt = time.perf_counter() - t0
        res[solver] = (dt, float(np.max(np.abs(got - ref))))
    print("2^%d samples: scipy on the host %.2f ms; %s (host arrays in and out; scale %.1f)" % (
        log2n, t_cpu * 1e3, "; ".join("%s %.2f ms, max |diff| %.2e" % (k, v[0] * 1e3, v[1]) for k, v in res.items()), np.max(np.abs(ref))))
