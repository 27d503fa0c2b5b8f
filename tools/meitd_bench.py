"""MEITD / XITD (MEITD.py:395-549) wall time on the three golden signals: pyitd_amd.meitd on the GPU operators against the same
control flow over the CPU oracle's operators (oracle/spline_oracle.py: scipy's splrep called as the reference calls it; numpy
knot counts) — a CPU restatement for scale, not the reference itself."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import meitd
from oracle import spline_oracle, cpu_oracle

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spline")


def cpu_extract(x, device=0):
    x = np.asarray(x, dtype=np.float64)
    b = spline_oracle.baseline(x, 0)
    return x - b, b


def cpu_count(x, device=0):
    return len(cpu_oracle.knots(np.asarray(x, dtype=np.float64)))


def cpu_extract_count(x, device=0):
    r, b = cpu_extract(x)
    return r, b, cpu_count(b)


for name in sorted(f for f in os.listdir(G) if f.startswith("meitd_")):
    x = np.load(os.path.join(G, name))["x"]
    meitd.MEITD(x.copy())                      # warm-up (workspaces)
    t0 = time.perf_counter(); hi, lo, res = meitd.MEITD(x.copy()); t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter(); xi = meitd.XITD(x.copy()); t_gpu_x = time.perf_counter() - t0
    saved = (meitd.itd_baseline_extract_spline, meitd._num_extrema, meitd._extract_and_count)
    meitd.itd_baseline_extract_spline, meitd._num_extrema, meitd._extract_and_count = cpu_extract, cpu_count, cpu_extract_count
    try:
        t0 = time.perf_counter(); hi2, lo2, res2 = meitd.MEITD(x.copy()); t_cpu = time.perf_counter() - t0
    finally:
        meitd.itd_baseline_extract_spline, meitd._num_extrema, meitd._extract_and_count = saved
    same = hi.shape == hi2.shape and lo.shape == lo2.shape and np.max(np.abs(res - res2)) < 1e-9
    print("%-24s %d samples: MEITD %.1f ms on the GPU operators (XITD %.1f ms), %.1f ms over the CPU restatement's operators; "
          "%d + %d components, same decisions: %s" % (name[:-4], len(x), t_gpu * 1e3, t_gpu_x * 1e3, t_cpu * 1e3, len(hi), len(lo), same))
