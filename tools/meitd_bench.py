"""MEITD / XITD (MEITD.py:395-549) wall time on the three golden signals: pyitd_amd.meitd on the GPU operators (the arrays stay on
the device) against the same control flow over the CPU oracle's operators (oracle/meitd_oracle.CpuWork: scipy's splrep called as the
reference calls it, numpy knot counts and entropy) — a CPU restatement for scale, not the reference itself."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd import meitd
from oracle import meitd_oracle

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spline")

for name in sorted(f for f in os.listdir(G) if f.startswith("meitd_")):
    x = np.load(os.path.join(G, name))["x"]
    meitd.MEITD(x.copy())                      # warm-up (workspaces)
    t_gpu = t_gpu_x = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); hi, lo, res = meitd.MEITD(x.copy()); t_gpu = min(t_gpu, time.perf_counter() - t0)
        t0 = time.perf_counter(); xi = meitd.XITD(x.copy()); t_gpu_x = min(t_gpu_x, time.perf_counter() - t0)
    wk = meitd._work_for(len(x), 0)
    how = dict(wk.last)
    t_host = 1e9
    if wk.one_launch:                           # the same call driven from the host: one launch per operator
        wk.one_launch = False
        try:
            for _ in range(5):
                t0 = time.perf_counter(); meitd.MEITD(x.copy()); t_host = min(t_host, time.perf_counter() - t0)
        finally:
            wk.one_launch = True
    saved = meitd._work_for
    cw = []
    meitd._work_for = lambda n, device=0, solver="auto": (cw.append(meitd_oracle.CpuWork(n)), cw[-1])[1]
    try:
        t0 = time.perf_counter(); hi2, lo2, res2 = meitd.MEITD(x.copy()); t_cpu = time.perf_counter() - t0
    finally:
        meitd._work_for = saved
    same = hi.shape == hi2.shape and lo.shape == lo2.shape and np.max(np.abs(res - res2)) < 1e-9
    print("%-24s %d samples: MEITD %.2f ms on the GPU (XITD %.2f ms; %s), %.2f ms with one launch per operator, %.1f ms over the CPU "
          "restatement's operators; %d + %d components, %d extractions, %d entropy probes, same decisions: %s"
          % (name[:-4], len(x), t_gpu * 1e3, t_gpu_x * 1e3, "the loop as one launch, status %d, in it %s us" % (how["status"], how["us"]) if how.get("one_launch") else
             "one launch per operator", t_host * 1e3, t_cpu * 1e3, len(hi), len(lo), cw[-1].calls["extract"], cw[-1].calls["probe"], same))
