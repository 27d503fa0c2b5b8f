"""No performance cliff when knots are extremely sparse (every tile has to search far for its neighbours' knots)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pyitd_amd
n = 1 << 24
t = np.arange(n, dtype=np.float64) / n
for cycles in (2.5, 300.0, 30000.0):
    x = torch.from_numpy((np.sin(2 * np.pi * cycles * t) + 0.3 * t * t).astype(np.float32)).cuda()
    rows = torch.empty((9, n), dtype=torch.float64, device="cuda")
    eng = pyitd_amd.Engine(n, 1, 0)
    torch.cuda.synchronize()
    for _ in range(2):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, 7, rows.data_ptr(), None, None)
    s = eng.summary(1)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, 7, rows.data_ptr(), None, None)
    s = eng.summary(1)
    dt = (time.perf_counter() - t0) / 5
    print("cycles %-8g rows %d knots/level %s : %.3f ms per decomposition" % (
        cycles, int(s["n_rows"][0]), [int(v) for v in s["knot_counts"][0] if v >= 0], dt * 1e3))
    eng.close()
