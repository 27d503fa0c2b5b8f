"""The ABI's size limit: one signal of 2^31 - 2 float32 samples (int32 knot indices, 32-bit tile-relative arithmetic in the kernels).
One extraction + the "Out of time!" row (max_iteration = 0) on the GPU; the level is local, so the last ~10^6 samples of both rows
must equal, bit for bit, what the CPU oracle computes on a slice that ends at the signal's end (away from the slice's own start),
and likewise a slice at the front and one across 2^30.  usage: python tools/big_n_check.py [n]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd.engine import Engine  # noqa: E402
from oracle import cpu_oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 31) - 2
M = 0
torch.manual_seed(5)
x = torch.empty(n, dtype=torch.float32, device="cuda")
step = 1 << 28
for a in range(0, n, step):            # sines + noise, generated in pieces
    b = min(n, a + step)
    t = torch.arange(a, b, device="cuda", dtype=torch.float64)
    x[a:b] = (torch.sin(t / 37.0) + 0.5 * torch.sin(t / 911.0)).to(torch.float32) + 0.05 * torch.randn(b - a, device="cuda")
    del t
rows = torch.empty((M + 2, n), dtype=torch.float64, device="cuda")
eng = Engine(n, 1)
print("n = %d, workspace %.1f GB, rows %.1f GB" % (n, eng.workspace_bytes / 1e9, rows.numel() * 8 / 1e9), flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
s = eng.summary(1)
dt = time.perf_counter() - t0
print("decomposed in %.1f ms: rows %d, stop %d, knots per level %s" % (dt * 1e3, s["n_rows"][0], s["stop"][0], [int(v) for v in s["knot_counts"][0] if v >= 0]), flush=True)
ok = True
W, G = 1 << 20, 4096                    # slice width, guard at the slice's artificial ends
for name, lo in (("front", 0), ("across 2^30", (1 << 30) - W // 2), ("across 2^31 - 2^20", n - W - (1 << 20)), ("end", n - W)):
    lo = max(0, min(lo, n - W))
    xs = x[lo:lo + W].cpu().numpy()
    ref = cpu_oracle.itd_lean(xs, M)
    a = G if lo > 0 else 0               # the slice's own first / last knots differ from the full signal's: compare inside
    b = W - G if lo + W < n else W
    for r in range(M + 2):
        got = rows[r, lo + a:lo + b].cpu().numpy()
        want = ref["rows"][r, a:b]
        same = np.array_equal(got.view(np.uint64), want.view(np.uint64))
        ok &= same
        print("%-20s row %d samples [%d, %d): %s" % (name, r, lo + a, lo + b, "bit-exact" if same else "DIFFERENT (%d values)" % int((got.view(np.uint64) != want.view(np.uint64)).sum())), flush=True)
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
