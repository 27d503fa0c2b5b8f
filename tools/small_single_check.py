"""Latency of single short signals, repeated: is any size off the ~65 us launch-bound floor?  (GPU box)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402
M = 7
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
for n in (8000, 32768, 65535, 65536, 65537, 131072, 65536, 8000):
    t = torch.arange(n, dtype=torch.float64) / 8000.0
    x = (torch.sin(2 * np.pi * 110 * t)[None, :] + 0.05 * torch.randn((1, n), generator=g, dtype=torch.float64)).to(torch.float32).to(dev)
    rows = torch.empty((1, M + 2, n), dtype=torch.float64, device=dev)
    eng = pyitd_amd.Engine(n, 1, 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    out = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(50):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 50 * 1e6)
    s = eng.summary(1)
    print("n = %7d: %s us per decomposition; rows %d knots %s" % (n, ["%.1f" % v for v in out], s["n_rows"][0], [int(v) for v in s["knot_counts"][0] if v >= 0]), flush=True)
    eng.close()
