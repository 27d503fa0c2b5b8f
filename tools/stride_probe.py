"""Does the power-of-two row stride of a 2^24-sample decomposition (rows 128 MiB apart: the sample pass writes six of them at
once) cost anything?  ms per decomposition and ns per sample for lengths around 2^24.  usage (GPU box): python tools/stride_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
import bench

M = 7
dev = torch.device("cuda:0")
for n in ((1 << 24) - 65536, (1 << 24) - 1536, 1 << 24, (1 << 24) + 1536, (1 << 24) + 65536 + 2560, 3 * (1 << 22), 5 * (1 << 22) + 512):
    x = torch.from_numpy(bench.sines_noise(n)).to(dev)
    rows = torch.empty((M + 2, n), dtype=torch.float64, device=dev)
    eng = pyitd_amd.Engine(n, 1, 0)
    for _ in range(30):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
    eng.summary(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 100
    print("n = %9d (2^24 %+8d): %7.4f ms  %6.3f ps per sample  repeats %d" % (n, n - (1 << 24), dt * 1e3, dt / n * 1e12, eng.fuse_repeats), flush=True)
    eng.close()
    del x, rows
