"""Where do the fused sparse levels start to pay?  One signal (or a small batch) of 2^k samples, 8 levels, fused against level by
level, device resident: ms per decomposition.  Below ~2^21 samples per launch sequence every launch is bound by its ~6.5 us
boundary and the fused form has more of them.  usage (GPU box): python tools/fuse_threshold.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_OFF, FUSE_ONLY
import bench

M = 7
dev = torch.device("cuda:0")
for batch, k in ((1, 16), (1, 17), (1, 18), (1, 19), (1, 20), (1, 21), (1, 22), (1, 23), (4, 16), (16, 16), (4, 18), (8, 18), (2, 20), (4, 20)):
    n = 1 << k
    x = bench.batch_signals_device(torch, dev, 0, batch, n)
    rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
    out = []
    for mode in (FUSE_OFF, FUSE_ONLY):
        eng = pyitd_amd.Engine(n, batch, 0)
        eng.set_fuse_mode(mode)
        try:
            for _ in range(20):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            eng.summary(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 200 if n * batch <= (1 << 21) else 50
            for _ in range(reps):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            torch.cuda.synchronize()
            out.append("%8.1f us" % ((time.perf_counter() - t0) / reps * 1e6))
            eng.summary(batch)
        except pyitd_amd.ITDError:
            out.append(" refused  ")
        eng.close()
    print("%3d x 2^%-2d  level by level %s   fused %s" % (batch, k, out[0], out[1]), flush=True)
