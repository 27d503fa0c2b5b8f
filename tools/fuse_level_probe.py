"""fused decomposition time by hand-over level and range, one signal of 2^k samples"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_ONLY
import bench
M = 7
dev = torch.device("cuda:0")
for k in [int(a) for a in sys.argv[1:]] or (20, 22, 23, 24):
    n = 1 << k
    x = bench.batch_signals_device(torch, dev, 0, 1, n)
    rows = torch.empty((1, M + 2, n), dtype=torch.float64, device=dev)
    for L0, rng in ((3, 64), (2, 64), (3, 32), (2, 32)):
        eng = pyitd_amd.Engine(n, 1, 0)
        eng.set_fuse_mode(FUSE_ONLY)
        eng.set_fuse_level(L0)
        eng.set_fuse_range(rng)
        try:
            for _ in range(20):
                eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
            eng.summary(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 100 * 1e6
            eng.summary(1)
            print("2^%d L0=%d range %d: %.1f us" % (k, L0, rng, t), flush=True)
        except pyitd_amd.ITDError as e:
            print("2^%d L0=%d range %d: refused %s" % (k, L0, rng, e), flush=True)
        eng.close()
