"""BASELINE configs[2]'s recipe (batch x 2^20 signals, 8 levels) by first fused level x streams x signals per launch sequence,
the summary read every step.  usage (GPU box): python tools/batch_level_sweep.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, M = 1 << 20, 7
dev = torch.device("cuda:0")
x = bench.batch_signals_device(torch, dev, 0, batch, n)
rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for level in (3, 2):
    for streams in (1, 2, 3, 4):
        line = []
        for chunk in (4, 8, 12, 16):
            eng = pyitd_amd.Engine(n, batch, 0)
            eng.set_fuse_mode(FUSE_AUTO)
            eng.set_fuse_level(level)
            eng.set_batch_streams(streams)
            eng.set_batch_chunk(chunk)
            for _ in range(2):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
                eng.summary(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fix0 = eng.fuse_signal_repairs
            for _ in range(3):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
                eng.summary(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            line.append("%2d: %6.2f (%d re-run, %d repeats)" % (chunk, dt * 1e3, (eng.fuse_signal_repairs - fix0) // 3, eng.fuse_repeats))
            eng.close()
        print("first fused level %d, %d stream(s): ms per %d signals by chunk  %s" % (level, streams, batch, "   ".join(line)), flush=True)
