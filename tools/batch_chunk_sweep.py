"""BASELINE configs[2] (batch x 2^20 signals, 8 levels) against the engine's batch geometry: signals per launch sequence
(itd_set_batch_chunk) x streams the sequences rotate over (itd_set_batch_streams), with the sparse levels fused (default) and not.
The summary is read every step (refusing signals are re-run inside the timed region).
usage (GPU box): python tools/batch_chunk_sweep.py [batch]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO, FUSE_OFF
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, M = 1 << 20, 7
dev = torch.device("cuda:0")
x = bench.batch_signals_device(torch, dev, 0, batch, n)
rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for mode, name in ((FUSE_AUTO, "fused"), (FUSE_OFF, "level by level")):
    for streams in ((2,) if os.environ.get("SWEEP_ONE") else (1, 2, 3)):
        line = []
        for chunk in ((12,) if os.environ.get("SWEEP_ONE") else (4, 6, 8, 12, 16, 24, 32, 48)):
            eng = pyitd_amd.Engine(n, batch, 0)
            eng.set_fuse_mode(mode)
            eng.set_batch_streams(streams)
            eng.set_batch_chunk(chunk)
            for _ in range(2):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
                eng.summary(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
                eng.summary(batch)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            line.append("%2d: %6.2f" % (chunk, dt * 1e3))
            eng.close()
        print("%-14s %d stream(s)  ms per %d signals by chunk  %s" % (name, streams, batch, "   ".join(line)), flush=True)
