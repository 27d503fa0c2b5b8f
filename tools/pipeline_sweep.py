"""The batch pipeline (itd_set_batch_pipeline) on BASELINE configs[2]'s recipe: batch x 2^20 samples, 8 levels, the summary read every
step — pipelined against rotating chunks, by signals per chunk; the rows of both forms are compared bit for bit.
usage (GPU box): python tools/pipeline_sweep.py [batch] [log2n] [chunks,comma,separated]
(PYITD_PIPE_GATE_US = the gate's time-out in microseconds: read by the library at engine creation)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
chunks = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 4, 8, 16]
n, M = 1 << log2n, 7
dev = torch.device("cuda:0")
x = bench.batch_signals_device(torch, dev, 0, batch, n)
rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
ref = None
torch.cuda.synchronize()


def digest(t):
    v = t.view(torch.int64)
    return int(v.sum().item()), int((v[:, :, ::4097] ^ (v[:, :, 1::4097][:, :, :v[:, :, ::4097].shape[2]])).sum().item())


for pipe in (0, 1, 0, 1):
    line = []
    for chunk in chunks:
        eng = pyitd_amd.Engine(n, batch, 0)
        eng.set_fuse_mode(FUSE_AUTO)
        eng.set_batch_pipeline(pipe)
        eng.set_batch_chunk(chunk)
        for _ in range(2):
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            eng.summary(batch)
        torch.cuda.synchronize()
        fix0 = eng.fuse_signal_repairs
        t0 = time.perf_counter()
        for _ in range(4):
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            eng.summary(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        d = digest(rows)
        if ref is None:
            ref = d
        line.append("%2d: %6.2f ms%s (%d re-run, %d repeats, level %d)" % (chunk, dt * 1e3, "" if d == ref else " ROWS DIFFER", (eng.fuse_signal_repairs - fix0) // 4,
                                                                          eng.fuse_repeats, eng.last_fuse_level))
        eng.close()
    print("pipeline %d: %d x 2^%d by signals per chunk  %s" % (pipe, batch, log2n, "   ".join(line)), flush=True)
