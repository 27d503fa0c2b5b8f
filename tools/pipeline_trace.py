"""A short fused batch for a rocprofv3 --kernel-trace timeline of the batch pipeline.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d DIR -o trace -- python3 tools/pipeline_trace.py [batch] [pipeline 0/1] [calls] [streams] [signals per chunk]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 1
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 2
chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n, M = 1 << 20, 7
dev = torch.device("cuda:0")
x = bench.batch_signals_device(torch, dev, 0, batch, n)
rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
eng = pyitd_amd.Engine(n, batch, 0)
eng.set_fuse_mode(FUSE_AUTO)
eng.set_batch_pipeline(pipe)
eng.set_batch_streams(streams)
eng.set_batch_chunk(chunk)
for _ in range(calls):
    eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
    eng.summary(batch)
torch.cuda.synchronize()
print("done", eng.last_fuse_level, eng.fuse_repeats)
