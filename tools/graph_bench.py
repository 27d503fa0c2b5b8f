"""Small signals are launch bound (10 dependent launches per decomposition): direct launches against a captured hipGraph replay.
usage (GPU box): python tools/graph_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402
from tests.helpers import sines_noise  # noqa: E402

M = 7
for lg in (12, 14, 16, 18, 20, 22):
    n = 1 << lg
    eng = pyitd_amd.Engine(n, 1, 0)
    x = torch.from_numpy(sines_noise(n, seed=1)).cuda()
    rows = torch.zeros((M + 2, n), dtype=torch.float64, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(50):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, s.cuda_stream)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, s.cuda_stream)
        s.synchronize()
        direct = (time.perf_counter() - t0) / 300
    eng.summary(1)
    ref = rows.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 300
    same = bool(torch.equal(rows.view(torch.int64), ref.view(torch.int64)))
    print("2^%-2d samples: direct %.1f us per decomposition, graph replay %.1f us (rows identical: %s)" % (lg, direct * 1e6, graph * 1e6, same))
    eng.close()
