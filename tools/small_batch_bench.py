"""Batches of SHORT signals (the reference's own demo sizes: 400-sample chirp, 8000-sample audio clip, image rows) through the
batched engine: ms per batch, Gsamples/s and the fraction of the HBM peak.  The bytes are those of the RESULT each signal really
has — its input once (4 B/sample) and every row it produced once (8 B/sample and row; short signals stop after 3 .. 9 rows) —
the least any form can move, so a fraction cannot exceed 1; the level-by-level form moves more (20 + 24 B per sample and level).
usage (GPU box): python tools/small_batch_bench.py [max_iteration]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 7
SHAPES = [(60000, 256), (40000, 400), (16384, 1024), (4096, 4096), (2048, 8000), (1024, 16384), (256, 65536), (16, 1 << 20),
          (1, 400), (1, 8000), (1, 65536)]
if os.environ.get("SMALL_RESIDENT_SHAPES"):   # the sizes around the resident form's classes (pyitd_amd/csrc/itd_resident.hpp)
    SHAPES = [(60000, 256), (32768, 512), (16384, 1024), (8192, 2048), (4096, 4096), (2048, 8000), (256, 1024), (64, 4096), (8, 4096),
              (1, 256), (1, 512), (1, 1024), (1, 2048), (1, 4096), (1, 8000)]


def main():
    global SHAPES
    if os.environ.get("SMALL_SHAPE"):   # e.g. SMALL_SHAPE=16384x1024 (one shape: for a rocprofv3 kernel trace)
        SHAPES = [tuple(int(v) for v in item.split("x")) for item in os.environ["SMALL_SHAPE"].split(",")]
    torch.set_num_threads(8)   # the box's CPU share is a cgroup quota: hundreds of spinning pool threads get the process throttled
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    for B, n in SHAPES:
        t = torch.arange(n, dtype=torch.float64) / 8000.0
        x = (torch.sin(2 * np.pi * 110 * t)[None, :] + 0.5 * torch.sin(2 * np.pi * 440 * t + 1.3)[None, :]
             + 0.05 * torch.randn((B, n), generator=g, dtype=torch.float64)).to(torch.float32).to(dev)
        rows = torch.empty((B, M + 2, n), dtype=torch.float64, device=dev)
        eng = pyitd_amd.Engine(n, B, 0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        steps = 20
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        s = eng.summary(B)
        alg = float(np.sum(4 + 8 * s["n_rows"].astype(np.int64))) * n / dt / 1e9
        print("%6d x %7d: %8.3f ms  %7.2f Gsamples/s  %6.0f GB/s (input + the rows produced) = %.3f of peak   rows %s" % (
            B, n, dt * 1e3, B * n / dt / 1e9, alg, alg / 8000.0, sorted(set(int(v) for v in s["n_rows"]))), flush=True)
        eng.close()
        del x, rows


if __name__ == "__main__":
    main()
