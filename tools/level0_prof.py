"""Where a wavefront of the fused level-0 launch spends its life: phase times from a diagnostic build (-DITD_PROF=1, loaded through
PYITD_HIP_LIB; s_memtime at the phase boundaries, one row per wavefront).
usage: PYITD_HIP_LIB=variants/libprof.so python tools/level0_prof.py [log2n]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd.engine import Engine  # noqa: E402
from tests.helpers import sines_noise  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n, M = 1 << lg, 7
x = torch.from_numpy(sines_noise(n)).cuda()
eng = Engine(n, 1)
rows = torch.empty((M + 2, n), dtype=torch.float64, device="cuda")
waves = (n + 511) // 512
buf = torch.zeros((waves, 16), dtype=torch.int64, device="cuda")
L = eng._L
L.itd_debug_prof_buffer.argtypes = [ctypes.c_void_p]
assert L.itd_debug_prof_buffer(buf.data_ptr()) == 0
for _ in range(3):
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
eng.summary(1)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
names = ["issue loads + end samples", "wait for the tile + knot predicate (6 groups)", "ranks + halo knots", "knots by rank -> LDS",
         "knot values (B)", "slopes (S)", "map + stores", "next-level scan + record", "tail (workgroup 0 only)"]
life = b[:, 9]
rt = b[:, 14]
ticks_per_us = life.sum() / (rt.sum() / 100.0)
print("n = 2^%d, %d wavefronts; s_memtime ticks per us (against s_memrealtime at 100 MHz): %.1f" % (lg, waves, ticks_per_us))
print("wavefront lifetime: mean %.2f us  median %.2f  p95 %.2f" % (life.mean() / ticks_per_us, np.median(life) / ticks_per_us, np.percentile(life, 95) / ticks_per_us))
span = (b[:, 15] + b[:, 14]).max() - b[:, 15].min()
print("launch span (first begin to last end): %.1f us; wavefront-time / span = %.1f resident wavefronts on average (%.2f per SIMD)"
      % (span / 100.0, rt.sum() / span, rt.sum() / span / 1024))
for k, nm in enumerate(names):
    print("  %-48s %7.2f us  %5.1f %%" % (nm, b[:, k].mean() / ticks_per_us, 100 * b[:, k].sum() / life.sum()))
# residency over time: wavefronts alive per 5-us slice of the launch
t0 = b[:, 15].min()
beg, end = (b[:, 15] - t0) / 100.0, (b[:, 15] + b[:, 14] - t0) / 100.0
edges = np.arange(0, span / 100.0 + 5, 5.0)
alive = [(np.minimum(end, hi) - np.maximum(beg, lo)).clip(0).sum() / (hi - lo) for lo, hi in zip(edges[:-1], edges[1:])]
print("resident wavefronts per 5-us slice: " + " ".join("%d" % a for a in alive))
