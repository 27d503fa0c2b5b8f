"""Phase times of the chain launch from a diagnostic build (-DITD_CHAIN_PROF=1, loaded through PYITD_HIP_LIB).
usage: PYITD_HIP_LIB=variants/libchain_prof.so python tools/chain_prof.py [log2n] [grid]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd.engine import Engine, CHAIN_ONLY, CHAIN_OFF  # noqa: E402
from tests.helpers import sines_noise  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n, M = 1 << lg, 7
x = torch.from_numpy(sines_noise(n)).cuda()
eng = Engine(n, 1)
if len(sys.argv) > 2:
    eng.set_chain_grid(int(sys.argv[2]))
rows = torch.empty((M + 2, n), dtype=torch.float64, device="cuda")
out = (ctypes.c_uint64 * 16)()
for mode, name in ((CHAIN_OFF, "level by level"), (CHAIN_ONLY, "chain")):
    eng.set_chain_mode(mode)
    for _ in range(3):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    eng.summary(1)
    eng._L.itd_debug_chain_prof(eng._h, out, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    eng.summary(1)
    print("%-16s %.3f ms per decomposition" % (name, dt * 1e3))
    if mode == CHAIN_ONLY:
        eng._L.itd_debug_chain_prof(eng._h, out, 1)
        v = [int(a) for a in out]
        tiles = max(v[6], 1)
        lv = tiles * (M + 2)
        us = lambda ticks, per: ticks / 100.0 / per   # noqa: E731
        print("tiles %d (per call %d), spins per tile-level %.2f" % (tiles, tiles // reps, v[7] / lv))
        print("per tile: total %.2f us" % us(v[0], tiles))
        print("per level: first sweep %.2f us | front end level0 %.2f us (per tile), later levels %.2f us | passes+map %.2f us | scan+publish %.2f us"
              % (us(v[1], tiles * (M + 1)), us(v[2], tiles), us(v[3], tiles * (M + 1)), us(v[4], lv), us(v[5], lv)))
        print("front end by level 1..8 (us per tile): " + " ".join("%.2f" % us(v[8 + j], tiles) for j in range(8)))
