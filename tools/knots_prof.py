"""Where the knot side's one launch (k_kf_knots) spends its time: the 100 MHz wall clock at every workgroup's phase boundaries, from a
diagnostic build (-DITD_PROF=1, loaded through PYITD_HIP_LIB).
usage: PYITD_HIP_LIB=variants/libprof.so python tools/knots_prof.py [log2n]"""
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
eng.set_fuse_mode(2)            # fused only
rows = torch.empty((M + 2, n), dtype=torch.float64, device="cuda")
wgs = 4096
buf = torch.zeros((wgs, 64), dtype=torch.int64, device="cuda")
L = eng._L
L.itd_debug_knots_prof_buffer.argtypes = [ctypes.c_void_p]
assert L.itd_debug_knots_prof_buffer(buf.data_ptr()) == 0
for _ in range(4):
    buf.zero_()
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    eng.summary(1)
torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
b = b[b[:, 60] > 0]
t0 = b[:, 0].min()
us = lambda v: (v - t0) / 100.0
print("n = 2^%d: %d workgroups; launch span (first start .. last end) %.2f us" % (lg, len(b), us(b[:, 60].max())))
print("start skew: last workgroup starts at %.2f us" % us(b[:, 0].max()))
L0 = eng.last_fuse_level
print("first fused level %d" % L0)
names = [(1, "ticket"), (2, "state + words + tie flags loaded"), (3, "expansion + values (hand-over done)")]
for li in range(M + 2 - L0):
    names += [(4 + 4 * li, "L%d: knots by rank, words" % (L0 + li)), (5 + 4 * li, "L%d: halo arrived" % (L0 + li)),
              (6 + 4 * li, "L%d: B, S, table" % (L0 + li)), (7 + 4 * li, "L%d: map + compaction" % (L0 + li))]
names += [(60, "end")]
fine = [(4 + 4, "second fused level: knots by rank, words"), (49, "  halo: search set up (wavefront 0)"), (50, "  halo: granules arrived"), (51, "  halo: walked"), (52, "    (wavefront 1: its side walked)"), (53, "    (wavefront 3: the tiles' words and run starts left)"), (5 + 4, "  halo: barrier"),
        (6 + 4, "second fused level: B, S, table"), (44, "  maps done"), (45, "  end samples (thread 0)"), (46, "  scan"), (47, "  record's fixed part (thread 0)"), (48, "  compaction writes"), (7 + 4, "  barrier")]
prev = b[:, 0]
print("%-44s %9s %9s %9s   %s" % ("mark", "median at", "max at", "d median", "(us since the first workgroup's start)"))
for k, nm in names:
    col = b[:, k]
    ok = col > 0
    if not ok.any():
        continue
    print("%-44s %9.2f %9.2f %9.2f" % (nm, np.median(us(col[ok])), us(col[ok]).max(), np.median((col - prev)[ok]) / 100.0))
    prev = col
if (b[:, 44] > 0).any():
    prev = b[:, fine[0][0]]
    for k, nm in fine[1:]:
        if k in (52, 53):       # other wavefronts' marks: since the level's start, not part of wavefront 0's chain
            print("%-44s %9.2f %9s %9s   (%.2f us behind the level's start)" % (nm, np.median(us(b[:, k])), "", "", np.median(b[:, k] - b[:, fine[0][0]]) / 100.0))
            continue
        print("%-44s %9.2f %9s %9.2f" % (nm, np.median(us(b[:, k])), "", np.median(b[:, k] - prev) / 100.0))
        prev = b[:, k]
