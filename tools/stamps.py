"""Diagnostic: per-phase shader-clock shares of k_extract (needs a library built with -DITD_STAMPS)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd import _lib
from bench import sines_noise
L = _lib.load()
n = 1 << 24
x = torch.from_numpy(sines_noise(n)).cuda()
rows = torch.empty((9, n), dtype=torch.float64, device="cuda")
eng = pyitd_amd.Engine(n, 1, 0)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
names = ["loads+tile0+halo assemble", "commit+sync", "own flags", "pass: run select + fill by rank", "pass: knot values + slopes", "pass: map + stores", "halo write + nan", "detect+record+count"]
for rep in range(2):
    L.itd_debug_stamps(None, 1)
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, 7, rows.data_ptr(), None, None)
    eng.summary(1)
    L.itd_debug_stamps(out, 0)
v = np.array(list(out)[:8], dtype=np.float64)
print("per-phase shader clocks per decomposition (9 levels), share of the stamped total")
for nm, c in zip(names, v):
    print("  %-24s %12.3e  %5.1f %%" % (nm, c, 100 * c / v.sum()))
print("  total %.3e clocks; per tile (9 x %d tiles): %.0f clocks" % (v.sum(), n // 256, v.sum() / (9 * (n // 256))))
