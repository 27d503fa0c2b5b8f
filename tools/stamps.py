"""Diagnostic: per-phase shader-clock shares, wavefront lifetime and residency of k_extract_r at one level
(library built with -DITD_STAMPS [-DITD_STAMP_LEVEL=j], selected through PYITD_HIP_LIB)."""
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
init = (ctypes.c_ulonglong * 16)(*([0] * 16))
names = ["loads + candidate records + halo knots", "tile-relative ranks", "pass: fill by rank", "pass: knot values + slopes",
         "pass: map + stores", "(pass loop exit)", "next level's knot scan + record + counts", "(unused)"]
for rep in range(3):
    L.itd_debug_stamps(None, 1)
    # slot 9 = min(start): preset to max
    eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, 7, rows.data_ptr(), None, None)
    eng.summary(1)
    L.itd_debug_stamps(out, 0)
v = np.array(list(out), dtype=np.float64)
waves = v[11]
print("sampled wavefronts %d (every 64th tile of one launch)" % waves)
tot = v[:8].sum()
for nm, c in zip(names, v[:8]):
    print("  %-40s %8.0f clocks/wavefront  %5.1f %%" % (nm, c / waves, 100 * c / tot))
life = v[8] / waves
print("  stamped total %.0f, lifetime %.0f s_memtime clocks per wavefront" % (tot / waves, life))
