"""Does the distance between the rotation rows and the baseline rows matter?  (caller-owned baselines buffer: both addresses are ours)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd
from pyitd_amd.engine import TIME_EXTRACT
from bench import sines_noise
n, M = 1 << 24, 7
R = M + 2
x = torch.from_numpy(sines_noise(n)).cuda()
big = torch.empty(2 * R * n + (1 << 22), dtype=torch.float64, device="cuda")
eng = pyitd_amd.Engine(n, 1, 0)
def run(tag, rows, bases):
    torch.cuda.synchronize()
    for _ in range(2): eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None if bases is None else bases.data_ptr(), None)
    eng.summary(1)
    eng.set_timing(12, stride=1)
    for _ in range(12): eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None if bases is None else bases.data_ptr(), None)
    eng.summary(1)
    ms, c = eng.kernel_timing(TIME_EXTRACT)
    d = 0 if bases is None else (bases.data_ptr() - rows.data_ptr())
    print("%-44s distance %% 128 MiB = %10d B   levels>=1 %.1f us" % (tag, d % (1 << 27), ms / c * 1e3))
    eng.set_timing(0)
rows = big[: R * n].view(R, n)
run("engine's rotating slots", rows, None)
for pad in (0, 512, 1 << 9 << 3, (1 << 17) + 512, (1 << 20) + 1536):
    b = big[R * n + pad: 2 * R * n + pad].view(R, n)
    run("caller baselines, pad %d doubles" % pad, rows, b)
