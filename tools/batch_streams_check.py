import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pyitd_amd.engine import Engine
from bench import sines_noise
for n, B in ((1 << 24, 4), (1 << 23, 6), (1 << 22, 16)):
    x = torch.from_numpy(np.stack([sines_noise(n, seed=b) for b in range(B)])).cuda()
    rows = torch.empty((B, 9, n), dtype=torch.float64, device="cuda")
    eng = Engine(n, B)
    for S in (1, 2):
        eng.set_batch_streams(S)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, 7, rows.data_ptr(), None, None)
            eng.summary(B); dt = (time.perf_counter() - t0) / 5
        print("n 2^%d B %d streams %d: %.3f ms = %.0f Msamples/s" % (n.bit_length() - 1, B, S, dt * 1e3, n * B / dt / 1e6))
    eng.close(); del x, rows
