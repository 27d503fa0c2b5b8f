"""BASELINE configs[2]: batch of independent 2^20-sample float32 signals, 8 levels, one MI355X, device resident.
Checks a sample of signals bit-exactly against the CPU oracle, then times whole-batch decompositions."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from bench import sines_noise

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
B, n, M = args.batch, 1 << args.log2n, 7
distinct = np.stack([sines_noise(n, seed=b) for b in range(16)])       # 16 distinct noise draws, tiled
x = torch.from_numpy(distinct).cuda().repeat((B + 15) // 16, 1)[:B].contiguous()
rows = torch.empty((B, M + 2, n), dtype=torch.float64, device="cuda")
eng = pyitd_amd.Engine(n, B, 0)
print("workspace %.1f GB, rows %.1f GB" % (eng.workspace_bytes / 1e9, rows.numel() * 8 / 1e9))
torch.cuda.synchronize()
eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
s = eng.summary(B)
from oracle import cpu_oracle
for b in (0, 7, 15, B - 1):
    ref = cpu_oracle.itd_lean(distinct[b % 16], M)
    nr = int(s["n_rows"][b])
    got = rows[b, :nr].cpu().numpy()
    assert nr == ref["rows"].shape[0] and np.array_equal(got.view(np.uint64), ref["rows"].view(np.uint64)), b
print("parity ok on signals 0, 7, 15, %d; rows per signal %s" % (B - 1, sorted(set(s["n_rows"].tolist()))))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
eng.summary(B)
dt = (time.perf_counter() - t0) / args.steps
print("batch %d x 2^%d, 8 levels: %.2f ms per batch decomposition = %.0f Msamples/s, %.0f GB/s algorithmic" % (
    B, args.log2n, dt * 1e3, B * n / dt / 1e6, 188.0 * B * n / dt / 1e9))
