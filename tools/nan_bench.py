"""Timing of inputs that take the reference's NaN branch (a leading plateau = digital silence at the head of the signal:
the first baseline goes NaN, ITD.py:115-116, and the stop test counts under detect_peaks' NaN rules, ITD.py:46-51,64-68).
  (1) one 2^22-sample signal with 0.5 s (at 48 kHz) of leading zeros vs the same signal without them;
  (2) a 64-signal batch of 2^20 samples where half of the signals start with silence vs a batch where none does.
  (3) a batch of 4096 x 4096-sample signals (the resident form) where every other signal starts with silence.
All are checked bit-exactly against the CPU oracle on a sample, then timed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from bench import sines_noise
from oracle import cpu_oracle

M = 9


def run(tag, x_np, check):
    B, n = x_np.shape
    x = torch.from_numpy(x_np).cuda()
    rows = torch.empty((B, M + 2, n), dtype=torch.float64, device="cuda")
    eng = pyitd_amd.Engine(n, B, 0)
    torch.cuda.synchronize()
    eng.decompose_dev(x.data_ptr(), x_np.dtype, n, B, n, M, rows.data_ptr(), None, None)
    s = eng.summary(B)
    for b in check:
        ref = cpu_oracle.itd_lean(x_np[b], M)
        nr = int(s["n_rows"][b])
        got = rows[b, :nr].cpu().numpy()
        a, r = got.view(np.uint64).copy(), ref["rows"].view(np.uint64).copy()
        a[np.isnan(got)] = 0
        r[np.isnan(ref["rows"])] = 0
        assert nr == ref["rows"].shape[0] and np.array_equal(a, r), (tag, b)
    steps = 10
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.decompose_dev(x.data_ptr(), x_np.dtype, n, B, n, M, rows.data_ptr(), None, None)
    eng.summary(B)
    dt = (time.perf_counter() - t0) / steps
    print("%-58s %8.3f ms per decomposition  (%d rows; bit-exact vs oracle on %s)" % (tag, dt * 1e3, int(s["n_rows"][0]), list(check)))
    eng.close()
    return dt


n = 1 << 22
x = sines_noise(n, seed=3)[None]
t_plain = run("2^22 samples, no silence", x, [0])
xs = x.copy()
xs[0, :24000] = 0.0
t_sil = run("2^22 samples, 0.5 s of leading zeros (NaN branch)", xs, [0])
print("  ratio %.3f" % (t_sil / t_plain))
B, n = 64, 1 << 20
xb = np.stack([sines_noise(n, seed=b) for b in range(B)])
t_plain = run("batch 64 x 2^20, no silence", xb, [0, 1])
xs = xb.copy()
rng = np.random.default_rng(0)
for b in range(0, B, 2):
    xs[b, : int(rng.integers(100, 48000))] = 0.0
t_sil = run("batch 64 x 2^20, every other signal starts with silence", xs, [0, 1, 2])
print("  ratio %.3f" % (t_sil / t_plain))
B, n = 4096, 4096
xb = np.stack([sines_noise(n, seed=b % 64, fscale=1.0 + b / 4096.0) for b in range(B)])
t_plain = run("batch 4096 x 4096 (resident), no silence", xb, [0, 1, 4095])
xs = xb.copy()
for b in range(0, B, 2):
    xs[b, : int(rng.integers(2, 1500))] = 0.0
t_sil = run("batch 4096 x 4096 (resident), every other signal starts with silence", xs, [0, 1, 2, 4094])
print("  ratio %.3f" % (t_sil / t_plain))
