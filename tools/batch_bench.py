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
ap.add_argument("--stream-pool", type=int, default=8, help="also time one-signal launches over this many streams/engines (0 = skip)")
ap.add_argument("--cpu-signals", type=int, default=0, help="also time the C oracle on this many signals over all host cores (0 = skip)")
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

if args.stream_pool > 0:
    # SURVEY 8d config 3, second variant: one signal per launch sequence, round-robin over a pool of streams (one engine each)
    S = args.stream_pool
    pool = [(pyitd_amd.Engine(n, 1, 0), torch.cuda.Stream()) for _ in range(S)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(B):
        e_, st_ = pool[b % S]
        e_.decompose_dev(x[b].data_ptr(), np.float32, n, 1, n, M, rows[b].data_ptr(), None, st_.cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e_, _ in pool[:1]:
        e_.summary(1)
    print("stream pool (%d streams, %d one-signal decompositions): %.2f ms = %.0f Msamples/s" % (S, B, dt * 1e3, B * n / dt / 1e6))
if args.cpu_signals > 0:
    # CPU baseline over independent signals on all host cores (ctypes releases the GIL inside the C oracle)
    from concurrent.futures import ThreadPoolExecutor
    K, T = args.cpu_signals, os.cpu_count()
    work = [distinct[b % 16] for b in range(K)]
    cpu_oracle.itd_lean(work[0], M)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(T, K)) as ex:
        list(ex.map(lambda v: cpu_oracle.itd_lean(v, M)["rows"].shape[0], work))
    dt = time.perf_counter() - t0
    print("CPU oracle, %d signals x 2^%d over %d threads: %.2f s = %.1f Msamples/s" % (K, args.log2n, min(T, K), dt, K * n / dt / 1e6))
