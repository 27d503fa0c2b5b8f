"""BASELINE configs[2]: batch of independent 2^20-sample float32 signals (noise draw b mod 16, f*(1+b/8192)), 8 levels, one
MI355X, device resident.  Checks a sample of signals bit-exactly against the CPU oracle, then times whole-batch
decompositions for several chunk sizes (signals per launch sequence; 0 = the engine's automatic choice, B = level-major
over the whole batch as in round 1) and, optionally, the stream-pool form."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from bench import batch_signals_device, sines_noise

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--log2n", type=int, default=20)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--chunks", type=str, default="0,1024,8,16,32,64", help="chunk sizes to time (0 = automatic)")
ap.add_argument("--streams", type=str, default="2", help="batch streams to time (the engine default is 2; e.g. 1,2 times both)")
ap.add_argument("--stream-pool", type=int, default=0, help="also time one-signal launches over this many streams/engines (0 = skip)")
args = ap.parse_args()
B, n, M = args.batch, 1 << args.log2n, 7
dev = torch.device("cuda", 0)
x = batch_signals_device(torch, dev, 0, B, n)
for b in (0, 7, 15, B - 1):      # these four come from the host recipe, bit for bit, so that the oracle can check them
    x[b] = torch.from_numpy(sines_noise(n, seed=b % 16, fscale=1.0 + b / 8192.0)).to(dev)
rows = torch.empty((B, M + 2, n), dtype=torch.float64, device=dev)
eng = pyitd_amd.Engine(n, B, 0)
print("workspace %.1f GB, rows %.1f GB" % (eng.workspace_bytes / 1e9, rows.numel() * 8 / 1e9))
torch.cuda.synchronize()
eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
s = eng.summary(B)
from oracle import cpu_oracle
for b in (0, 7, 15, B - 1):
    ref = cpu_oracle.itd_lean(x[b].cpu().numpy(), M)
    nr = int(s["n_rows"][b])
    got = rows[b, :nr].cpu().numpy()
    assert nr == ref["rows"].shape[0] and np.array_equal(got.view(np.uint64), ref["rows"].view(np.uint64)), b
print("parity ok on signals 0, 7, 15, %d; rows per signal %s" % (B - 1, sorted(set(s["n_rows"].tolist()))))
for chunk, streams in [(int(c), int(k)) for k in args.streams.split(",") for c in args.chunks.split(",")]:
    eng.set_batch_chunk(chunk)
    eng.set_batch_streams(streams)
    eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
    eng.summary(B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
    eng.summary(B)
    dt = (time.perf_counter() - t0) / args.steps
    print("streams %d chunk %4d: batch %d x 2^%d, 8 levels: %.2f ms per batch decomposition = %.0f Msamples/s, %.0f GB/s algorithmic = %.3f of 8 TB/s" % (
        streams, chunk, B, args.log2n, dt * 1e3, B * n / dt / 1e6, 188.0 * B * n / dt / 1e9, 188.0 * B * n / dt / 8e12))
eng.set_batch_chunk(0)

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
