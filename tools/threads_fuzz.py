"""Engines on concurrent host threads (an engine is not thread-safe; several of them — one per host thread, each with its own stream — share the
GPU): every thread decomposes random signals of random sizes in a loop (host form and device form, fused levels on for the long ones, MEITD on
some), every result against the oracle.  The open-ended form of tests/test_gpu_configs.py::test_engines_on_concurrent_host_threads.
usage: python tools/threads_fuzz.py [iterations per thread] [threads] [seed]"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal, canon_u64
from oracle import cpu_oracle as O
import pyitd_amd as P

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.set_num_threads(2)
errors = []
done = [0] * threads


def worker(k):
    rng = np.random.default_rng(seed * 1000 + k)
    stream = torch.cuda.Stream()
    try:
        for it in range(iters):
            n = int(rng.choice([400, 4096, 8000, 50000, 65536, 131072, 300000]))
            m = int(rng.integers(2, 9))
            x = fuzz_signal(rng, int(rng.integers(0, 8)), n).astype(np.float32 if it % 2 else np.float64)
            with np.errstate(all="ignore"):
                ref = O.itd(x, m)
            eng = P.Engine(n, 1, 0)
            if n >= 65536:
                eng.set_fuse_min_samples(65536)
            if it % 3 == 2:
                res = eng.decompose_host(x, m, want_baselines=True)
                got, nr = res["rows"], res["rows"].shape[0]
            else:
                xd = torch.from_numpy(x).cuda()
                rows = torch.empty((m + 2, n), dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                eng.decompose_dev(xd.data_ptr(), x.dtype.type, n, 1, n, m, rows.data_ptr(), None, stream.cuda_stream)
                nr = int(eng.summary(1)["n_rows"][0])
                got = rows[:nr].cpu().numpy()
            eng.close()
            if nr != ref["rows"].shape[0] or not np.array_equal(canon_u64(got), canon_u64(ref["rows"])):
                errors.append("thread %d iteration %d: n %d m %d %s" % (k, it, n, m, x.dtype))
            done[k] += 1
    except Exception as ex:  # noqa: BLE001
        errors.append("thread %d: %r" % (k, ex))


t0 = time.time()
with ThreadPoolExecutor(threads) as ex:
    list(ex.map(worker, range(threads)))
for e in errors[:10]:
    print("MISMATCH " + e)
print("%d threads x %d iterations: %d decompositions, %d mismatches, %.1f s" % (threads, iters, sum(done), len(errors), time.time() - t0))
sys.exit(1 if errors else 0)
