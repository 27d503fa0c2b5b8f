"""numpy -> numpy latency of ITD().itd(x) on short signals (the reference's own demo sizes), against the C oracle on one host
thread.  (GPU box)  usage: python tools/host_api_small.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402
from oracle import cpu_oracle  # noqa: E402  (the checker / CPU figure only)
cpu_oracle.lib()
rng = np.random.default_rng(3)
for n in (400, 8000, 65536, 1 << 20):
    t = np.arange(n) / 8000.0
    x = np.sin(2 * np.pi * 110 * t) + 0.5 * np.sin(2 * np.pi * 440 * t + 1.3) + 0.05 * rng.standard_normal(n)
    dec = pyitd_amd.ITD()
    for _ in range(5):
        rows = dec.itd(x, max_iteration=11)
    reps = 200 if n <= 65536 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        rows = dec.itd(x, max_iteration=11)
    gpu = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(max(1, reps // 10)):
        ref = cpu_oracle.itd(x, 11)
    cpu = (time.perf_counter() - t0) / max(1, reps // 10)
    same = rows.shape == ref["rows"].shape and np.array_equal(rows.view(np.uint64), ref["rows"].view(np.uint64))
    print("n = %8d: ITD().itd %.3f ms (%.1f Msamples/s), C oracle one thread %.3f ms; rows %s bit-exact %s" % (
        n, gpu * 1e3, n / gpu / 1e6, cpu * 1e3, rows.shape, same), flush=True)
