"""PCIe-inclusive rate of the numpy -> numpy API (never the headline value): ITD().itd(x) on a 2^24 float32 signal."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd
from bench import sines_noise
n = 1 << 24
x = sines_noise(n)
d = pyitd_amd.ITD()
d.itd(x, 7)   # warm-up: engine creation, staging buffers
for keep in ("rows+baselines (ITD.itd)",):
    t0 = time.perf_counter(); rows = d.itd(x, 7); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); b = d.get_baselines(); dtb = time.perf_counter() - t0
    print("ITD.itd (rows; the baselines stay on the GPU): %.1f ms end to end = %.0f Msamples/s (H2D %.0f MB, D2H %.0f MB); "
          "get_baselines() afterwards: %.1f ms (D2H %.0f MB)" % (dt * 1e3, n / dt / 1e6, x.nbytes / 1e6, rows.nbytes / 1e6, dtb * 1e3, b.nbytes / 1e6))
eng = pyitd_amd.Engine(n, 1, 0)
t0 = time.perf_counter(); r = eng.decompose_host(x, 7, want_baselines=False); dt = time.perf_counter() - t0
t0 = time.perf_counter(); r = eng.decompose_host(x, 7, want_baselines=False); dt = time.perf_counter() - t0
print("rows only (Engine.decompose_host): %.1f ms = %.0f Msamples/s (D2H %.0f MB)" % (dt * 1e3, n / dt / 1e6, r["rows"].nbytes / 1e6))
