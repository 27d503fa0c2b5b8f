"""Timing of the cubic-spline baseline variant (pyitd_amd/csrc/itd_cubic.hpp) and of the instantaneous amplitude/frequency
step on one 2^24-sample float64 signal, device resident, against the CPU oracle on the host (single thread)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd import _lib
from bench import sines_noise
from oracle import cpu_oracle

n = 1 << 24
x_host = sines_noise(n).astype(np.float64)
x = torch.from_numpy(x_host).cuda()
base = torch.empty(n, dtype=torch.float64, device="cuda")
eng = pyitd_amd.Engine(n, 1, 0)
L = _lib.load()
idx = ctypes.c_int64(0)


def run():
    rc = L.itd_baseline_extract_cubic_f64(eng._h, x.data_ptr(), n, None, 0, base.data_ptr(), ctypes.byref(idx), None)
    assert rc == 0, rc


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    run()
dt = (time.perf_counter() - t0) / 10
e, m = cpu_oracle.extrema_cpp(x_host)
tc = time.perf_counter()
ref = cpu_oracle.itd_baseline_extract_fast(x_host, e, m)
tc = time.perf_counter() - tc
got = base.cpu().numpy()
err = float(np.max(np.abs(got - ref)))
print("cubic baseline (detect mode), 2^24 float64 samples, %d knots: %.3f ms per call = %.0f Msamples/s (synchronous call incl. the "
      "knot-count read-back); CPU oracle %.2f s = %.1f Msamples/s; max |diff| %.2e" % (idx.value, dt * 1e3, n / dt / 1e6, tc, n / tc / 1e6, err))
amp = torch.empty(n, dtype=torch.float64, device="cuda")
ph = torch.empty(n, dtype=torch.float64, device="cuda")
fr = torch.empty(n, dtype=torch.float64, device="cuda")
rot = x - base
L.itd_instantaneous_f64(eng._h, rot.data_ptr(), n, amp.data_ptr(), ph.data_ptr(), fr.data_ptr(), None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    L.itd_instantaneous_f64(eng._h, rot.data_ptr(), n, amp.data_ptr(), ph.data_ptr(), fr.data_ptr(), None)
dt = (time.perf_counter() - t0) / 10
print("instantaneous amplitude/phase/frequency, 2^24 samples: %.3f ms per call = %.0f Msamples/s" % (dt * 1e3, n / dt / 1e6))
