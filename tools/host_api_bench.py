"""PCIe-inclusive time of the numpy -> numpy API on a 2^24-sample float32 signal (never the headline value), taken apart: the C call
into a resident result array, the same with the fresh result array the API has to return, ITD().itd() as a user calls it (the
previous result is released inside the call), and what the host's memory management alone costs for 1.2 GB."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402
from bench import sines_noise  # noqa: E402

n = 1 << 24
x = sines_noise(n)


def best(f, rep=3):
    b = 1e9
    for _ in range(rep):
        t0 = time.perf_counter()
        r = f()
        b = min(b, time.perf_counter() - t0)
        del r
    return b * 1e3


print("host memory alone: first touch of a fresh 1.2 GB array (one thread) %.1f ms" % best(lambda: np.empty((9, n)).fill(1.0)), end="")
a = np.empty((9, n))
a.fill(1.0)
t0 = time.perf_counter()
del a
print(", releasing it %.1f ms" % ((time.perf_counter() - t0) * 1e3))
eng = pyitd_amd.Engine(n, 1, 0)
eng.decompose_host(x, 7, want_baselines=False)
rows = np.empty((9, n))
rows.fill(0.0)
nr, nb, stop = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
kc = np.zeros(23, np.int64)
P = ctypes.c_void_p
print("itd_decompose_host_f32 into a resident array (H2D 67 MB, 0.56 ms of kernels, D2H 1208 MB): %.1f ms"
      % best(lambda: eng._L.itd_decompose_host_f32(eng._h, x.ctypes.data_as(P), n, 7, rows.ctypes.data_as(P), None, ctypes.byref(nr),
                                                   ctypes.byref(nb), ctypes.byref(stop), kc.ctypes.data_as(P))))
print("Engine.decompose_host (a fresh result array per call, rows only): %.1f ms" % best(lambda: eng.decompose_host(x, 7, want_baselines=False)))
d = pyitd_amd.ITD()
d.itd(x, 7)
dt = best(lambda: d.itd(x, 7))
print("ITD().itd(x, 7) (the previous call's rows are released inside; the baselines stay on the GPU): %.1f ms = %.0f Msamples/s" % (dt, n / dt / 1e3))
buf = np.empty((9, n))
buf.fill(0.0)
d.itd(x, 7, out=buf)
dt = best(lambda: d.itd(x, 7, out=buf), rep=5)
print("ITD().itd(x, 7, out=buf) in a loop (the result array is the caller's, reused): %.1f ms = %.0f Msamples/s" % (dt, n / dt / 1e3))
d.itd(x, 7)
t0 = time.perf_counter()
b = d.get_baselines()
print("get_baselines() afterwards (D2H %.0f MB into a fresh array): %.1f ms" % (b.nbytes / 1e6, (time.perf_counter() - t0) * 1e3))
