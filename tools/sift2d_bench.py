"""The only timing the reference records (siftED2D.ipynb cell 3: totalextract2d on a 512 x 512 image = 10.1457 s on the
author's machine, numba prange incl. JIT): 20 ensemble members x 4 sweeps x 512 signals of 512 samples through the FITPACK
flavour of the baseline.  Here: the same workload on one MI355X (host arrays in, host arrays out), and the scipy-backed CPU
oracle on a bounded sample of it for scale."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd
from oracle import spline_oracle

rng = np.random.default_rng(0)
yy, xx = np.mgrid[0:512, 0:512]
img = (128 + 60 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 25 * rng.standard_normal((512, 512))).clip(0, 255).round()
np.random.seed(1)
pyitd_amd.totalextract2d(img, verbose=False)          # warm-up: workspace allocation
ts = []
for _ in range(5):
    np.random.seed(1)
    t0 = time.perf_counter()
    hl = pyitd_amd.totalextract2d(img, verbose=False)
    ts.append(time.perf_counter() - t0)
print("totalextract2d(512 x 512), 40 960 extractions of 512 samples: %.1f ms end to end (best of 5; host numpy noise generation, "
      "H2D/D2H included) = %.0f Msamples/s; the reference's recorded run: 10.1457 s" % (min(ts) * 1e3, 40960 * 512 / min(ts) / 1e6))
assert abs((hl.sum(axis=0) - img)).max() < 1e-9
t0 = time.perf_counter()
cw = pyitd_amd.crossways_itd_baseline_extract(img)
t1 = time.perf_counter() - t0
print("one crossways sweep (2048 extractions): %.2f ms" % (t1 * 1e3))
t0 = time.perf_counter()
ref = spline_oracle.crossways(img[:128, :128])
t2 = time.perf_counter() - t0
print("CPU oracle (scipy splrep + numpy de Boor), crossways of a 128 x 128 crop (512 extractions of 128 samples): %.2f s" % t2)
err = np.max(np.abs(pyitd_amd.crossways_itd_baseline_extract(img[:128, :128]) - ref))
print("max |GPU - oracle| on the crop: %.2e" % err)
