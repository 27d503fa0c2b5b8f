"""The instantaneous amplitude / phase / frequency step (pyitd_amd.instantaneous, itd_tfe.hpp) against the independent numpy statement of
tests/test_gpu_tfe.py on random oscillations of random lengths: amplitudes exact, phase and frequency to 1e-12.
usage: python tools/tfe_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_tfe import numpy_tfe
import pyitd_amd as P

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(cases):
    n = int(rng.choice([3, 5, 64, 100, 511, 512, 513, 1000, 4097, 20000, 65536, 100003, 300001]))
    t = np.arange(n)
    fam = case % 5
    if fam == 0:
        x = np.sin(2 * np.pi * rng.uniform(0.001, 0.2) * t + rng.uniform(0, 6)) * (1 + 0.5 * np.sin(2 * np.pi * rng.uniform(1e-4, 1e-2) * t))
    elif fam == 1:
        x = rng.standard_normal(n)
    elif fam == 2:
        x = np.sin(2 * np.pi * (rng.uniform(0.001, 0.01) * t + rng.uniform(1e-9, 1e-7) * t * t))
    elif fam == 3:
        x = np.round(3 * np.sin(2 * np.pi * rng.uniform(0.001, 0.05) * t))          # zeros and plateaus
    else:
        x = np.cumsum(rng.standard_normal(n)) * 10.0 ** rng.integers(-5, 6)
    try:
        with np.errstate(all="ignore"):
            a, p, f = P.instantaneous(x)
            ra, rp, rf = numpy_tfe(x)
        assert np.array_equal(a, ra), "amplitude"
        assert np.max(np.abs(p - rp)) < 1e-12 and np.max(np.abs(f - rf)) < 1e-12, "phase %.2e frequency %.2e" % (np.max(np.abs(p - rp)), np.max(np.abs(f - rf)))
    except AssertionError as ex:
        bad += 1
        print("MISMATCH case %d (family %d, n %d): %s" % (case, fam, n, str(ex)[:200]))
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
