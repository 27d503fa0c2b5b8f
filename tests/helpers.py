"""Shared helpers for the parity tests (canonical hashing identical to oracle/gen_golden.py)."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def canon_bytes(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a.copy()
        a.view(np.uint64)[np.isnan(a)] = np.uint64(0x7FF8000000000000)
    return a.tobytes()


def sha(a):
    return hashlib.sha256(canon_bytes(a)).hexdigest()


def assert_bits_equal(a, b, what=""):
    """Bit-for-bit equality of float64 arrays, any NaN == any NaN."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if canon_bytes(a) == canon_bytes(b):
        return
    bad = ~((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b)))
    idx = np.argwhere(bad)
    first = tuple(idx[0])
    raise AssertionError("%s: %d of %d values differ bitwise; first at %s: %r vs %r" % (
        what, bad.sum(), a.size, first, a[first], b[first]))


def chirp(n, dtype=np.float32):
    """BASELINE config 1 signal."""
    t = np.arange(n, dtype=np.float64) / n
    return np.sin(2 * np.pi * (50 * t + 0.5 * (8000 - 50) * t * t)).astype(dtype)


def sines_noise(n, seed=0, fscale=1.0, dtype=np.float32, fs=48000.0):
    """BASELINE config 2/3 signal."""
    t = np.arange(n, dtype=np.float64) / fs
    x = np.zeros(n, dtype=np.float64)
    for a, f, p in ((1, 110, 0.1), (0.5, 440, 1.3), (0.25, 1760, 2.1), (0.125, 7040, 0.7)):
        x += a * np.sin(2 * np.pi * (f * fscale) * t + p)
    x += 0.05 * np.random.default_rng(seed).standard_normal(n)
    return x.astype(dtype)


def fuzz_signal(rng, kind, n):
    """Random signal families of the parity fuzz (tools/fuzz_parity.py and tests/test_gpu_configs.py)."""
    t = np.arange(n) / max(n - 1, 1)
    if kind == 0:   # white noise
        return rng.standard_normal(n)
    if kind == 1:   # random walk
        return np.cumsum(rng.standard_normal(n))
    if kind == 2:   # quantised (plateaus)
        return np.round(rng.standard_normal(n) * rng.integers(1, 6)) / 4.0
    if kind == 3:   # smooth + few knots
        return np.sin(2 * np.pi * rng.uniform(0.3, 30) * t) + rng.uniform(-1, 1) * t * t
    if kind == 4:   # sines + noise at random level
        return np.sin(2 * np.pi * rng.uniform(5, 500) * t) + rng.uniform(0, 0.3) * rng.standard_normal(n)
    if kind == 5:   # long constant stretches with bursts (leading plateaus -> the reference's NaN branch)
        x = np.zeros(n)
        for _ in range(rng.integers(1, 6)):
            a = rng.integers(0, n)
            b = min(n, a + rng.integers(2, max(3, n // 4)))
            x[a:b] = rng.standard_normal(b - a)
        return x
    if kind == 6:   # alternating with random amplitudes (every sample a knot)
        return ((-1.0) ** np.arange(n)) * (1 + rng.random(n))
    return rng.standard_normal(n) * np.exp(rng.uniform(-300, 300))   # extreme magnitudes


def kf_rates_draws(seed=11, cases=12, families=11):
    """The random draws of tools/kf_rates.py, in its order: (draw index, family, max_iteration, signal) — families 0 .. 7 are fuzz_signal's,
    8 the bench signal (sines + noise), 9 / 10 the same as 16- / 12-bit PCM; 70 000 .. 400 000 samples, half of them float32; a draw with
    a non-finite sample is counted and skipped.  The sweep's seed and the index name a case for good (tests keep indices, not arrays)."""
    rng = np.random.default_rng(seed)
    idx = 0
    for kind in range(families):
        for m in (3, 7, 11):
            for _ in range(cases):
                n = int(rng.integers(70000, 400000))
                if kind >= 8:
                    x = sines_noise(n, seed=int(rng.integers(0, 1 << 30)))
                    if kind >= 9:        # int16 / 12-bit PCM as float32 (the reference's own domain: PyITD.ipynb cell 2)
                        sc = 32768.0 if kind == 9 else 2048.0
                        x = (np.round(x.astype(np.float64) / np.abs(x).max() * (sc - 1)) / sc).astype(np.float32)
                else:
                    x = fuzz_signal(rng, kind, n)
                idx += 1
                if not np.all(np.isfinite(x)):
                    continue
                if kind not in (7,) and rng.random() < 0.5:
                    x = x.astype(np.float32)
                yield idx - 1, kind, m, x


# Draws of kf_rates_draws(seed=11) on which round 4's first knot-side launch delivered WRONG rows (a halo search re-read an end workgroup
# it had already passed and doubled a knot) with every verification of that time green: the delivery-rate sweep found them
# (profiles/r04); tests/test_gpu_fused.py holds them "delivered or refused, never wrong" at every range size.
ROUND4_WRONG_ROWS = (25, 26, 35, 54, 57, 62, 64, 67, 162, 169, 279)


def canon_u64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return np.where(np.isnan(a), np.uint64(0x7FF8000000000000), a.view(np.uint64))


def coarse(x):
    """x rounded to quarter steps (a few quantisation levels: plateaus everywhere).  The fused sparse levels keep both samples of
    every near tie as candidates; with ties at most samples their lists outgrow a workgroup's capacity and they refuse — the one
    family of inputs (beside NaN-producing ones) that still takes the level-by-level repeat since round 4."""
    return (np.round(np.asarray(x, dtype=np.float64) * 3) / 4.0).astype(np.asarray(x).dtype)
