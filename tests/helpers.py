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
