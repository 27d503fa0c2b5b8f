"""Randomised parity sweep of the GPU path against the CPU oracle (bit-exact rows, baselines, stop reason).
usage (GPU box): python tools/fuzz_parity.py [cases] [seed]          single signals through ITD.itd
                 python tools/fuzz_parity.py batch [cases] [seed]    random batches through itd_batch (grid.y = signal)
A third of the inputs get NaNs sprinkled in (the reference's NaN branch at level 0), some an infinity.  PYITD_FUSE_MODE=1 in
the environment switches the fused sparse levels off (long signals), PYITD_LEVEL0_MODE=1 selects the record-driven level 0.
FUZZ_MIN_N=65536 lifts every length into the fused levels' range.
FUZZ_MAX_N=4096 folds every length into 3 .. 4096 (the resident form's range), FUZZ_NO_NAN=1 leaves the inputs as drawn (a NaN input
sends the engine's next 16 decompositions level by level: without them nearly every case runs resident; PYITD_RESIDENT_MODE=1 = none)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd
from oracle import cpu_oracle

batch_mode = len(sys.argv) > 1 and sys.argv[1] == "batch"
argv = sys.argv[2:] if batch_mode else sys.argv[1:]
cases = int(argv[0]) if len(argv) > 0 else 200
rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 0)
cpu_oracle.lib()
MAX_N = int(os.environ.get("FUZZ_MAX_N", "0"))
MIN_N = int(os.environ.get("FUZZ_MIN_N", "0"))


def fold(n):
    if MIN_N and n < MIN_N:
        n += MIN_N
    return 3 + (n - 3) % (MAX_N - 2) if MAX_N and n > MAX_N else n


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import fuzz_signal


def make(kind, n):
    x = fuzz_signal(rng, kind, n)
    r = rng.random()
    if os.environ.get("FUZZ_NO_NAN"):
        r = 1.0
    if r < 0.33 and np.all(np.isfinite(x)):          # NaNs in the input: singles, runs, at the ends, on tile boundaries
        k = int(rng.integers(1, 7))
        at = rng.integers(0, n, k)
        if rng.random() < 0.5:
            at = np.concatenate([at, np.array([0, n - 1, 511, 512, 1023])[: int(rng.integers(0, 6))] % n])
        x = x.copy()
        x[at] = np.nan
        if rng.random() < 0.3:
            x[int(rng.integers(0, n))] = np.inf * (1 if rng.random() < 0.5 else -1)
    return x


def canon(a):
    return np.where(np.isnan(a), 0x7ff8000000000000, a.view(np.uint64))


if batch_mode:
    bad = 0
    t0 = time.time()
    for c in range(cases):
        B = int(rng.integers(1, 40))
        n = fold(int(rng.choice([3, 17, 511, 512, 513, 4095, 4096, 4097, int(rng.integers(3, 40000))])))
        m = int(rng.integers(0, 9))
        dtype = np.float32 if rng.random() < 0.5 else np.float64
        x = np.stack([make(int(rng.integers(0, 7)), n) for _ in range(B)]).astype(dtype)
        if np.any(np.isinf(x) & ~np.isinf(x.astype(np.float64))):    # float32 overflow of an extreme-magnitude draw
            continue
        out = pyitd_amd.itd_batch(x, m, keep_baselines=bool(rng.random() < 0.3))
        for b in range(B):
            ref = cpu_oracle.itd(x[b], m)
            nr = int(out["n_rows"][b])
            ok = nr == ref["rows"].shape[0] and ("natural", "timeout")[int(out["stop"][b])] == ref["stop"] and \
                np.array_equal(canon(out["rows"][b, :nr]), canon(ref["rows"]))
            if ok and "baselines" in out:
                nb = int(out["n_baselines"][b])
                ok = nb == ref["baselines"].shape[0] and np.array_equal(canon(out["baselines"][b, :nb]), canon(ref["baselines"]))
            if not ok:
                bad += 1
                print("batch case %d B %d n %d m %d %s signal %d: MISMATCH" % (c, B, n, m, dtype.__name__, b))
    print("%d batches, %d mismatching signals, %.1f s" % (cases, bad, time.time() - t0))
    sys.exit(1 if bad else 0)

bad = 0
t0 = time.time()
for c in range(cases):
    kind = int(rng.integers(0, 8))
    n = fold(int(rng.choice([3, 4, 5, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 4097, int(rng.integers(3, 70000)), int(rng.integers(3, 300000))])))
    m = int(rng.integers(0, 12))
    dtype = np.float32 if rng.random() < 0.5 and kind != 7 else np.float64
    x = make(kind, n).astype(dtype)
    ref = cpu_oracle.itd(x, m)
    dec = pyitd_amd.ITD()
    try:
        rows = dec.itd(x, max_iteration=m)
    except Exception as e:   # noqa
        print("case %d kind %d n %d m %d %s: EXCEPTION %r" % (c, kind, n, m, dtype.__name__, e))
        bad += 1
        continue
    ok = rows.shape == ref["rows"].shape and np.array_equal(
        np.where(np.isnan(rows), 0x7ff8000000000000, rows.view(np.uint64)),
        np.where(np.isnan(ref["rows"]), 0x7ff8000000000000, ref["rows"].view(np.uint64))) and dec.stop_reason == ref["stop"]
    if ok and c % 3 == 0:
        b = dec.get_baselines()
        ok = b.shape == ref["baselines"].shape and np.array_equal(canon(b), canon(ref["baselines"]))
    if not ok:
        bad += 1
        print("case %d kind %d n %d m %d %s: MISMATCH rows %s vs %s stop %s vs %s" % (c, kind, n, m, dtype.__name__, rows.shape, ref["rows"].shape, dec.stop_reason, ref["stop"]))
print("%d cases, %d mismatches, %.1f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
