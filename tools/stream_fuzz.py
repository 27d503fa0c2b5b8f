"""The block-wise operators (pyitd_amd.streaming, itd_stream_*) against the CPU statement of the recipe (oracle/stream_oracle.py) on random
streams: block sizes 8 .. 4096, 1 .. 6 blocks, 1 .. 3 channels, margins 1 .. 11, shared knots or not, seven signal families — the open-ended
form of tests/test_gpu_stream.py::test_streams_follow_the_oracle_on_seeded_signals.  Tier-1 (linear): bit for bit; cubic: 1e-9 of the scale.
usage: python tools/stream_fuzz.py [trials] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal, assert_bits_equal
from pyitd_amd import streaming as S
from oracle import stream_oracle as so

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = skipped = 0
t0 = time.time()
for trial in range(trials):
    kind = int(rng.integers(0, 7))
    L = int(rng.choice([8, 24, 100, 512, 1000, 4096]))
    nb = int(rng.integers(1, 7))
    C = int(rng.integers(1, 4))
    x = np.stack([fuzz_signal(rng, kind if c == 0 else int(rng.integers(0, 7)), L * nb) for c in range(C)])
    if kind == 5:
        x += 1e-3 * rng.standard_normal(x.shape)     # plateaus give 0/0 in the cubic operator's reference too: keep finite
    margin = int(rng.integers(1, 12))
    shared = bool(rng.integers(0, 2))
    what = "trial %d (kind %d L %d blocks %d channels %d margin %d shared %d)" % (trial, kind, L, nb, C, margin, shared)
    with np.errstate(all="ignore"):
        ref = so.oracle_blockwise_cubic(x, L, margin, shared)
        got = S.blockwise(x, L, "cubic", margin, shared)
        if np.isfinite(ref).all():
            scale = max(1.0, float(np.abs(ref).max()))
            if got.shape != ref.shape or not np.isfinite(got).all() or np.abs(got - ref).max() > 1e-9 * scale:
                bad += 1
                print("cubic MISMATCH " + what)
        else:
            skipped += 1
        rref, bref = so.oracle_blockwise_linear(x, L)
        rot, base = S.blockwise(x, L, "linear")
        try:                                         # (bit for bit, any NaN equal to any NaN: the helper of the suite)
            assert_bits_equal(base, bref, "baseline")
            assert_bits_equal(rot, rref, "rotation")
        except AssertionError as ex:
            bad += 1
            print("linear MISMATCH " + what + ": " + str(ex)[:160])
            if os.environ.get("STREAM_FUZZ_DUMP"):
                np.savez(os.path.join(os.environ["STREAM_FUZZ_DUMP"], "stream_case_%d.npz" % trial), x=x, L=L)
print("%d trials, %d mismatches (%d cubic comparisons skipped: the reference's 0/0), %.1f s" % (trials, bad, skipped, time.time() - t0))
sys.exit(1 if bad else 0)
