"""The resident form of short signals (itd_set_resident_mode, pyitd_amd/csrc/itd_resident.hpp): one launch, one workgroup
per signal, the signal in LDS through the whole driver loop (ITD.py:384-432).  Checked bit for bit against the CPU oracle
and, summary entry for summary entry, against the level-by-level engine; the level-by-level repeat of calls that meet a
non-finite value (leading plateaus -> the reference's NaN branch) included."""
import numpy as np
import pytest

from helpers import assert_bits_equal, fuzz_signal, load_golden, sines_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    cpu_oracle.lib()
    return cpu_oracle


def _run(P, torch, x_np, m, mode, keep=True, window=0):
    from pyitd_amd.engine import LEVEL0_AUTO, RESIDENT_OFF
    B, n = x_np.shape
    xd = torch.from_numpy(x_np).cuda()
    rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bases = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda") if keep else None
    eng = P.Engine(n, B, 0)
    eng.set_level0_mode(LEVEL0_AUTO)     # whatever PYITD_LEVEL0_MODE says: the automatic resident form needs
    eng.set_resident_mode(mode)
    eng.set_resident_window(window)
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), x_np.dtype, n, B, n, m, rows.data_ptr(), bases.data_ptr() if keep else None, None)
    s = eng.summary(B)
    repeats = eng.resident_repeats
    out = rows.cpu().numpy(), (bases.cpu().numpy() if keep else None), s, repeats
    eng.close()
    return out


def _check_against_oracle(oracle, x, m, rows, bases, s, what):
    for b in range(x.shape[0]):
        ref = oracle.itd(x[b], m)
        nr, nb = int(s["n_rows"][b]), int(s["n_baselines"][b])
        assert nr == ref["rows"].shape[0] and nb == ref["baselines"].shape[0], "%s signal %d: %d rows / %d baselines" % (what, b, nr, nb)
        assert ("natural", "timeout")[int(s["stop"][b])] == ref["stop"]
        assert_bits_equal(rows[b, :nr], ref["rows"], "%s signal %d rows" % (what, b))
        if bases is not None:
            assert_bits_equal(bases[b, :nb], ref["baselines"], "%s signal %d baselines" % (what, b))
        kc = [int(v) for v in s["knot_counts"][b] if v >= 0]
        want = [int(v) for v in ref["knot_counts"]]
        assert kc[1: 1 + len(want)] == want, "%s signal %d knot counts" % (what, b)


FINITE_KINDS = (0, 1, 3, 4, 6, 7)     # fuzz families without plateaus at the signal's ends


@pytest.mark.parametrize("n", [3, 4, 5, 7, 63, 64, 65, 127, 129, 400, 511, 512, 513, 1000, 1024, 1025, 2047, 2049, 3000, 4095, 4096,
                               4097, 5000, 8000, 8191, 8192])
def test_resident_is_bit_exact_at_every_length(P, torch, oracle, n):
    from pyitd_amd.engine import RESIDENT_AUTO, RESIDENT_OFF, RESIDENT_ONLY
    rng = np.random.default_rng(1000 + n)
    for dtype, m in ((np.float64, 7), (np.float32, 3), (np.float64, 0), (np.float32, 20)):
        x = np.stack([fuzz_signal(rng, FINITE_KINDS[b % len(FINITE_KINDS)], n) for b in range(12)])
        if dtype == np.float32:
            x = np.clip(x, -3e38, 3e38)
        x = x.astype(dtype)
        x[5] = np.linspace(-1, 1, n)                       # monotone: stops at once with one all-zero row
        x[7] = np.sin(np.linspace(0, 3.0, n)) + 0.01       # one extremum
        rows, bases, s, rep = _run(P, torch, x, m, RESIDENT_AUTO)
        _check_against_oracle(oracle, x, m, rows, bases, s, "n=%d %s m=%d" % (n, np.dtype(dtype).name, m))
        # the level-by-level engine reports the same summary, entry for entry (knot counts incl. the unevaluated -1 slots)
        rows2, bases2, s2, _ = _run(P, torch, x, m, RESIDENT_OFF)
        for key in ("n_rows", "n_baselines", "stop", "nan_levels", "knot_counts"):
            assert np.array_equal(s[key], s2[key]), key
        for b in range(x.shape[0]):
            assert_bits_equal(rows[b, : s["n_rows"][b]], rows2[b, : s["n_rows"][b]], "vs level-by-level, signal %d" % b)
    if n >= 64:
        # noise and random walks have no plateaus: the resident form completes on its own (RESIDENT_ONLY would refuse otherwise)
        x = np.stack([fuzz_signal(rng, b % 2, n) for b in range(6)])
        rows, bases, s, rep = _run(P, torch, x, 7, RESIDENT_ONLY)
        assert rep == 0
        _check_against_oracle(oracle, x, 7, rows, bases, s, "n=%d resident only" % n)


@pytest.mark.parametrize("n,window", [(300, 8), (300, 64), (2000, 8), (2000, 100), (2000, 1000), (4096, 64), (8000, 8), (8000, 500), (8192, 3000)])
def test_results_do_not_depend_on_the_rank_window(P, torch, oracle, n, window):
    """The by-rank knot arrays hold a window of consecutive segments; a level with more knots takes several passes (knot list,
    values, map per pass).  Tiny windows force many passes on every level — alternating signals have n - 2 knots."""
    from pyitd_amd.engine import RESIDENT_ONLY
    rng = np.random.default_rng(n + window)
    x = np.stack([fuzz_signal(rng, k, n) for k in (0, 1, 4, 6, 0, 1)])
    for dtype, m in ((np.float64, 5), (np.float32, 2)):
        xd = x.astype(dtype)
        rows, bases, s, rep = _run(P, torch, xd, m, RESIDENT_ONLY, window=window)
        assert rep == 0
        _check_against_oracle(oracle, xd, m, rows, bases, s, "n=%d window=%d %s" % (n, window, np.dtype(dtype).name))


def test_resident_goldens(P, torch, oracle):
    """Every reference-generated golden vector short enough for the resident form, through the drop-in class (automatic mode)."""
    import os
    from conftest import golden_cases
    seen = 0
    for name in golden_cases():
        g = load_golden(name)
        if "x" not in g.files or "rows" not in g.files or g["x"].ndim != 1 or g["x"].shape[0] > 8192:
            continue
        x, m = g["x"], int(g["max_iteration"])
        if not np.isfinite(x).all():
            continue
        d = P.ITD()
        rows = d.itd(x, m)
        assert_bits_equal(rows, g["rows"], name)
        seen += 1
    assert seen >= 3 or os.environ.get("PYITD_RESIDENT_MODE") == "1"


def test_nan_in_the_input_follows_the_reference_inside_the_resident_kernel(P, torch, oracle):
    """A NaN in the caller's signal: the reference's first extraction takes its knots from detect_peaks' NaN branch on x (valleys
    away from the NaNs; the NaNs become +inf in place) and detect_peaks(-x) of the mutated array (ITD.py:46-51, 87-95), and
    decomposes the mutated values.  The resident kernel does that in LDS (RESIDENT_ONLY: no repeat); an engine told to reject NaN
    input leaves the kernel instead and the level-by-level engine reports the signal (nan_levels = -2)."""
    from pyitd_amd.engine import LEVEL0_AUTO, NAN_INPUT_REJECT, RESIDENT_AUTO, RESIDENT_OFF, RESIDENT_ONLY
    n, m = 3000, 9
    x = np.stack([sines_noise(n, seed=b, fscale=20.0 + b, dtype=np.float64) for b in range(10)])
    x[1, 1500] = np.nan
    x[2, [0, n - 1]] = np.nan                 # at the ends
    x[3, 63:66] = np.nan                      # a run across a word boundary
    x[4, [511, 512, 1023, 1024, 2047]] = np.nan
    x[5, 700] = np.nan
    x[5, 701] = np.inf                        # next to an infinity
    x[6, 100] = np.inf                        # infinity alone: plain data
    x[7, :40] = 0.0
    x[7, 900] = np.nan                        # NaN input and a leading plateau
    x[8, 1::2] = np.nan                       # every other sample
    for dtype in (np.float64, np.float32):
        xd = x.astype(dtype)
        keep = xd.copy()
        for window in (0, 16):
            rows, bases, s, rep = _run(P, torch, xd, m, RESIDENT_ONLY, window=window)
            assert rep == 0 and (s["nan_levels"] == -1).all()
            _check_against_oracle(oracle, xd, m, rows, bases, s, "NaN input %s window=%d" % (np.dtype(dtype).name, window))
        assert np.array_equal(np.isnan(xd), np.isnan(keep))
        # the level-by-level engine (k_nan_level0 in front of a record-driven level 0) reports the same summary, entry for entry
        _, _, s2, _ = _run(P, torch, xd, m, RESIDENT_OFF)
        for key in ("n_rows", "n_baselines", "stop", "nan_levels", "knot_counts"):
            assert np.array_equal(s[key], s2[key]), key
        for b in range(xd.shape[0]):
            assert int(s["knot_counts"][b, 0]) == len(oracle.knots(xd[b].astype(np.float64)))
    # rejected instead when the engine says so: the kernel leaves, the level-by-level engine flags the signals
    B = x.shape[0]
    xd = torch.from_numpy(x).cuda()
    r = torch.zeros((B, m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, B, 0)
    eng.set_level0_mode(LEVEL0_AUTO)
    eng.set_resident_mode(RESIDENT_AUTO)
    eng.set_nan_input_mode(NAN_INPUT_REJECT)
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), np.float64, n, B, n, m, r.data_ptr(), None, None)
    sm = eng.summary(B)
    assert eng.resident_repeats == 1
    assert sm["nan_levels"].tolist() == [-1, -2, -2, -2, -2, -2, -1, -2, -2, -1]
    eng.close()
    eng = P.Engine(n, B, 0)
    eng.set_resident_mode(RESIDENT_ONLY)
    eng.set_nan_input_mode(NAN_INPUT_REJECT)
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), np.float64, n, B, n, m, r.data_ptr(), None, None)
    with pytest.raises(P.ITDError):
        eng.summary(B)
    eng.close()


@pytest.mark.parametrize("n", [200, 1500, 4096, 8000])
def test_plateaus_follow_the_nan_rules_inside_the_resident_kernel(P, torch, oracle, n):
    """A leading / trailing plateau makes a baseline NaN (0/0, ITD.py:115-116); the reference's stop test then runs detect_peaks
    through its NaN branch, overwrites the NaNs with +inf in place (ITD.py:46-51, 64-68, 400-404) and decomposes the mutated
    baseline on.  The resident kernel does the same in LDS — no repeat (RESIDENT_ONLY), bit-exact rows, baselines, knot counts.
    Infinite samples in the input are plain data (no NaN branch)."""
    from pyitd_amd.engine import RESIDENT_ONLY
    rng = np.random.default_rng(500 + n)
    x = np.stack([sines_noise(n, seed=b, fscale=25.0 + 3 * b, dtype=np.float64) for b in range(10)])
    x[0, : max(2, n // 50)] = 0.0                     # digital silence at the head
    x[1, -max(2, n // 40):] = 0.25                    # trailing plateau
    x[2, :3] = x[2, 3]                                # the shortest leading plateau that matters
    x[3, : n // 3] = 0.0
    x[3, -n // 5:] = 0.0                              # both ends
    x[4] = np.round(x[4] * 4) / 4                     # quantised: plateaus everywhere, usually at the ends too
    x[5, n // 2] = np.inf
    x[6, 1] = -np.inf
    x[7] = fuzz_signal(rng, 5, n)                     # bursts between long constant stretches
    x[8] = fuzz_signal(rng, 2, n)
    for dtype, m in ((np.float64, 9), (np.float32, 4)):
        xd = x.astype(dtype)
        for window in (0, 8):
            rows, bases, s, rep = _run(P, torch, xd, m, RESIDENT_ONLY, window=window)
            assert rep == 0
            _check_against_oracle(oracle, xd, m, rows, bases, s, "plateaus n=%d %s window=%d" % (n, np.dtype(dtype).name, window))


def test_resident_large_batches_and_mixed_stops(P, torch, oracle):
    """70 000 signals of 256 samples in one launch (one workgroup each); signals of a batch stop at different levels."""
    from pyitd_amd.engine import RESIDENT_ONLY
    B, n, m = 70000, 256, 7
    rng = np.random.default_rng(3)
    x = rng.standard_normal((B, n)).astype(np.float32)
    t = np.linspace(0, 1, n, dtype=np.float32)
    x[::7] = np.sin(2 * np.pi * 2.5 * t)[None] + 0.001 * x[::7]     # smoother ones: earlier natural stops
    rows, _, s, rep = _run(P, torch, x, m, RESIDENT_ONLY, keep=False)
    assert rep == 0 and len(set(s["n_rows"].tolist())) >= 3
    pick = np.concatenate([np.arange(0, 64), rng.choice(B, 192, replace=False), [B - 1]])
    for b in pick:
        ref = oracle.itd(x[b], m)
        nr = int(s["n_rows"][b])
        assert nr == ref["rows"].shape[0]
        assert_bits_equal(rows[b, :nr], ref["rows"], "signal %d" % b)
    # reconstruction for all of them: the valid rows sum back to the input
    ok = 0
    for b0 in range(0, B, 10000):
        blk = rows[b0:b0 + 10000]
        nr = s["n_rows"][b0:b0 + 10000]
        mask = np.arange(m + 2)[None, :, None] < nr[:, None, None]
        rec = np.where(mask, blk, 0.0).sum(axis=1)
        ok = max(ok, float(np.abs(rec - x[b0:b0 + 10000].astype(np.float64)).max()))
    assert ok < 1e-12


def test_resident_fuzz_slice(P, torch, oracle):
    """Fixed-seed fuzz: random lengths 3 .. 8192, random max_iteration, all signal families (plateau families go through the
    level-by-level repeat), float32 and float64."""
    from pyitd_amd.engine import RESIDENT_AUTO
    rng = np.random.default_rng(20261004)
    for case in range(60):
        n = int(rng.integers(3, 8193)) if case % 3 else int(rng.choice([3, 64, 512, 513, 4096, 4097, 8192]))
        m = int(rng.integers(0, 21))
        B = int(rng.integers(1, 9))
        dtype = np.float32 if case % 2 else np.float64
        x = np.stack([fuzz_signal(rng, int(rng.integers(0, 8)), n) for _ in range(B)]).astype(dtype)
        if not np.isfinite(x).all():          # float32 overflow of the extreme-magnitude family
            x = np.nan_to_num(x, nan=0.0, posinf=3e38, neginf=-3e38).astype(dtype)
        rows, bases, s, _ = _run(P, torch, x, m, RESIDENT_AUTO)
        _check_against_oracle(oracle, x, m, rows, bases, s, "fuzz case %d (n=%d m=%d B=%d)" % (case, n, m, B))


def test_resident_call_is_graph_capturable(P, torch, oracle):
    """The resident launch is complete in itself (the kernel initialises the states it works on): captured into a hipGraph — also
    as the engine's very first call — and replayed on new data it gives the reference's rows."""
    from pyitd_amd.engine import RESIDENT_ONLY
    B, n, m = 4, 3000, 6
    x_np = np.stack([sines_noise(n, seed=b, fscale=15.0 + b) for b in range(B)])
    x = torch.from_numpy(x_np).cuda()
    rows = torch.zeros((B, m + 2, n), dtype=torch.float64, device="cuda")
    eng = P.Engine(n, B, 0)
    eng.set_resident_mode(RESIDENT_ONLY)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for seed in (11, 12):
        y_np = np.stack([sines_noise(n, seed=seed + b, fscale=9.0 + b) for b in range(B)])
        x.copy_(torch.from_numpy(y_np))
        rows.zero_()
        g.replay()
        torch.cuda.synchronize()
        s = eng.summary(B)
        _check_against_oracle(oracle, y_np, m, rows.cpu().numpy(), None, s, "graph replay seed %d" % seed)
    eng.close()


def test_resident_call_with_device_side_repair(P, torch, oracle):
    """itd_set_valid_flags / itd_set_device_repair under the resident form: a batch of short signals of which two lead with a plateau
    (a NaN baseline) — rows, baselines and validity words are final when the stream has drained, no summary in between; the summary
    afterwards agrees and repeats nothing."""
    from pyitd_amd.engine import LEVEL0_AUTO, RESIDENT_AUTO
    n, m, B = 3000, 6, 12
    x = np.stack([sines_noise(n, seed=200 + b, dtype=np.float64) for b in range(B)])
    x[3, :40] = 0.25
    x[9, :7] = -1.0
    xd = torch.from_numpy(x).cuda()
    rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bases = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    valid = torch.zeros((B,), dtype=torch.int32, device="cuda")
    eng = P.Engine(n, B, 0)
    eng.set_level0_mode(LEVEL0_AUTO)
    eng.set_resident_mode(RESIDENT_AUTO)
    eng.set_valid_flags(valid.data_ptr())
    eng.set_device_repair(True)
    torch.cuda.synchronize()
    eng.decompose_dev(xd.data_ptr(), np.float64, n, B, n, m, rows.data_ptr(), bases.data_ptr(), None)
    torch.cuda.synchronize()                      # (the engine's own stream has drained: no summary was read)
    assert valid.cpu().numpy().tolist() == [1] * B
    r_np, b_np = rows.cpu().numpy(), bases.cpu().numpy()
    s = eng.summary(B)
    assert eng.resident_repeats == 0          # (the resident form runs the reference's NaN rules itself: nothing for either repair to do)
    _check_against_oracle(oracle, x, m, r_np, b_np, s, "resident + device repair")
    eng.close()
