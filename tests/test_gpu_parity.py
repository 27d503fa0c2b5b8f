"""GPU parity tests proper: the HIP path (through the C ABI) against the golden vectors and the CPU oracle.
Bit-exact on knot indices (integers) AND on the float64 rows (stricter than the 1e-6 the north star allows)."""
import os

import numpy as np
import pytest

from conftest import golden_cases
from helpers import assert_bits_equal, chirp, load_golden, sha, sines_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _finite(name):
    return bool(load_golden(name)["finite"])


@pytest.mark.parametrize("name", golden_cases())
def test_driver_matches_golden(P, name):
    g = load_golden(name)
    dec = P.ITD()
    rows = dec.itd(g["x"], max_iteration=int(g["max_iteration"]))
    assert rows.dtype == np.float64 and rows.shape[0] == int(g["n_rows"])
    assert dec.stop_reason == str(g["stop"])
    assert sha(rows) == str(g["rows_sha256"]), "rows are not bit-identical to the reference"
    b = dec.get_baselines()
    assert tuple(b.shape) == tuple(g["baselines_shape"])
    assert sha(b) == str(g["baselines_sha256"])
    if "rows" in g:
        assert_bits_equal(rows, g["rows"], name)
    if not bool(g["finite"]):
        return   # NaN-path cases: rows/baselines (canonical-NaN hashes above) are the whole record
    # knot indices of every level, bit-exact (level 0 = input, level j = stored baseline j-1)
    levels = [np.asarray(g["x"], dtype=np.float64)] + [b[j] for j in range(b.shape[0])]
    for j, xl in enumerate(levels):
        k = P.detect_knots(xl)
        assert k.dtype == np.int64
        np.testing.assert_array_equal(k, g["knots_L%d" % j], err_msg="%s level %d" % (name, j))
    # the counts the engine saw on the device agree with the reference's lists
    # (on timeout get_baselines() ends with the all-zero row, ITD.py:424: not a level the engine evaluated)
    kc = [int(v) for v in dec.knot_counts if v >= 0]
    want = [len(g["knots_L%d" % j]) for j in range(len(levels) - (1 if dec.stop_reason == "timeout" else 0))]
    assert kc[: len(want)] == want and len(kc) >= len(want)


def test_nan_input_follows_the_reference(P):
    """NaN in the input: the reference runs detect_peaks' NaN branch at level 0 and overwrites the NaNs with +inf (ITD.py:46-51,
    64-68); so does the engine, on a copy (the golden cases nanin_* pin that to the reference's own runs).  The caller's array is
    not written.  An engine told to reject such input raises instead."""
    from oracle import cpu_oracle
    from pyitd_amd.engine import Engine, NAN_INPUT_REJECT
    rng = np.random.default_rng(3)
    n = 70000                                   # many tiles; NaNs on and next to tile boundaries, at both ends, in a run
    x = np.cumsum(rng.standard_normal(n)) * 0.05 + np.sin(np.arange(n) / 11.0)
    for at in (0, 1, 511, 512, 513, 4095, 4096, 30000, 30001, 30002, n - 2, n - 1):
        x[at] = np.nan
    keep = x.copy()
    ref = cpu_oracle.itd(x.copy(), 6)
    d = P.ITD()
    rows = d.itd(x, 6)
    assert np.array_equal(np.isnan(x), np.isnan(keep)), "the caller's array must not be written"
    assert_bits_equal(rows, ref["rows"], "NaN input rows")
    assert_bits_equal(d.get_baselines(), ref["baselines"], "NaN input baselines")
    kc = [int(v) for v in d.knot_counts if v >= 0]      # [0] = the signal's own knots; [j >= 1] = what the reference prints (ITD.py:403)
    want = [int(v) for v in ref["knot_counts"]]
    assert kc[1: 1 + len(want)] == want
    # float32 input, and a batch in which only some signals hold a NaN
    xs = np.stack([np.sin(np.arange(9000) / (5.0 + b)) + 0.2 * rng.standard_normal(9000) for b in range(6)]).astype(np.float32)
    xs[1, 100] = np.nan
    xs[4, [0, 8999]] = np.nan
    out = P.itd_batch(xs, 5, keep_baselines=True)
    for b in range(6):
        r = cpu_oracle.itd(xs[b], 5)
        assert_bits_equal(out["rows"][b, : out["n_rows"][b]], r["rows"], "batch signal %d" % b)
        assert_bits_equal(out["baselines"][b, : out["n_baselines"][b]], r["baselines"], "batch signal %d baselines" % b)
    eng = Engine(n, 1)
    eng.set_nan_input_mode(NAN_INPUT_REJECT)
    assert eng.decompose_host(x, 3)["nonfinite"]
    eng.close()
    np.testing.assert_array_equal(P.detect_peaks(x), cpu_oracle.detect_peaks(x))    # the single-level functions follow it too
    y = np.sin(np.linspace(0, 30, 500))
    y[200] = np.inf          # infinities follow the reference's plain rules (raw differences), then its NaN path
    ref = cpu_oracle.itd(y, 4)
    assert_bits_equal(d.itd(y, 4), ref["rows"], "inf input rows")
    assert_bits_equal(d.get_baselines(), ref["baselines"], "inf input baselines")


@pytest.mark.parametrize("name", ["radio8000_m11", "chirp4096_f32_m3", "edge_int_valued", "edge_zigzag1024",
                                  "edge_n3", "edge_n5_zigzag", "edge_trail_plateau", "edge_denormal"])
def test_single_level_operators(P, oracle, name):
    g = load_golden(name)
    x = np.asarray(g["x"], dtype=np.float64)
    rot, base = P.itd_baseline_extract(x)
    assert sha(rot) == str(g["extract_rot_sha256"])
    assert sha(base) == str(g["extract_base_sha256"])
    np.testing.assert_array_equal(P.detect_peaks(x), oracle.detect_peaks(x))
    np.testing.assert_array_equal(P.matlab_detect_peaks(x), oracle.detect_peaks(x, matlab=True))
    np.testing.assert_array_equal(P.detect_peaks(-x), P.matlab_detect_peaks(x))
    k = g["knots_L0"]
    e = np.concatenate([[0], k, [len(x) - 1]]).astype(np.int64)
    bk = np.zeros(len(e))
    bk[0], bk[-1] = np.mean(x[:2]), np.mean(x[-2:])
    want = oracle.knot_values(x, e)
    got = P.baseline_knot_estimation(bk, x, e)
    assert_bits_equal(got, want, name + " knot values")
    assert P.isin(np.array([1, 5, 9]), np.array([5, 9, 11])).tolist() == [False, True, True]


def test_api_surface_and_errors(P):
    with pytest.raises(AssertionError):
        P.ITD(extrema_detection="spline")
    d = P.ITD()
    with pytest.raises(ValueError):
        d.get_baselines()
    with pytest.raises(ValueError):
        d.get_rotations()
    with pytest.raises(ValueError):
        d.itd(np.zeros(2))
    for bad in (np.zeros((4, 50)), np.zeros((1, 50)), np.zeros((50, 1))):   # the reference's driver raises ValueError on these
        with pytest.raises(ValueError):
            d.itd(bad)
    x = np.sin(np.linspace(0, 40, 1000)) + np.random.default_rng(2).standard_normal(1000)
    rows = d(x, max_iterations=2)           # __call__ (ITD.py:189)
    assert rows.shape == (4, 1000) and d.get_rotations() is rows
    rot, base = P.itd_levels(x, 3)          # north-star form
    assert rot.shape == (3, 1000) and base.shape == (1000,)
    assert_bits_equal(np.vstack([rot, base[None]]), rows, "itd_levels")
    assert_bits_equal(P.itd(x, 2), rows, "free function")
    # out=: the caller's result array, reused across calls (shape >= (max_iteration + 2, n)); the result is a view of it
    buf = np.full((6, 1000), -5.0)
    r2 = d.itd(x, max_iteration=2, out=buf)
    assert r2.base is buf and r2.shape == (4, 1000) and np.all(buf[4:] == -5.0)
    assert_bits_equal(r2, rows, "out= rows")
    r3 = d.itd(x[::-1].copy(), max_iteration=2, out=buf)
    assert r3.base is buf and not np.array_equal(r3, rows)
    for bad_out in (np.empty((3, 1000)), np.empty((6, 999)), np.empty((6, 1000), np.float32), np.empty((6, 2000))[:, ::2]):
        with pytest.raises(ValueError):
            d.itd(x, max_iteration=2, out=bad_out)
    # more rows than the reference's 22-row buffers can hold
    rng = np.random.default_rng(5)
    z = np.where(np.arange(1024) % 2 == 0, -1.0, 1.0) * (1 + 0.001 * np.arange(1024))
    with pytest.raises(IndexError):
        d.itd(z, max_iteration=25)
    assert d.itd(rng.standard_normal(500), max_iteration=25).shape[0] < 22  # natural stop first: fine


@pytest.mark.parametrize("n", [3, 4, 5, 63, 64, 65, 2047, 2048, 2049, 4095, 4097, 6144, 10001, 65536 + 17])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_ragged_sizes_vs_oracle(P, oracle, n, dtype):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) + np.sin(np.arange(n) / 37.0) * 3).astype(dtype)
    for m in (0, 4):
        dec = P.ITD()
        rows = dec.itd(x, max_iteration=m)
        ref = oracle.itd(x, m)
        assert dec.stop_reason == ref["stop"]
        assert_bits_equal(rows, ref["rows"], "n=%d m=%d rows" % (n, m))
        assert_bits_equal(dec.get_baselines(), ref["baselines"], "n=%d m=%d baselines" % (n, m))


def test_f32_denormals_and_extremes(P, oracle):
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(5000) * 1e-41).astype(np.float32)   # float32 subnormals must widen exactly
    assert np.any((x != 0) & (np.abs(x) < np.finfo(np.float32).tiny))
    assert_bits_equal(P.ITD().itd(x, 5), oracle.itd(x, 5)["rows"], "f32 subnormal")
    y = (rng.standard_normal(5000) * 1e37).astype(np.float32)
    assert_bits_equal(P.ITD().itd(y, 5), oracle.itd(y, 5)["rows"], "f32 huge")
    z = rng.standard_normal(5000) * 1e-320
    assert_bits_equal(P.ITD().itd(z, 5), oracle.itd(z, 5)["rows"], "f64 subnormal")


def test_config1_chirp_2p16(P, oracle):
    """BASELINE configs[0]: 2^16 float32 chirp, 4 ITD levels (max_iteration=3)."""
    x = chirp(1 << 16)
    dec = P.ITD()
    rows = dec.itd(x, 3)
    g = load_golden("chirp65536_f32_m3")
    assert sha(rows) == str(g["rows_sha256"])
    assert [int(v) for v in dec.knot_counts[:4]] == [8050, 3884, 1997, 800]


def test_one_million_samples_bit_exact(P, oracle):
    x = sines_noise(1 << 20)
    dec = P.ITD()
    rows = dec.itd(x, 7)
    ref = oracle.itd_lean(x, 7, want_knots=True)
    assert dec.stop_reason == ref["stop"]
    assert_bits_equal(rows, ref["rows"], "2^20 rows")
    b = dec.get_baselines()
    for j in range(1, rows.shape[0]):
        np.testing.assert_array_equal(P.detect_knots(b[j - 1]), ref["knots"][j])
    assert [int(v) for v in dec.knot_counts[: len(ref["knot_counts"])]] == ref["knot_counts"].tolist()


@pytest.mark.parametrize("seed", range(6))
def test_dense_and_plateau_signals_vs_oracle(P, oracle, seed):
    """Dense knots (every sample a knot) force several passes per tile; quantised data makes plateaus and staircases;
    all must stay bit-exact, whatever the tile/pass boundaries hit."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(3000, 9000))
    kind = seed % 3
    if kind == 0:      # zigzag with slowly varying envelope: knot at every interior sample
        x = np.where(np.arange(n) % 2 == 0, -1.0, 1.0) * (1 + 0.3 * np.sin(np.arange(n) / 50.0))
    elif kind == 1:    # coarse quantisation: long plateaus, equal neighbours everywhere
        x = np.round(rng.standard_normal(n).cumsum() / 3.0) / 4.0
        x[0] += 0.125  # no leading plateau (that is the NaN path, tested separately)
    else:              # dense noise followed by a smooth stretch (dense and empty tiles side by side)
        x = np.concatenate([rng.standard_normal(n // 2), np.sin(np.linspace(0, 3, n - n // 2)) * 5])
    for m in (2, 9):
        dec = P.ITD()
        rows = dec.itd(x, max_iteration=m)
        ref = oracle.itd(x, m)
        assert dec.stop_reason == ref["stop"]
        assert_bits_equal(rows, ref["rows"], "seed %d m=%d rows" % (seed, m))
        assert_bits_equal(dec.get_baselines(), ref["baselines"], "seed %d m=%d baselines" % (seed, m))


@pytest.mark.parametrize("lead", [2, 3, 7, 64, 300, 700])
def test_leading_plateau_nan_path_vs_oracle(P, oracle, lead):
    """A signal that starts with a plateau divides by zero on its first segment (ITD.py:115-116); the reference
    carries on through detect_peaks' NaN rules and its NaN -> +inf mutation.  Same here, bit for bit."""
    rng = np.random.default_rng(lead)
    x = np.concatenate([np.full(lead, 0.25), rng.standard_normal(4000 - lead)])
    for dtype in (np.float64, np.float32):
        xx = x.astype(dtype)
        ref = oracle.itd(xx, 6)
        dec = P.ITD()
        rows = dec.itd(xx, 6)
        assert dec.stop_reason == ref["stop"]
        assert not np.isfinite(ref["rows"]).all()          # the case really is a NaN case
        assert_bits_equal(rows, ref["rows"], "lead %d rows" % lead)
        assert_bits_equal(dec.get_baselines(), ref["baselines"], "lead %d baselines" % lead)


def test_config5_real_audio_tiled_2p22(P, oracle):
    """BASELINE configs[4] substitute (the wav files are missing blobs upstream): the reference's own
    8000-sample shortwave clip (PyITD.ipynb cell 2) tiled to 2^22 float32 samples, 10 levels (max_iteration = 9);
    knot indices of every level bit-exact against the oracle, rows too."""
    radio = load_golden("radio8000_input")["x"]
    x = np.resize(radio, 1 << 22).astype(np.float32)
    dec = P.ITD()
    rows = dec.itd(x, max_iteration=9)
    ref = oracle.itd_lean(x, 9, want_knots=True)
    assert dec.stop_reason == ref["stop"] and rows.shape[0] == ref["rows"].shape[0]
    assert_bits_equal(rows, ref["rows"], "config 5 rows")
    b = dec.get_baselines()
    np.testing.assert_array_equal(P.detect_knots(x.astype(np.float64)), ref["knots"][0])
    for j in range(1, rows.shape[0]):
        got = P.detect_knots(b[j - 1])
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, ref["knots"][j], err_msg="level %d" % j)
    assert [int(v) for v in dec.knot_counts[: len(ref["knot_counts"])]] == ref["knot_counts"].tolist()


def _wav_inputs(tmp_path):
    """A stereo 16-bit 48 kHz wav written here from the reference's own clip (the ingestion path always runs) and, when the
    user supplies one, the file named by PYITD_WAV (daveandsimon.wav / girl.wav are missing blobs upstream)."""
    from scipy.io import wavfile
    radio = load_golden("radio8000_input")["x"]
    pcm = np.clip(np.round(radio / np.max(np.abs(radio)) * 30000), -32768, 32767).astype(np.int16)
    path = str(tmp_path / "clip_stereo16.wav")
    wavfile.write(path, 48000, np.stack([pcm, pcm[::-1]], axis=1))
    out = [(path, "generated stereo int16 clip")]
    if os.environ.get("PYITD_WAV"):
        out.append((os.environ["PYITD_WAV"], "user-supplied (PYITD_WAV)"))
    return out


def test_config5_from_wav_files(P, oracle, tmp_path):
    """BASELINE configs[4] through the wav ingestion path of bench.py (--wav / PYITD_WAV): mono / first channel, float32 in
    [-1, 1], numpy.resize to 2^22, 10 levels; knot indices of every level and the rows bit-exact against the oracle."""
    import torch
    import bench
    for path, label in _wav_inputs(tmp_path):
        sr, a = bench.load_wav_mono(path)
        assert a.dtype == np.float32 and a.ndim == 1 and np.max(np.abs(a)) <= 1.0, label
        res = bench.audio_leg(torch, torch.device("cuda", 0), path)
        assert res["knot_indices_bit_exact_every_level"] and res["rows_bit_exact"], (label, res)
        assert res["rows"] >= 2 and res["input"] == os.path.basename(path)


@pytest.mark.parametrize("log2n,cycles", [(18, 3.0), (18, 0.8), (21, 2.5), (21, 40.0)])
def test_very_sparse_knots_over_many_tiles(P, oracle, log2n, cycles):
    """A handful of knots spread over hundreds or thousands of tiles: the neighbours' records are far away, so the
    halo search has to leave its 64-tile windows and walk through the group sums in both directions."""
    n = 1 << log2n
    t = np.arange(n, dtype=np.float64) / n
    x = np.sin(2 * np.pi * cycles * t) + 0.3 * t * t
    for m in (1, 6):
        dec = P.ITD()
        rows = dec.itd(x, max_iteration=m)
        ref = oracle.itd_lean(x, m, want_knots=True)
        assert dec.stop_reason == ref["stop"] and rows.shape[0] == ref["rows"].shape[0]
        assert_bits_equal(rows, ref["rows"], "sparse 2^%d x%g m=%d" % (log2n, cycles, m))
        assert [int(v) for v in dec.knot_counts[: len(ref["knot_counts"])]] == ref["knot_counts"].tolist()


def _zigzag_with_knots(n, knots, seed):
    """Piecewise-linear signal whose level-0 knots are exactly `knots` (strictly alternating slopes, random amplitudes)."""
    rng = np.random.default_rng(seed)
    e = np.array([0] + sorted(knots) + [n - 1])
    v = np.zeros(len(e))
    sign = 1.0
    for k in range(1, len(e)):
        v[k] = v[k - 1] + sign * (0.5 + rng.random()) * (1 + (e[k] - e[k - 1]) / 64.0)
        sign = -sign
    return np.interp(np.arange(n), e, v)


@pytest.mark.parametrize("case", ["edges", "pairs", "every_tile_edge", "second_and_last"])
def test_knots_on_tile_boundaries(P, oracle, case):
    """Knots placed exactly on the first / last sample of 512-sample tiles and next to them: the sample next to a tile
    is itself a knot, the two halo samples come out of the neighbours' records, the packed record positions are 0 / 511,
    tiles in between hold no knot at all."""
    T, n = 512, 512 * 40 + 37
    if case == "edges":
        knots = [T - 1, T, 3 * T - 1, 5 * T, 9 * T - 1, 9 * T, 9 * T + 1, 20 * T, 31 * T - 1]
    elif case == "pairs":
        knots = [k * T + d for k in (2, 7, 8, 15, 33) for d in (-2, -1, 0, 1, 2)]
    elif case == "every_tile_edge":
        knots = sorted({k * T - 1 for k in range(1, 40)} | {k * T for k in range(1, 40, 3)})
    else:
        knots = [1, 2, 3, T - 2, n - 4, n - 3, n - 2]
    x = _zigzag_with_knots(n, knots, seed=len(knots))
    ref0 = oracle.itd_lean(x, 0, want_knots=True)
    assert ref0["knots"][0].tolist() == sorted(knots)        # the construction really puts the knots there
    for dtype in (np.float64, np.float32):
        xx = x.astype(dtype)
        for m in (0, 2, 5):
            dec = P.ITD()
            rows = dec.itd(xx, max_iteration=m)
            ref = oracle.itd_lean(xx, m, want_knots=True)
            assert dec.stop_reason == ref["stop"] and rows.shape[0] == ref["rows"].shape[0]
            assert_bits_equal(rows, ref["rows"], "%s %s m=%d" % (case, np.dtype(dtype).name, m))
            assert [int(v) for v in dec.knot_counts[: len(ref["knot_counts"])]] == ref["knot_counts"].tolist()


def test_baselines_stay_on_the_device_until_asked_for(P, oracle):
    """ITD.itd leaves the baselines on the GPU (half the PCIe traffic of a call); `baselines` / get_baselines() fetch them, and
    nothing that reuses or frees the engine's staging buffer in between may lose them (ITD.py:413-414: the reference keeps them
    on the instance)."""
    rng = np.random.default_rng(77)
    xa, xb = rng.standard_normal(5000), rng.standard_normal(7001).astype(np.float32)
    ra, rb = oracle.itd(xa, 4), oracle.itd(xb, 6)
    a, b = P.ITD(), P.ITD()
    a.itd(xa, 4)
    b.itd(xb, 6)                                    # a's baselines must have been brought home before b's run
    assert_bits_equal(a.get_baselines(), ra["baselines"], "first instance, after a second one ran")
    assert_bits_equal(b.baselines, rb["baselines"], "second instance, attribute form")
    a.itd(xa, 4)
    P.itd_baseline_extract(rng.standard_normal(300000))   # grows (replaces) the cached engine
    assert_bits_equal(a.get_baselines(), ra["baselines"], "after the engine was replaced")
    a.itd(xa, 4)
    P.release_engines()
    assert_bits_equal(a.get_baselines(), ra["baselines"], "after release_engines()")
    c = P.ITD()
    c.itd(xb, 6)
    del c                                           # an instance that never asked: nothing to fetch, nothing to leak
    a.itd(xa, 4)
    assert_bits_equal(a.get_baselines(), ra["baselines"], "plain use")
    a.baselines = None
    with pytest.raises(ValueError):
        a.get_baselines()


def test_integer_ratio_equals_the_full_division_bit_for_bit():
    """The knot spacings' ratio of ITD.py:107 — two int64 differences true-divided in float64 — is formed on the GPU by the division's
    own instruction sequence without its range scaling and special-case fix-up (itd_kernels.hpp: int_ratio).  Exhaustively equal to
    the compiler's full division for every pair up to 2048 and for 2^30 random pairs below 2^31."""
    import ctypes
    from pyitd_amd import _lib
    L = _lib.load()
    bad = ctypes.c_int64(-1)
    assert L.itd_debug_int_ratio_check(0, 2048, ctypes.byref(bad)) == 0
    assert bad.value == 0
