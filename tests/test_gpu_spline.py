"""The FITPACK flavour of the baseline on the GPU (pyitd_amd/csrc/itd_spline.hpp + itd_fitpack.hpp) and its 2-D / ensemble
consumers against the reference-generated vectors (tests/golden/spline) and the scipy-backed oracle.  Knot counts exact; the
floats are expected bit for bit (the restated curfit reproduces scipy's splrep bit for bit on the host and the kernels perform
the same operations in the same order) — asserted to 1e-12 of the signal's scale so that a last-bit libm difference between
the GPU and the host cannot fail the suite; the number of exactly equal values is asserted separately."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from test_oracle_spline import row_cases

pytestmark = pytest.mark.gpu
SPLINE = os.path.join(GOLDEN, "spline")


@pytest.fixture(scope="module")
def P():
    import pyitd_amd
    return pyitd_amd


@pytest.fixture(scope="module")
def so():
    from oracle import spline_oracle
    return spline_oracle


def _close(got, ref, what, tol=1e-12):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, what
    scale = max(1.0, float(np.max(np.abs(ref))))
    err = float(np.max(np.abs(got - ref)))
    assert err <= tol * scale, "%s: max |diff| %.3e (scale %.3e)" % (what, err, scale)
    return float(np.mean(got.view(np.uint64) == ref.view(np.uint64)))


@pytest.mark.parametrize("name", row_cases())
def test_rows_match_reference_goldens(P, name):
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    from pyitd_amd import spline
    exact = _close(spline.itd_baseline_extract_modified(g["x"], solver="serial"), g["baseline"], name)
    assert exact > 0.99, "%s: only %.4f of the values are bit-identical" % (name, exact)
    # the same spline from its second derivatives, parallel in the knots (itd_nak.hpp): equal to rounding, not bit for bit
    _close(spline.itd_baseline_extract_modified(g["x"], solver="parallel"), g["baseline"], name + " (parallel solver)", 1e-10)
    if "meitd_baseline" in g:
        rot, base = P.itd_baseline_extract_spline(g["x"])
        _close(base, g["meitd_baseline"], name + " (MEITD form)")
        _close(rot, g["meitd_rotation"], name + " rotation")
    else:
        with pytest.raises(TypeError):
            P.itd_baseline_extract_spline(g["x"])


def test_crossways_and_ensemble_match_reference(P):
    g = np.load(os.path.join(SPLINE, "image48x64.npz"))
    _close(P.crossways_itd_baseline_extract(g["image"]), g["crossways"], "crossways")
    np.random.seed(int(g["seed"]))                       # the reference draws its noise from numpy's global generator
    _close(P.retrieve_statistical_image_component(g["image"]), g["lowpass"], "ensemble low-pass")
    hl = P.totalextract2d(g["image"], verbose=False)
    assert hl.shape == (2,) + g["image"].shape
    assert np.max(np.abs(hl.sum(axis=0) - g["image"])) < 1e-9     # siftED2D.ipynb cell 4: the two parts sum back


def test_batch_of_image_rows_vs_oracle(P, so):
    """The reference's workload shape: hundreds of 512-sample rows in one call; mixed rows (noisy, smooth = unchanged,
    alternating = equi-spaced knots)."""
    rng = np.random.default_rng(5)
    B, n = 300, 512
    x = rng.integers(0, 256, (B, n)).astype(np.float64)
    x[7] = np.linspace(0, 255, n)
    x[8] = ((-1.0) ** np.arange(n)) * (1 + rng.random(n))
    x[9] = np.round(100 * np.sin(np.arange(n) / 40.0))
    got = P.itd_baseline_extract_rows(x)
    for b in list(range(12)) + [100, 299]:
        _close(got[b], so.baseline(x[b], 10), "row %d" % b)
    assert np.array_equal(got[7], x[7])


def test_long_signal(P, so):
    """One long signal through the same operator: FITPACK's sweep is serial in the knots (one GPU lane: slow but bit-level), the
    moment form runs parallel in the knots (the automatic choice for one long signal) — both against scipy on the host."""
    import time
    from pyitd_amd import spline
    n = 1 << 17
    x = np.cumsum(np.random.default_rng(6).standard_normal(n))
    ref = so.baseline(x, 10)
    _close(spline.itd_baseline_extract_modified(x, solver="serial"), ref, "2^17 samples, serial")
    _close(spline.itd_baseline_extract_modified(x, solver="parallel"), ref, "2^17 samples, parallel", 1e-10)
    _close(P.itd_baseline_extract_modified(x), ref, "2^17 samples, automatic", 1e-10)
    n = 1 << 20
    x = np.cumsum(np.random.default_rng(7).standard_normal(n))
    t0 = time.perf_counter(); ref = so.baseline(x, 10); t_cpu = time.perf_counter() - t0
    P.itd_baseline_extract_modified(x[: 1 << 16])
    t0 = time.perf_counter(); got = P.itd_baseline_extract_modified(x); t_gpu = time.perf_counter() - t0
    _close(got, ref, "2^20 samples, automatic", 1e-10)
    print("2^20-sample signal, FITPACK flavour: %.1f ms on the GPU (host arrays in and out) vs %.1f ms scipy on the host" % (t_gpu * 1e3, t_cpu * 1e3))
    # mixed batch through the parallel solver: rows with fewer than 10 knots come back unchanged, every other row equals the oracle
    rng = np.random.default_rng(8)
    xb = rng.standard_normal((5, 3000))
    xb[2] = np.linspace(0, 1, 3000)
    xb[3] = ((-1.0) ** np.arange(3000)) * (1 + rng.random(3000))       # equally spaced sites: the reference's equi_spaced branch
    xb[4, :2990] = np.sin(np.arange(2990) / 300.0)                     # 3 + a few knots
    got = spline.itd_baseline_extract_rows(xb, solver="parallel")
    for b in range(5):
        _close(got[b], so.baseline(xb[b], 10), "parallel row %d" % b, 1e-10)


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(SPLINE) if f.startswith("meitd_")))
def test_meitd_and_xitd_match_reference(P, name):
    """MEITD / XITD (MEITD.py:395-549) end to end on the GPU operators against the reference's own run."""
    from pyitd_amd import meitd
    g = np.load(os.path.join(SPLINE, name + ".npz"))
    hi, lo, res = meitd.MEITD(g["x"].copy())
    assert hi.shape == g["high"].shape and lo.shape == g["low"].shape        # the same selection decisions
    if hi.size:
        _close(hi, g["high"], name + " high", 1e-10)
    if lo.size:
        _close(lo, g["low"], name + " low", 1e-10)
    _close(res, g["residual"], name + " residual", 1e-10)
    _close(meitd.XITD(g["x"].copy()), g["xitd"], name + " XITD", 1e-10)


def test_crossways_with_more_rows_than_one_grid(P):
    """20 planes x 3300 rows = 66 000 signals per sweep stage: more than one launch's grid.y (65 535) — the stages run in
    chunks; every plane must equal the same plane run on its own, and a NaN anywhere is still reported."""
    from pyitd_amd.itd import _engine_for
    from pyitd_amd import ITDError
    rng = np.random.default_rng(11)
    planes, rows, cols = 20, 3300, 24
    img = rng.integers(0, 256, (planes, rows, cols)).astype(np.float64)
    eng = _engine_for(max(rows, cols))
    full = eng.crossways_host(img, 10)
    for p in (0, 7, 19):
        one = eng.crossways_host(img[p:p + 1], 10)
        assert np.array_equal(full[p].view(np.uint64), one[0].view(np.uint64)), "plane %d" % p
    bad = img.copy()
    bad[0, 5, 3] = np.nan            # in the FIRST chunk of the first stage only
    with pytest.raises(ITDError):
        eng.crossways_host(bad, 10)


# ---- MEITD's operators on device-resident signals (include/pyitd_hip.h, ABI revision 8) ------------------------------------
def _wpe_cases():
    rng = np.random.default_rng(11)
    n = 5000
    t = np.arange(n)
    cases = {
        "noise": rng.standard_normal(n),
        "walk": np.cumsum(rng.standard_normal(n)),
        "quantised (ties in most windows)": np.round(3.0 * rng.standard_normal(n)),
        "constant": np.full(n, 2.5),
        "two values": (t // 3 % 2).astype(np.float64),
        "with NaNs": np.where(rng.random(n) < 0.01, np.nan, rng.standard_normal(n)),
        "three samples": np.array([0.5, -1.0, 0.25]),
        "with infinities": np.where(rng.random(n) < 0.01, np.inf, rng.standard_normal(n)) * np.where(rng.random(n) < 0.5, 1.0, -1.0),
        "exactly 65536 windows": rng.standard_normal(65538),
    }
    for f in sorted(os.listdir(SPLINE)):
        if f.startswith("meitd_"):
            cases[f[:-4]] = np.load(os.path.join(SPLINE, f))["x"]
    return cases


def test_wpe3_bins_are_the_reference_sums_bit_for_bit(P):
    """itd_wpe3_f64 against oracle/meitd_oracle.py (numpy's argsort, numpy.var, cumsum in index order — MEITD.py:79-128): pattern
    populations exact, weighted sums bit for bit up to 65536 windows; the public function equals the oracle's entropy."""
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    from pyitd_amd.engine import DeviceBuffer
    from pyitd_amd.spline import _eng
    for name, x in _wpe_cases().items():
        x = np.ascontiguousarray(x, dtype=np.float64)
        buf = DeviceBuffer(x.nbytes)
        buf.upload(x)
        eng = _eng(len(x), 0)
        w, c = eng.wpe3_dev(buf.ptr, len(x))
        w2, c2, knots = eng.wpe3_dev(buf.ptr, len(x), want_knots=True)       # the knot count rides along in the same launch
        assert knots == eng.count_knots_dev(buf.ptr, len(x)), "%s: knot count %d" % (name, knots)
        assert np.array_equal(c, c2) and (np.array_equal(w.view(np.uint64), w2.view(np.uint64)) or np.isnan(w).any())
        buf.free()
        with np.errstate(all="ignore"):
            wo, co = meitd_oracle.bins(x)
            ref = meitd_oracle.weighted_permutation_entropy(x, order=3, normalize=True)
            got = meitd.weighted_permutation_entropy(x, order=3, normalize=True)
        assert np.array_equal(c, co), "%s: pattern populations %s != %s" % (name, c, co)
        same = (w.view(np.uint64) == wo.view(np.uint64)) | (np.isnan(w) & np.isnan(wo))      # (a NaN's sign and payload are not compared)
        assert same.all(), "%s: weighted sums %s != %s" % (name, w, wo)
        assert got == ref or (np.isnan(got) and np.isnan(ref)), "%s: entropy %r != %r" % (name, got, ref)


def test_wpe3_long_signal_in_segments(P):
    """More than 65536 windows: segments of 4096 windows summed in order, the segments added in order — populations exact, sums equal
    to rounding, the same every time."""
    from oracle import meitd_oracle
    from pyitd_amd.engine import DeviceBuffer
    from pyitd_amd.spline import _eng
    x = np.cumsum(np.random.default_rng(5).standard_normal(300001))
    buf = DeviceBuffer(x.nbytes)
    buf.upload(x)
    eng = _eng(len(x), 0)
    w, c = eng.wpe3_dev(buf.ptr, len(x))
    w2, c2, knots = eng.wpe3_dev(buf.ptr, len(x), want_knots=True)
    assert knots == eng.count_knots_dev(buf.ptr, len(x))
    buf.free()
    wo, co = meitd_oracle.bins(x)
    assert np.array_equal(c, co) and np.array_equal(c, c2)
    assert np.array_equal(w.view(np.uint64), w2.view(np.uint64))
    assert np.max(np.abs(w - wo) / wo) < 1e-12


def test_device_row_operators_equal_their_host_forms(P):
    """itd_count_knots_f64, itd_baseline_extract_spline2_f64, itd_subtract_f64 and itd_copy on device rows against the host entry
    points of the same operators (bit for bit: the same kernels, no transfer in between)."""
    from pyitd_amd.engine import DeviceBuffer
    from pyitd_amd.spline import _eng
    rng = np.random.default_rng(3)
    n = 6000
    x = np.cumsum(rng.standard_normal(n)) + 3.0 * np.sin(np.arange(n) / 17.0)
    eng = _eng(n, 0)
    base_h, rot_h, knots_h, bk_h = eng.spline_extract_host(x[None, :], 0, want_rotation=True, want_baseline_knots=True)
    buf = DeviceBuffer(5 * n * 8)
    px, pb, pr, pd, pz = (buf.ptr + i * n * 8 for i in range(5))
    eng.copy(px, x.ctypes.data, x.nbytes, 1, wait=True)
    assert eng.count_knots_dev(px, n) == int(eng.count_knots_host(x)[0])
    knots, bk = eng.spline_extract_dev(px, n, pb, pr, 0, want_baseline_knots=True)
    assert (knots, bk) == (int(knots_h[0]), int(bk_h[0]))
    assert eng.spline_extract_dev(px, n, pb, None, 0) == knots
    eng.subtract_dev(px, pr, pd, n)                     # x - rotation
    eng.copy(pz, px, n * 8, 2)
    eng.copy(pz, None, (n // 2) * 8, 3)                 # zero the first half of the copy
    got = np.empty((5, n))
    eng.copy(got.ctypes.data, buf.ptr, got.nbytes, 0, wait=True)
    buf.free()
    assert np.array_equal(got[0], x)
    assert np.array_equal(got[1].view(np.uint64), base_h[0].view(np.uint64))
    assert np.array_equal(got[2].view(np.uint64), rot_h[0].view(np.uint64))
    assert np.array_equal(got[3], x - rot_h[0])
    assert np.array_equal(got[4][:n // 2], np.zeros(n // 2)) and np.array_equal(got[4][n // 2:], x[n // 2:])


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(SPLINE) if f.startswith("meitd_")))
def test_meitd_helpers_match_the_reference_flow(P, name, so):
    """retrieve_proper_rotation / determine_if_first_is_proper_rotation (MEITD.py:344-392) on the GPU against the same flow over the
    oracle's operators (oracle.meitd_oracle.CpuWork): the same decisions, the arrays to 1e-10."""
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    x = np.load(os.path.join(SPLINE, name + ".npz"))["x"]
    for wpemax in (0.6, 2.0):
        cw = meitd_oracle.CpuWork(len(x))
        src, rot, base = cw.take(), cw.take(), cw.take()
        cw.upload(x, src)
        proper_ref = meitd._determine(cw, src, rot, base, wpemax)
        r, b, proper = meitd.determine_if_first_is_proper_rotation(x, wpemax)
        assert proper == proper_ref
        _close(r, cw.rows[rot], name + " determine: rotation", 1e-10)
        _close(b, cw.rows[base], name + " determine: baseline", 1e-10)
        cw = meitd_oracle.CpuWork(len(x))
        rot = cw.take()
        cw.upload(x, rot)
        out_ref, proper_ref = meitd._retrieve(cw, rot, wpemax)
        out, proper = meitd.retrieve_proper_rotation(x, wpemax)
        assert proper == proper_ref
        _close(out, cw.rows[out_ref], name + " retrieve", 1e-10)


def test_meitd_keeps_its_arrays_on_the_device(P):
    """One MEITD call uploads the signal once, downloads the components once, and reads back only scalars in between: the number of
    extractions stays below 60 on the golden signals (upstream: up to 292, most of them discarded — see pyitd_amd/meitd.py)."""
    from pyitd_amd import meitd
    x = np.load(os.path.join(SPLINE, "meitd_two_tone_noise.npz"))["x"]
    wk = meitd._work_for(len(x), 0)
    calls = {"extract": 0, "upload": 0, "download": 0}
    orig = {k: getattr(wk, k) for k in calls}

    def counted(k):
        def f(*a, **kw):
            calls[k] += 1
            return orig[k](*a, **kw)
        return f

    for k in calls:
        setattr(wk, k, counted(k))
    try:
        hi, lo, res = meitd.MEITD(x.copy())
    finally:
        for k in calls:
            delattr(wk, k)
    assert calls["upload"] == 1 and calls["download"] <= 3 and calls["extract"] < 60, calls
    assert len(lo) + len(hi) == 21


def _meitd_both_ways(meitd, x, wpemax=0.6):
    """MEITD(x) as one launch and as the host-driven loop (the same operators, one launch each): results and operator counts"""
    got = meitd.MEITD(x.copy(), WPEMAX=wpemax)
    wk = meitd._work_for(len(x), 0)
    last = dict(wk.last)
    calls = {"extract": 0, "probe": 0}
    orig = {k: getattr(wk, k) for k in calls}

    def counted(k):
        def f(*a, **kw):
            calls[k] += 1
            return orig[k](*a, **kw)
        return f

    for k in calls:
        setattr(wk, k, counted(k))
    wk.one_launch = False
    try:
        ref = meitd.MEITD(x.copy(), WPEMAX=wpemax)
    finally:
        wk.one_launch = True
        for k in calls:
            delattr(wk, k)
    return got, ref, last, calls


@pytest.mark.parametrize("name", sorted(f[:-4] for f in os.listdir(SPLINE) if f.startswith("meitd_")))
def test_meitd_as_one_launch_equals_the_host_driven_loop(P, name):
    """itd_meitd_small_f64 (csrc/itd_meitd.hpp): the whole loop of MEITD.py:395-534 in one launch — the same extractions, counts and
    entropy sums as the host-driven loop's launches, the branch taken on the device: the same components bit for bit, the same number
    of extractions and probes, and every logged threshold test re-drawn with numpy agrees (else the call would have gone to the host)."""
    from pyitd_amd import meitd
    x = np.load(os.path.join(SPLINE, name + ".npz"))["x"]
    got, ref, last, calls = _meitd_both_ways(meitd, x)
    assert last.get("one_launch") and last["status"] == 0, last
    assert last["extractions"] == calls["extract"] and last["probes"] == calls["probe"], (last, calls)
    for a, b, what in zip(got, ref, ("high", "low", "residual")):
        assert a.shape == b.shape and np.array_equal(a, b), "%s: %s differs from the host-driven loop" % (name, what)


@pytest.mark.parametrize("n, seed, wpemax", [(1024, 11, 0.6), (2048, 12, 0.45), (4800, 13, 0.6), (5000, 14, 0.7), (8192, 15, 0.6)])
def test_meitd_as_one_launch_on_other_signals(P, n, seed, wpemax):
    """the one-launch loop on other lengths (4800: the last one whose solver arrays fit LDS; 5000, 8192: arrays in global memory) and
    thresholds"""
    from pyitd_amd import meitd
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 1000.0
    x = np.sin(2 * np.pi * 3.0 * t) * (1.0 + 0.5 * np.sin(2 * np.pi * 0.4 * t)) + 0.3 * np.sin(2 * np.pi * 41.0 * t + 1.0) + 0.1 * rng.standard_normal(n)
    got, ref, last, calls = _meitd_both_ways(meitd, x, wpemax)
    assert last.get("one_launch") and last["status"] == 0, last
    assert last["extractions"] == calls["extract"] and last["probes"] == calls["probe"], (last, calls)
    for a, b, what in zip(got, ref, ("high", "low", "residual")):
        assert a.shape == b.shape and np.array_equal(a, b), what


def test_meitd_fuzz_slice(P):
    """200 fixed-seed cases of tools/meitd_fuzz.py: MEITD as one launch against the host-driven loop (one launch per operator, every
    scalar polled from the mapped result words) — the same components bit for bit, the same operator counts, the same errors.  (The
    open-ended runs of the tool are what caught the result words arriving out of order behind a flag word: DESIGN.md section 9.)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("meitd_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "meitd_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    bad, handed, early = mod.run(200, 5, log=lines.append)
    assert bad == 0, "\n".join(lines)


def test_meitd_one_launch_hands_over_what_it_does_not_model(P):
    """a NaN in the signal, fewer than four extrema, and a signal below the solver's threshold: the one-launch loop reports / is not
    taken, and the call behaves like the host-driven loop"""
    from pyitd_amd import meitd
    from pyitd_amd._lib import ITDError
    rng = np.random.default_rng(21)
    x = rng.standard_normal(2000)
    x[777] = np.nan
    with pytest.raises(ITDError) as ei:
        meitd.MEITD(x.copy())
    assert meitd._work_for(len(x), 0).last["status"] == 2
    wk = meitd._work_for(len(x), 0)
    wk.one_launch = False
    try:
        with pytest.raises(ITDError) as ej:
            meitd.MEITD(x.copy())
    finally:
        wk.one_launch = True
    assert ei.value.status == ej.value.status
    ramp = np.linspace(-1.0, 2.0, 1500)
    hi, lo, res = meitd.MEITD(ramp.copy())
    assert meitd._work_for(len(ramp), 0).last["status"] == 1
    assert not hi.any() and not lo.any() and np.array_equal(res, ramp)
    short = rng.standard_normal(600)
    meitd.MEITD(short.copy())
    assert not meitd._work_for(len(short), 0).one_launch
    got = meitd.MEITD(short.copy(), solver="parallel")
    assert meitd._work_for(len(short), 0, "parallel").last.get("one_launch")
    assert abs((got[0].sum(0) + got[1].sum(0) + got[2]) - short).max() < 1e-9


@pytest.mark.parametrize("n, seed", [(500, 1), (2048, 2), (20000, 3), (70000, 4)])
def test_meitd_on_other_signals_matches_the_flow_over_the_oracles_operators(P, n, seed):
    """MEITD on signals the goldens do not cover (lengths from 500 samples to beyond the entropy's exact-order limit of 65536 windows):
    the GPU run and the same control flow over the oracle's operators (oracle.meitd_oracle.CpuWork) keep the same components, to 1e-9."""
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 1000.0
    x = np.sin(2 * np.pi * 3.0 * t) * (1.0 + 0.5 * np.sin(2 * np.pi * 0.4 * t)) + 0.3 * np.sin(2 * np.pi * 41.0 * t + 1.0) + 0.1 * rng.standard_normal(n)
    hi, lo, res = meitd.MEITD(x.copy())
    saved = meitd._work_for
    meitd._work_for = lambda nn, device=0, solver="auto": meitd_oracle.CpuWork(nn)
    try:
        hi2, lo2, res2 = meitd.MEITD(x.copy())
    finally:
        meitd._work_for = saved
    assert hi.shape == hi2.shape and lo.shape == lo2.shape, "components %s + %s against %s + %s" % (hi.shape, lo.shape, hi2.shape, lo2.shape)
    if hi.size:
        _close(hi, hi2, "high", 1e-9)
    if lo.size:
        _close(lo, lo2, "low", 1e-9)
    _close(res, res2, "residual", 1e-9)
    assert np.max(np.abs(hi.sum(0) + lo.sum(0) + res - x)) < 1e-9 * max(1.0, np.max(np.abs(x)))     # the components add up to the signal


def test_meitd_early_returns(P):
    """Signals with fewer than 4 extrema (MEITD.py:411-413: zeros, zeros, the signal itself), a short random one, and XITD of both:
    the GPU run equals the flow over the oracle's operators."""
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    rng = np.random.default_rng(9)
    for name, x in (("constant", np.full(100, 1.5)), ("ramp", np.linspace(-1.0, 2.0, 300)), ("one bump", np.exp(-np.linspace(-3, 3, 200) ** 2)),
                    ("50 random samples", rng.standard_normal(50))):
        with np.errstate(all="ignore"):
            got = meitd.MEITD(x.copy())
            gx = meitd.XITD(x.copy())
            saved = meitd._work_for
            meitd._work_for = lambda nn, device=0, solver="auto": meitd_oracle.CpuWork(nn)
            saved_wpe = meitd.weighted_permutation_entropy
            meitd.weighted_permutation_entropy = lambda ts, order=3, normalize=False, device=0: meitd_oracle.weighted_permutation_entropy(ts, order, normalize)
            try:
                ref = meitd.MEITD(x.copy())
                rx = meitd.XITD(x.copy())
            finally:
                meitd._work_for = saved
                meitd.weighted_permutation_entropy = saved_wpe
        for a, b, what in zip(got, ref, ("high", "low", "residual")):
            assert np.shape(a) == np.shape(b), "%s: %s" % (name, what)
            if np.size(a):
                _close(a, b, name + " " + what, 1e-9)
        assert gx.shape == rx.shape, name
        ok = np.isfinite(rx)
        assert np.array_equal(np.isfinite(gx), ok) and np.max(np.abs(gx[ok] - rx[ok]), initial=0.0) < 1e-9, name


@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_entropy_of_other_orders_matches_the_reference_expressions(P, order):
    """weighted_permutation_entropy(x, order) for the orders the GPU operators take (itd_wpe3_f64 / itd_wpe_f64) against the
    reference's numpy expressions (oracle/meitd_oracle.py, MEITD.py:79-128): hash populations exact, weighted sums bit for bit (index
    order per hash, up to 65536 windows), the entropy equal.  Tied values within a window keep their index order (numpy's insertion
    sort; an AVX-512 numpy orders ties of rows of 4 or more by its sorting networks: the oracle is asked for the stable order there)."""
    kind = "quicksort" if order <= 3 else "stable"
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    from pyitd_amd.engine import DeviceBuffer
    from pyitd_amd.spline import _eng
    rng = np.random.default_rng(100 + order)
    cases = {"noise": rng.standard_normal(4000), "quantised": np.round(2.0 * rng.standard_normal(4000)), "walk": np.cumsum(rng.standard_normal(9000)),
             "with NaNs": np.where(rng.random(3000) < 0.01, np.nan, rng.standard_normal(3000)), "shortest": rng.standard_normal(order)}
    for name, x in cases.items():
        with np.errstate(all="ignore"):
            ref = meitd_oracle.weighted_permutation_entropy(x, order=order, normalize=True, sort_kind=kind)
            got = meitd.weighted_permutation_entropy(x, order=order, normalize=True)
        assert got == ref or (np.isnan(got) and np.isnan(ref)), "%s, order %d: %r != %r" % (name, order, got, ref)
        if order != 3:
            # the operator's raw output against the reference's per-hash sums
            buf = DeviceBuffer(x.nbytes)
            buf.upload(np.ascontiguousarray(x))
            w, c = _eng(len(x), 0).wpe_dev(buf.ptr, len(x), order)
            buf.free()
            sorted_idx = meitd_oracle._embed(x, order=order).argsort(kind=kind)
            hashval = (sorted_idx * np.power(order, np.arange(order))).sum(1)
            with np.errstate(all="ignore"):
                weights = np.var(np.lib.stride_tricks.sliding_window_view(x, order), 1)
            for h in range(order ** order):
                sel = weights[hashval == h]
                assert c[h] == sel.size, "%s, order %d, hash %d" % (name, order, h)
                if sel.size:
                    r = np.cumsum(sel)[-1]
                    assert w[h] == r or (np.isnan(w[h]) and np.isnan(r)), "%s, order %d, hash %d: %r != %r" % (name, order, h, w[h], r)
    with pytest.raises(ValueError):
        meitd.weighted_permutation_entropy(np.arange(10.0), order=6)


def test_entropy_of_order_4_long_signal_in_segments(P):
    from oracle import meitd_oracle
    from pyitd_amd import meitd
    x = np.cumsum(np.random.default_rng(8).standard_normal(150000))
    ref = meitd_oracle.weighted_permutation_entropy(x, order=4, normalize=True)
    got = meitd.weighted_permutation_entropy(x, order=4, normalize=True)
    assert abs(got - ref) < 1e-12 and meitd.weighted_permutation_entropy(x, order=4, normalize=True) == got


@pytest.mark.parametrize("n", [1024, 3000, 4097, 8192])
def test_one_short_signal_in_one_launch_equals_the_launch_sequence(P, n):
    """A single signal of 1024 .. 8192 samples takes the parallel-in-knots form as ONE launch of one workgroup (itd_nak.hpp:
    k_nak_small: MEITD's per-call latency); the same signal as a member of a batch of two goes through the launch sequence
    (knots, compaction, jobs, values, rows, forward, backward, evaluation, count).  Same expressions, same order: baseline,
    rotation and both knot counts equal bit for bit — noisy, smooth, too few knots for a spline, a NaN in the input."""
    from pyitd_amd import ITDError
    from pyitd_amd.spline import _eng
    rng = np.random.default_rng(n)
    t = np.arange(n) / 100.0
    cases = {"noisy": np.cumsum(rng.standard_normal(n)) + 3.0 * np.sin(t), "two tones": np.sin(2 * t) + 0.3 * np.sin(31 * t + 1.0),
             "every sample a knot": ((-1.0) ** np.arange(n)) * (1 + rng.random(n)), "one extremum": -(t - t[n // 2]) ** 2, "monotone": t ** 3}
    eng = _eng(n, 0)
    for name, x in cases.items():
        base2, rot2, k2, bk2 = eng.spline_extract_host(np.stack([x, x[::-1].copy()]), 0, want_rotation=True, want_baseline_knots=True)
        base1, rot1, k1, bk1 = eng.spline_extract_host(x[None, :], 0, want_rotation=True, want_baseline_knots=True)
        assert (int(k1[0]), int(bk1[0])) == (int(k2[0]), int(bk2[0])), name
        assert np.array_equal(base1[0].view(np.uint64), base2[0].view(np.uint64)), name + ": baseline"
        assert np.array_equal(rot1[0].view(np.uint64), rot2[0].view(np.uint64)), name + ": rotation"
    x = cases["noisy"].copy()
    x[n // 3] = np.nan
    with pytest.raises(ITDError):
        eng.spline_extract_host(x[None, :], 0)


def test_meitd_solver_argument_and_release(P):
    """MEITD(..., solver=): "serial" runs every extraction through FITPACK's own sweep (bit-level against scipy), "auto" / "parallel"
    through the parallel-in-knots form — on the golden signal both select the reference's components; release() frees the rows the
    module keeps between calls."""
    from pyitd_amd import meitd
    g = np.load(os.path.join(SPLINE, "meitd_am.npz"))
    for solver in ("auto", "serial", "parallel"):
        hi, lo, res = meitd.MEITD(g["x"].copy(), solver=solver)
        assert hi.shape == g["high"].shape and lo.shape == g["low"].shape, solver
        _close(res, g["residual"], "residual, solver " + solver, 1e-10)
    assert meitd._work
    meitd.release()
    assert not meitd._work
    hi, lo, res = meitd.MEITD(g["x"].copy())             # allocates again
    assert hi.shape == g["high"].shape
