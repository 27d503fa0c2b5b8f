#!/usr/bin/env python3
"""Generate golden vectors for the ITD hot path from the reference itself.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where the read-only
reference checkout lives at /root/reference.  It imports the reference's own
kernels (`ITD.py:23-121`: isin, detect_peaks, itd_baseline_extract) and the
only runnable driver (`PyITD.ipynb cell 1`, = `ITD.py:351-432` minus the
undefined `S/T` check) under a no-op `numba` stand-in (numba is not installed
here; the bodies are plain numpy, and numba's njit does not enable fast-math
or contraction, so interpreted numpy executes the same IEEE fp64 operations in
the same order).  Outputs are *data only* (inputs + expected outputs) written
to tests/golden/*.npz; no reference source text is stored.

Usage:  python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import types

import numpy as np


# --------------------------------------------------------------------------
# numba stand-in: decorators return the function unchanged
# --------------------------------------------------------------------------
def _install_numba_shim():
    class _Ty:
        """Subscriptable / callable dummy type: float64[:], Tuple((..))(..)"""

        def __init__(self, np_dtype=None):
            self.dtype = np_dtype

        def __getitem__(self, _):
            return self

        def __call__(self, *a, **k):
            return self

    def _decorator(*dargs, **dkw):
        # used both as @njit and @njit(signature, parallel=True)
        if len(dargs) == 1 and callable(dargs[0]) and not isinstance(dargs[0], _Ty) and not dkw:
            return dargs[0]
        return lambda f: f

    nb = types.ModuleType("numba")
    nb.njit = _decorator
    nb.jit = _decorator
    nb.prange = range
    # `numba.boolean[:](...)` appears in a decorator argument (evaluated at import) and
    # `numpy.empty(n, dtype=numba.boolean)` inside isin(): numpy accepts any object
    # carrying a `.dtype` attribute, so one subscriptable dummy serves both uses
    nb.boolean = _Ty(np.dtype(np.bool_))
    nb.int64 = _Ty(np.dtype(np.int64))
    nb.int32 = _Ty(np.dtype(np.int32))
    nb.float64 = _Ty(np.dtype(np.float64))
    nb.float32 = _Ty(np.dtype(np.float32))
    nb.objmode = lambda **k: contextlib.nullcontext()
    nbt = types.ModuleType("numba.types")
    nbt.Tuple = _Ty()
    nbt.float64 = nb.float64
    nbt.int32 = nb.int32
    nbt.int64 = nb.int64
    nb.types = nbt
    sys.modules["numba"] = nb
    sys.modules["numba.types"] = nbt


def load_reference(ref_dir):
    """Return (module ITD.py, class ITD from PyITD.ipynb cell 1, inputarray)."""
    _install_numba_shim()
    sys.path.insert(0, ref_dir)
    try:
        import ITD as ref_itd  # noqa: N811
    finally:
        sys.path.pop(0)

    with open(os.path.join(ref_dir, "PyITD.ipynb")) as f:
        nb_json = json.load(f)
    cell1 = "".join(nb_json["cells"][1]["source"])
    cell2 = "".join(nb_json["cells"][2]["source"])
    ns1 = {}
    exec(compile(cell1, "PyITD.ipynb:cell1", "exec"), ns1)
    ns2 = {}
    exec(compile(cell2, "PyITD.ipynb:cell2", "exec"), ns2)
    return ref_itd, ns1, np.asarray(ns2["inputarray"], dtype=np.float64)


# --------------------------------------------------------------------------
def canon_bytes(a):
    """Bytes of a float array with every NaN replaced by one canonical NaN."""
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a.copy()
        a[np.isnan(a)] = np.float64("nan")
        a.view(np.uint64)[np.isnan(a)] = np.uint64(0x7FF8000000000000)
    return a.tobytes()


def sha(a):
    return hashlib.sha256(canon_bytes(a)).hexdigest()


def run_driver(ns1, x, max_iteration):
    """Run the notebook driver; returns rows, baselines, stop ('natural'|'timeout')."""
    itd = ns1["ITD"]()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        rows = itd.itd(x, max_iteration=max_iteration)
    out = buf.getvalue()
    if "No more decompositions possible" in out:
        stop = "natural"
    elif "Out of time!" in out:
        stop = "timeout"
    else:
        raise RuntimeError("driver produced no stop message: %r" % out)
    return np.array(rows), np.array(itd.get_baselines()), stop


def knots_of(ns1, x64):
    """Reference knot set of a float64 array (ITD.py:87-98), on private copies."""
    dp = ns1["detect_peaks"]
    with np.errstate(all="ignore"):
        a = np.asarray(dp(x64.copy()))
        b = np.asarray(dp(-x64))
    return np.sort(np.unique(np.hstack((a, b)))).astype(np.int64)


def make_case(ns1, ref_mod, name, x, max_iteration, out_dir, full_limit=1 << 13):
    x = np.asarray(x)
    x_in = x.copy()
    # the driver may write NaN->inf into a float64 caller array (ITD.py:41,50,389);
    # hand it a private copy so the stored input stays pristine
    rows, baselines, stop = run_driver(ns1, x.copy(), max_iteration)
    n = x.shape[0]
    rec = {
        "x": x_in,
        "max_iteration": np.int64(max_iteration),
        "n_rows": np.int64(rows.shape[0]),
        "stop": np.array(stop),
        "rows_sha256": np.array(sha(rows)),
        "baselines_shape": np.array(baselines.shape, dtype=np.int64),
        "baselines_sha256": np.array(sha(baselines)),
    }
    finite = bool(np.isfinite(rows).all() and np.isfinite(baselines).all())
    rec["finite"] = np.bool_(finite)
    if n <= full_limit:
        rec["rows"] = rows
        rec["baselines"] = baselines
    else:
        idx = np.unique(np.concatenate([np.arange(0, n, 257), np.arange(64), np.arange(n - 64, n)]))
        rec["sample_idx"] = idx.astype(np.int64)
        rec["rows_sample"] = rows[:, idx]
        rec["baselines_sample"] = baselines[:, idx]
    # per-level knots: level 0 = the input, level j>=1 = stored baseline j-1.
    # Also the reference's module-level single-level operator on the input.
    if finite:
        levels = [np.asarray(x_in, dtype=np.float64)] + [baselines[j] for j in range(baselines.shape[0])]
        counts = []
        for j, xl in enumerate(levels):
            k = knots_of(ns1, xl)
            counts.append(len(k))
            rec["knots_L%d" % j] = k
        rec["knot_counts"] = np.array(counts, dtype=np.int64)
        with np.errstate(all="ignore"):
            r1, b1 = ref_mod.itd_baseline_extract(np.asarray(x_in, dtype=np.float64).copy())
        rec["extract_rot_sha256"] = np.array(sha(r1))
        rec["extract_base_sha256"] = np.array(sha(b1))
        if n <= full_limit:
            rec["extract_rot"] = np.array(r1)
            rec["extract_base"] = np.array(b1)
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
    print("%-28s N=%-7d dtype=%-8s m=%-2d rows=%-2d stop=%-8s finite=%s knots=%s" % (
        name, n, x_in.dtype, max_iteration, rows.shape[0], stop, finite,
        list(rec.get("knot_counts", []))))


def load_cubic_reference(ref_dir):
    """itd_fourier_decomposition.py (natural-cubic baseline with externally supplied knots: find_extrema :17-31,
    itd_baseline_extract_fast :49-122 — the Python twin of itd.cpp:156-239), imported under the same numba stand-in."""
    _install_numba_shim()
    sys.path.insert(0, ref_dir)
    try:
        import itd_fourier_decomposition as ref_cubic
    finally:
        sys.path.pop(0)
    return ref_cubic


def extrema_cpp(x):
    """The knot predicate of itd.cpp:161-168 (compute_extrema = true; strict on the left, non-strict on the right), as
    numpy index arithmetic; zero padded to len(x) like the file's static arrays at first call.  itd.cpp itself is a
    non-compilable fragment: this one-line predicate feeds the reference's own Python twin of the rest."""
    f = ((x[:-2] < x[1:-1]) & (x[1:-1] >= x[2:])) | ((x[:-2] > x[1:-1]) & (x[1:-1] <= x[2:]))
    k = np.flatnonzero(f) + 1
    e = np.zeros(x.shape[0], dtype=np.int64)
    e[: k.shape[0]] = k
    return e, int(k.shape[0])


def make_cubic_case(ref_cubic, name, I, extrema, idx, out_dir, knots_from):
    I = np.asarray(I, dtype=np.float64)
    try:
        with np.errstate(all="ignore"):
            base = np.asarray(ref_cubic.itd_baseline_extract_fast(I.copy(), extrema.copy(), idx))
    except IndexError as ex:
        # find_extrema's extrapolated last index (itd_fourier_decomposition.py:29) can lie beyond the signal: the
        # reference itself fails there (IndexError here, an out-of-bounds read under numba) — not a usable vector
        print("%-28s skipped: the reference raises %s" % (name, ex))
        return
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), I=I, extrema=extrema[: idx + 1].astype(np.int64),
                        idx=np.int64(idx), baseline=base, knots_from=np.array(knots_from))
    print("%-28s N=%-7d idx=%-6d knots from %s, finite=%s" % (name, I.shape[0], idx, knots_from, bool(np.isfinite(base).all())))


def make_spline_cases(ref_dir, out_dir, radio):
    """The FITPACK flavour of the baseline and its 2-D consumers, from the reference's own functions:
    numba_accelerated_itd.itd_baseline_extract_modified (:182-211), MEITD.itd_baseline_extract (:303-338) and the functions
    of siftED2D.ipynb cell 1 (exec'd like PyITD.ipynb's), all under the numba stand-in and with this image's real scipy —
    the third-party dependency the reference calls (interpolate.splrep, numba_accelerated_itd.py:84)."""
    import scipy
    _install_numba_shim()
    sys.path.insert(0, ref_dir)
    try:
        import numba_accelerated_itd as R
        try:
            import MEITD as M
        except Exception as ex:           # noqa
            print("MEITD.py not importable here:", ex)
            M = None
    finally:
        sys.path.pop(0)
    with open(os.path.join(ref_dir, "siftED2D.ipynb")) as f:
        cell1 = "".join(json.load(f)["cells"][1]["source"])
    ns = {}
    exec(compile(cell1, "siftED2D.ipynb:cell1", "exec"), ns)
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(777)
    n = 512
    rows = {
        "pixels512": rng.integers(0, 256, n).astype(np.float64),
        "walk2000": np.cumsum(rng.standard_normal(2000)),
        "sines4096": sines_noise(4096, dtype=np.float64),
        "radio8000": radio,
        "quant600": np.round(rng.standard_normal(600) * 3),
        "alternating100": ((-1.0) ** np.arange(100)) * (1 + rng.random(100)),      # every sample an extremum: equi_spaced
        "few_extrema300": np.sin(np.linspace(0, 7 * np.pi, 300)),                   # 6 extrema < 10: returned unchanged
        "monotone64": np.linspace(0, 1, 64) ** 2,
        "exactly10_200": np.sin(np.linspace(0, 10.5 * np.pi, 200)),
    }
    for name, x in rows.items():
        with np.errstate(all="ignore"):
            base = np.array(R.itd_baseline_extract_modified(x.copy()))
        rec = {"x": x, "baseline": base, "scipy_version": np.array(scipy.__version__)}
        if M is not None:
            try:
                with np.errstate(all="ignore"):
                    r, b = M.itd_baseline_extract(x.copy())
                rec["meitd_rotation"], rec["meitd_baseline"] = np.array(r), np.array(b)
            except Exception as ex:       # noqa
                rec["meitd_error"] = np.array(type(ex).__name__)
        np.savez_compressed(os.path.join(out_dir, "row_" + name + ".npz"), **rec)
        print("%-28s N=%-5d unchanged=%s meitd=%s" % ("spline row_" + name, x.size, bool(base is not None and np.array_equal(base, x)),
                                                     "meitd_baseline" in rec or str(rec.get("meitd_error"))))
    if M is not None:      # MEITD.py:395-549: the selection drivers on top of the operator
        t = np.arange(3000)
        sigs = {"two_tone_noise": np.sin(2 * np.pi * t / 37.0) + 0.6 * np.sin(2 * np.pi * t / 411.0) + 0.2 * rng.standard_normal(3000),
                "walk": np.cumsum(rng.standard_normal(3000)) * 0.3,
                "am": (1 + 0.5 * np.sin(t / 200.0)) * np.sin(t / 9.0) + 0.05 * t / 3000}
        for name, x in sigs.items():
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
                hi, lo, res = M.MEITD(x.copy())
                xi = M.XITD(x.copy())
                wpe = M.weighted_permutation_entropy(x, order=3, normalize=True)
            np.savez_compressed(os.path.join(out_dir, "meitd_" + name + ".npz"), x=x, high=np.array(hi), low=np.array(lo),
                                residual=np.array(res), xitd=np.array(xi), wpe=np.float64(wpe))
            print("spline meitd_%-18s high %d low %d rows, XITD %d rows" % (name, len(hi), len(lo), len(xi)))
    img = rng.integers(0, 256, (48, 64)).astype(np.float64)
    img[10:20] = np.round(np.linspace(0, 255, 64))[None, :]          # smooth rows: fewer than 10 extrema -> unchanged rows
    with np.errstate(all="ignore"):
        cw = np.array(ns["crossways_itd_baseline_extract"](img))
    np.random.seed(20240)
    with np.errstate(all="ignore"):
        low = np.array(ns["retrieve_statistical_image_component"](img))
    np.savez_compressed(os.path.join(out_dir, "image48x64.npz"), image=img, crossways=cw, lowpass=low, seed=np.int64(20240),
                        scipy_version=np.array(scipy.__version__))
    print("spline image48x64: crossways + ensemble low-pass (numpy.random.seed(20240))")


def make_stream_cases(cub, ns1, out_dir, radio):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import stream_oracle as so
    os.makedirs(out_dir, exist_ok=True)

    def ref_fast(W, sel, idx):
        with np.errstate(all="ignore"):
            return np.asarray(cub.itd_baseline_extract_fast(np.array(W, dtype=np.float64), np.array(sel, dtype=np.int64), int(idx)))

    def ref_extract(W):
        with np.errstate(all="ignore"):
            r, b = ns1["itd_baseline_extract"](np.array(W, dtype=np.float64))
        return np.asarray(r), np.asarray(b)

    rng = np.random.default_rng(31044)
    t = np.arange(8 * 1024)
    chans = np.stack([np.cumsum(rng.standard_normal(t.size)) * 0.05 + np.sin(t / 40.0),
                      np.sin(t / 17.0) + 0.3 * rng.standard_normal(t.size),
                      np.round(3 * np.sin(t / 90.0) + rng.standard_normal(t.size)) / 2.0])
    smooth = np.sin(2 * np.pi * t[:4096] / 1500.0) + 0.2 * np.sin(2 * np.pi * t[:4096] / 333.0)   # few extrema per block
    cases = [("walk_L1024_m8", chans[0], 1024, 8, False), ("walk_L1024_m1", chans[0], 1024, 1, False),
             ("radio_L2000_m8", radio, 2000, 8, False), ("radio_L500_m3", radio, 500, 3, False),
             ("smooth_L512_m8", smooth, 512, 8, False), ("oneblock_L1024_m8", chans[1][:1024], 1024, 8, False),
             ("twoblocks_L512_m2", chans[1][:1024], 512, 2, False),
             ("chan3_shared_L1024_m8", chans, 1024, 8, True), ("chan3_own_L2048_m4", chans, 2048, 4, False)]
    for name, x, L, margin, shared in cases:
        base = so.blockwise_cubic(ref_fast, extrema_cpp, x, L, margin, shared)
        np.savez_compressed(os.path.join(out_dir, "cubic_" + name + ".npz"), x=x, block=np.int64(L), margin=np.int64(margin),
                            shared_knots=np.int64(shared), baseline=base)
        print("stream cubic_%-24s shape %s L=%d margin=%d shared=%d finite=%s" % (name, np.shape(x), L, margin, shared,
                                                                                  bool(np.isfinite(base).all())))
    lead = np.concatenate([np.zeros(300), chans[1][:3796]])        # a leading plateau: 0/0 on the first window's end segment
    for name, x, L in (("walk_L1024", chans[0], 1024), ("radio_L1000", radio, 1000), ("smooth_L512", smooth, 512),
                       ("quant3_L512", chans, 512), ("oneblock_L2048", chans[1][:2048], 2048), ("leadzeros_L1024", lead, 1024)):
        rot, base = so.blockwise_linear(ref_extract, x, L)
        np.savez_compressed(os.path.join(out_dir, "linear_" + name + ".npz"), x=x, block=np.int64(L), rotation=rot, baseline=base)
        print("stream linear_%-23s shape %s L=%d" % (name, np.shape(x), L))
    # retained extrema along channels without blocks (itd.cpp:40-44): the knots of channel 0, every channel evaluated on them
    for name, x in (("chan3_8192", chans), ("radio4", np.stack([radio, radio[::-1], np.roll(radio, 123) * 0.5, radio ** 3]))):
        e, idx = extrema_cpp(np.asarray(x[0], dtype=np.float64))
        bases = np.stack([ref_fast(x[c], e[: idx + 1], idx) for c in range(x.shape[0])])
        np.savez_compressed(os.path.join(out_dir, "channels_" + name + ".npz"), x=x, extrema=e[: idx + 1].astype(np.int64),
                            idx=np.int64(idx), baselines=bases)
        print("stream channels_%-21s shape %s idx=%d" % (name, x.shape, idx))


def chirp(n, dtype=np.float32):
    t = np.arange(n, dtype=np.float64) / n
    return np.sin(2 * np.pi * (50 * t + 0.5 * (8000 - 50) * t * t)).astype(dtype)


def sines_noise(n, seed=0, fscale=1.0, dtype=np.float32, fs=48000.0):
    """BASELINE config 2/3 recipe."""
    t = np.arange(n, dtype=np.float64) / fs
    x = np.zeros(n, dtype=np.float64)
    for a, f, p in ((1, 110, 0.1), (0.5, 440, 1.3), (0.25, 1760, 2.1), (0.125, 7040, 0.7)):
        x += a * np.sin(2 * np.pi * (f * fscale) * t + p)
    x += 0.05 * np.random.default_rng(seed).standard_normal(n)
    return x.astype(dtype)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    ref_mod, ns1, radio = load_reference(args.ref)

    def case(name, x, m, **kw):
        make_case(ns1, ref_mod, name, x, m, args.out, **kw)

    # (1) the reference's only recorded known-answer: PyITD.ipynb cell 2/3
    np.savez_compressed(os.path.join(args.out, "radio8000_input.npz"), x=radio)
    for m in (0, 1, 3, 7, 11):
        case("radio8000_m%d" % m, radio, m, **({} if m in (3, 11) else {"full_limit": 0}))
    # (2) ITD.py:491-496 demo signal
    T = np.linspace(0, 2 * np.pi, 400, dtype=np.float64)
    case("demo400_m11", np.sin(20 * T * (1 + 0.2 * T)) + T ** 2 + np.sin(13 * T), 11)
    # class docstring example ITD.py:146-152 (the docstring's shape is wrong; record the truth)
    T = np.linspace(0, 1, 100)
    case("doc100_m11", np.sin(2 * 2 * np.pi * T), 11)
    # (3) chirps (BASELINE config 1 = 2^16, m=3)
    case("chirp4096_f32_m3", chirp(1 << 12), 3)
    case("chirp65536_f32_m3", chirp(1 << 16), 3)
    # (4) sines+noise
    case("sines16384_f32_m7", sines_noise(1 << 14), 7)
    case("sines16384_f64_m7", sines_noise(1 << 14, dtype=np.float64), 7)
    case("sines131072_f32_m7", sines_noise(1 << 17), 7)
    # (5) quantised + tiled radio clip: plateaus (config-5 substitute, small)
    q12 = np.round(radio * 2048.0) / 2048.0
    case("radio_q12_tiled8_m9", np.resize(q12, 8000 * 8), 9)
    q8 = np.round(radio * 64.0) / 64.0
    case("radio_q8_tiled2_m9", np.resize(q8, 16000), 9)
    case("radio_tiled_32768_f32_m9", np.resize(radio, 1 << 15).astype(np.float32), 9)
    # (6) edge cases
    rng = np.random.default_rng(1234)
    case("edge_n3", np.array([0.0, 1.0, 0.5]), 5)
    case("edge_n4", np.array([0.0, 1.0, -1.0, 0.5]), 5)
    case("edge_n5_zigzag", np.array([0.0, 1.0, -1.0, 2.0, 0.0]), 5)
    case("edge_monotone", np.linspace(-1.0, 3.0, 257) ** 3, 5)
    case("edge_constant", np.full(64, 2.5), 5)
    case("edge_zigzag1024", np.where(np.arange(1024) % 2 == 0, -1.0, 1.0) * (1 + 0.001 * np.arange(1024)), 20)
    case("edge_staircase", np.repeat(np.arange(40.0), 7), 5)
    case("edge_plateau_peaks", np.repeat(rng.standard_normal(300), 5), 9)
    case("edge_lead_plateau_nan", np.concatenate([np.ones(5), rng.standard_normal(200)]), 6)
    case("edge_lead_plateau2_nan", np.concatenate([np.zeros(3), np.repeat(rng.standard_normal(100), 3)]), 4)
    case("edge_trail_plateau", np.concatenate([rng.standard_normal(200), np.ones(6)]), 6)
    case("edge_one_extremum", -(np.linspace(-1, 1, 101) ** 2), 5)
    case("edge_two_extrema", np.sin(np.linspace(0, 2 * np.pi, 200)), 5)
    case("edge_noise_m20", rng.standard_normal(5000), 20)
    case("edge_noise_m0", rng.standard_normal(777), 0)
    case("edge_noise_f32_odd", rng.standard_normal(2049).astype(np.float32), 7)
    case("edge_denormal", rng.standard_normal(512) * 1e-310, 5)
    case("edge_large", rng.standard_normal(512) * 1e300, 5)
    case("edge_int_valued", rng.integers(-3, 4, 4000).astype(np.float64), 9)
    # (6b) NaN in the INPUT: detect_peaks' NaN branch at level 0 and the in-place NaN -> +inf mutation (ITD.py:46-51, 64-68);
    #      a generator of their own, so that the cases above keep their draws
    rng_n = np.random.default_rng(987)

    def with_nan(x, at):
        x = np.array(x, copy=True)
        x[at] = np.nan
        return x

    base = np.sin(np.arange(3000) / 7.0) + 0.3 * rng_n.standard_normal(3000)
    case("nanin_mid", with_nan(base[:1500], [700]), 5)
    case("nanin_first_sample", with_nan(base[:600], [0]), 5)
    case("nanin_second_sample", with_nan(base[:600], [1]), 5)
    case("nanin_last_sample", with_nan(base[:600], [599]), 5)
    case("nanin_tile_edges", with_nan(base, [5, 6, 300, 510, 511, 512, 1023, 1025, 1536, 2047, 2048]), 6)
    case("nanin_run", with_nan(base[:2000], list(range(1000, 1010))), 4)
    case("nanin_f32", with_nan(base[:2500].astype(np.float32), [3, 1234, 1235, 2400]), 7)
    case("nanin_with_inf", with_nan(np.where(np.arange(1200) == 400, np.inf, base[:1200]), [800]), 4)

    # (6c) the single-level functions on NaN input: detect_peaks (ITD.py:33-76), matlab_detect_peaks
    #      (numba_accelerated_itd.py:17-59: the NaN branch on the negated differences), itd_baseline_extract (ITD.py:79-121)
    import importlib
    sys.path.insert(0, args.ref)
    try:
        nai = importlib.import_module("numba_accelerated_itd")
    finally:
        sys.path.pop(0)
    rng_h = np.random.default_rng(4711)
    rec = {}
    for c in range(12):
        n = int(rng_h.integers(5, 2600)) if c else 1500
        x = np.sin(np.arange(n) / 5.0) + 0.4 * rng_h.standard_normal(n)
        at = rng_h.integers(0, n, int(rng_h.integers(1, 6)))
        if c == 0:
            at = np.array([0, 1, 511, 512, 513, 1023, 1024, 1498, 1499])
        x[at] = np.nan
        if c % 3 == 1:
            x[int(at[0] + 2) % n] = np.inf
        with np.errstate(all="ignore"):
            rot, base = ns1["itd_baseline_extract"](x.copy())
            rec["x_%d" % c] = x
            rec["valleys_%d" % c] = np.asarray(ns1["detect_peaks"](x.copy()), dtype=np.int64)
            rec["matlab_%d" % c] = np.asarray(nai.matlab_detect_peaks(x.copy()), dtype=np.int64)
            rec["rot_%d" % c] = np.asarray(rot)
            rec["base_%d" % c] = np.asarray(base)
    rec["cases"] = np.int64(12)
    np.savez_compressed(os.path.join(args.out, "helpers_nan_input.npz"), **rec)
    print("helpers_nan_input       12 signals: detect_peaks / matlab_detect_peaks / itd_baseline_extract on NaN input")

    # (7) cubic-spline baseline variant with externally supplied knots: the reference's
    #     itd_baseline_extract_fast fed (a) by its own find_extrema on synthetic sines, as itd_sine_wrapper does
    #     (itd_fourier_decomposition.py:33-47), (b) by itd.cpp's knot predicate on the signal itself
    cub = load_cubic_reference(args.ref)
    cdir = os.path.join(args.out, "cubic")
    os.makedirs(cdir, exist_ok=True)
    rng = np.random.default_rng(4321)
    sr = 8000
    sig = radio[:4000]
    for f in (7.0, 50.0, 440.0, 1234.5, 2000.0, 3990.0):
        s = cub.generate_sine_wave(f, sr, len(sig) / sr)
        e, idx = cub.find_extrema(s)
        make_cubic_case(cub, "cubic_radio4000_sine%g" % f, sig, np.asarray(e), int(idx), cdir, "find_extrema(sine %g Hz)" % f)
    x = np.cumsum(rng.standard_normal(6000))
    s = cub.generate_sine_wave(97.0, 48000, len(x) / 48000)
    e, idx = cub.find_extrema(s)
    make_cubic_case(cub, "cubic_walk6000_sine97", x, np.asarray(e), int(idx), cdir, "find_extrema(sine 97 Hz @48k)")
    for nm, x in (("cubic_detect_radio8000", radio), ("cubic_detect_walk3000", np.cumsum(rng.standard_normal(3000))),
                  ("cubic_detect_quant5000", np.round(rng.standard_normal(5000) * 3) / 2.0),
                  ("cubic_detect_sines16384", sines_noise(1 << 14, dtype=np.float64)),
                  ("cubic_detect_smooth2000", np.sin(np.linspace(0, 9 * np.pi, 2000)) + 0.1 * np.linspace(0, 1, 2000) ** 2)):
        e, idx = extrema_cpp(np.asarray(x, dtype=np.float64))
        make_cubic_case(cub, nm, x, e, idx, cdir, "itd.cpp:161-168 predicate")

    # (8) the FITPACK flavour of the baseline and its 2-D consumers
    make_spline_cases(args.ref, os.path.join(args.out, "spline"), radio)

    # (9) block-wise operation and retained extrema along several channels (itd.cpp:31-44).  The recipe has
    #     no upstream code: oracle/stream_oracle.py states it once over operators that are passed in — here the REFERENCE's own
    #     itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122) and itd_baseline_extract (ITD.py:79-121).
    make_stream_cases(cub, ns1, os.path.join(args.out, "stream"), radio)


if __name__ == "__main__":
    main()
