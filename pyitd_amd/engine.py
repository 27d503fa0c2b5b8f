"""Thin Python handle over the C-ABI engine (include/pyitd_hip.h).

Python never touches samples: it marshals pointers, sizes, a stream and the level count.
Device memory for the device-resident entry points is whatever the caller owns (torch tensors,
hipMalloc'd buffers, ...) — only raw pointers cross the boundary.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import ITDError, MAX_ROWS

STOP_NATURAL, STOP_TIMEOUT = 0, 1
DETECT_KNOTS, DETECT_VALLEYS, DETECT_PEAKS = 0, 1, 2
ITD_OK, ITD_ERR_INVALID_ARG, ITD_ERR_NONFINITE = 0, 1, 6
LEVEL0_AUTO, LEVEL0_RECORDS, LEVEL0_FUSED = 0, 1, 2
TIME_EXTRACT, TIME_EXTRACT_L0, TIME_EXTRACT_FINAL, TIME_DECOMPOSE, TIME_SCAN0, TIME_KF_APPLY, TIME_KF_KNOTS = 0, 1, 2, 3, 4, 5, 6
NAN_INPUT_FOLLOW, NAN_INPUT_REJECT = 0, 1
RESIDENT_AUTO, RESIDENT_OFF, RESIDENT_ONLY = 0, 1, 2
SPLINE_AUTO, SPLINE_SERIAL, SPLINE_PARALLEL = 0, 1, 2
FUSE_AUTO, FUSE_OFF, FUSE_ONLY = 0, 1, 2


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """One engine = one GPU + one workspace sized for (max_n, max_batch).  Not thread-safe."""

    def __init__(self, max_n, max_batch=1, device=0):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        rc = self._L.itd_engine_create(ctypes.byref(h), int(device), int(max_n), int(max_batch))
        if rc:
            raise ITDError(rc, "itd_engine_create(max_n=%d, max_batch=%d, device=%d)" % (max_n, max_batch, device))
        self._h = h
        self.max_n, self.max_batch, self.device = int(max_n), int(max_batch), int(device)
        mode = os.environ.get("PYITD_LEVEL0_MODE")      # diagnostic override: run a whole test suite in one level-0 mode
        if mode:
            self.set_level0_mode(int(mode))
        mode = os.environ.get("PYITD_FUSE_MODE")        # the fused sparse levels (FUSE_*)
        if mode:
            self.set_fuse_mode(int(mode))
        mode = os.environ.get("PYITD_FUSE_LEVEL")       # the first fused level (diagnostic sweeps)
        if mode:
            self.set_fuse_level(int(mode))
        mode = os.environ.get("PYITD_FUSE_CAP")         # capped fused levels (a whole suite with the fused levels cut at this level)
        if mode:
            self.set_fuse_cap(int(mode))
        mode = os.environ.get("PYITD_FUSE_GROUP")       # chunks of a batch per knot side of the fused levels (sweeps)
        if mode:
            self.set_fuse_group(int(mode))
        mode = os.environ.get("PYITD_FUSE_RANGE")       # tiles per knot-side workgroup of the fused levels (sweeps; 16 / 32 / 64)
        if mode:
            self.set_fuse_range(int(mode))
        mode = os.environ.get("PYITD_FUSE_MIN")         # samples per launch sequence from which FUSE_AUTO fuses (tests: 65536)
        if mode:
            self.set_fuse_min_samples(int(mode))
        mode = os.environ.get("PYITD_BATCH_PIPELINE")   # fused batches: 1 = pipelined instead of rotating chunks (A/B runs)
        if mode:
            self.set_batch_pipeline(int(mode))
        mode = os.environ.get("PYITD_RESIDENT_MODE")    # and for the one-workgroup form of short signals (RESIDENT_*)
        if mode:
            self.set_resident_mode(int(mode))

    def close(self):
        if getattr(self, "_h", None):
            self._L.itd_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc and rc not in allow:
            raise ITDError(rc, self._L.itd_last_error(self._h).decode())
        return rc

    @property
    def workspace_bytes(self):
        return self._L.itd_engine_workspace_bytes(self._h)

    # ---- device-resident path ---------------------------------------------------------------
    def decompose_dev(self, x_ptr, dtype, n, batch, x_stride, max_iteration, rows_ptr, baselines_ptr=None,
                      stream=None):
        """Enqueue a decomposition of device data (no host sync).  Pointers are ints."""
        f = self._L.itd_decompose_f32 if np.dtype(dtype) == np.float32 else self._L.itd_decompose_f64
        self._check(f(self._h, x_ptr, n, batch, x_stride, max_iteration, rows_ptr, baselines_ptr, stream))

    def summary(self, batch):
        n_rows = np.zeros(batch, np.int32)
        n_b = np.zeros(batch, np.int32)
        stop = np.zeros(batch, np.int32)
        kc = np.zeros((batch, MAX_ROWS + 1), np.int64)
        nanlv = np.zeros(batch, np.int32)
        self._check(self._L.itd_get_summary(self._h, _np_ptr(n_rows), _np_ptr(n_b), _np_ptr(stop), _np_ptr(kc),
                                            _np_ptr(nanlv)))
        return {"n_rows": n_rows, "n_baselines": n_b, "stop": stop, "knot_counts": kc, "nan_levels": nanlv}

    def set_timing(self, max_decompositions, stride=1):
        """Record hipEvent pairs around the extraction launches of the next `max_decompositions` runs (0 = off);
        only every `stride`-th run is instrumented."""
        self._check(self._L.itd_set_kernel_timing_stride(self._h, int(stride)))
        self._check(self._L.itd_set_kernel_timing(self._h, int(max_decompositions)))

    def set_nan_input_mode(self, mode):
        """NAN_INPUT_FOLLOW (default: a NaN in the input is treated as the reference treats it, ITD.py:46-51) or NAN_INPUT_REJECT."""
        self._check(self._L.itd_set_nan_input_mode(self._h, int(mode)))

    def set_level0_mode(self, mode):
        """LEVEL0_AUTO (fused level 0, record-driven repeat if the input is too smooth), LEVEL0_RECORDS, LEVEL0_FUSED."""
        self._check(self._L.itd_set_level0_mode(self._h, int(mode)))

    def set_fuse_mode(self, mode):
        """FUSE_AUTO (long signals: the sparse levels run on the knot list, the samples take one pass for all of them; a call whose
        verification fails is repeated level by level), FUSE_OFF, FUSE_ONLY (never repeat)."""
        self._check(self._L.itd_set_fuse_mode(self._h, int(mode)))

    def set_valid_flags(self, valid_dev_ptr):
        """valid_dev_ptr: device pointer to int32[batch] (0 / None = off): behind every later decomposition's last launch the engine
        writes 1 where a signal's rows are final, 0 where `summary()` still has a level-by-level repeat to run (include/pyitd_hip.h)."""
        self._check(self._L.itd_set_valid_flags(self._h, ctypes.c_void_p(int(valid_dev_ptr or 0))))

    def set_device_repair(self, on=True):
        """The level-by-level repeat of a refused optimistic form enqueued by the engine itself, guarded per signal on the device:
        the rows are final when the stream has drained (no `summary()` needed in between; graph-capturable)."""
        self._check(self._L.itd_set_device_repair(self._h, 1 if on else 0))

    @property
    def device_repairs(self):
        return int(self._L.itd_get_device_repairs(self._h))

    def set_timing_mode(self, mode):
        """0: every launch class carries events; 1: only each decomposition's level-0 launch (step_periods())."""
        self._check(self._L.itd_set_kernel_timing_mode(self._h, int(mode)))

    def kernel_timing_samples(self, which=TIME_EXTRACT, cap=4096):
        """The individual durations (ms) of the recorded launches of class `which`, in launch order."""
        buf = (ctypes.c_double * cap)()
        cnt = ctypes.c_int32(0)
        self._check(self._L.itd_get_kernel_timing_samples(self._h, which, buf, cap, ctypes.byref(cnt)))
        return np.array(buf[: min(cnt.value, cap)], dtype=np.float64)

    def step_periods(self, cap=4096):
        """ms from one recorded decomposition's first launch to the next one's (set_timing_mode(1), back-to-back calls)."""
        buf = (ctypes.c_double * cap)()
        cnt = ctypes.c_int32(0)
        self._check(self._L.itd_get_step_periods(self._h, buf, cap, ctypes.byref(cnt)))
        return np.array(buf[: min(cnt.value, cap)], dtype=np.float64)

    def set_fuse_range(self, tiles):
        """Tiles per knot-side workgroup of the fused levels: 64, 32, 16, or 0 = automatic (halved after a call whose lists outgrew them)."""
        self._check(self._L.itd_set_fuse_range(self._h, int(tiles)))

    def debug_kf_fault(self, kind, level=3, where=0, slot=0, delta=1):
        """Tests only: arm one fault in what the fused levels' knot side hands to their sample pass (include/pyitd_hip.h:
        itd_debug_kf_fault); kind < 0 disarms."""
        self._check(self._L.itd_debug_kf_fault(self._h, int(kind), int(level), int(where), int(slot), int(delta)))

    def debug_kf_fault_signal(self, signal):
        """Tests only: the armed fault lands in signal `signal` of the batch (default 0)."""
        self._check(self._L.itd_debug_kf_fault_signal(self._h, int(signal)))

    def set_fuse_cap(self, first_level_not_fused):
        """Capped fused levels: 0 (default) = automatic (a refusal's failing level caps the next calls' fused levels, the rest runs level
        by level), -1 = never, 4 .. max_iteration + 1 = always this cap."""
        self._check(self._L.itd_set_fuse_cap(self._h, int(first_level_not_fused)))

    @property
    def last_fuse_cap(self):
        """The cap of the last decomposition's fused levels as enqueued (0 = none)."""
        return self._L.itd_get_last_fuse_cap(self._h)

    def set_fuse_level(self, first_fused_level):
        """The first fused level, 2 .. max_iteration, or 0 (default) = automatic: 2 where a launch sequence covers >= 2^22 samples, else 3."""
        self._check(self._L.itd_set_fuse_level(self._h, int(first_fused_level)))

    def set_fuse_min_samples(self, samples):
        """FUSE_AUTO fuses calls whose launch sequences cover at least this many samples (default 6 * 2^20: below that every
        launch is bound by its boundary and the fused form has more of them)."""
        self._check(self._L.itd_set_fuse_min_samples(self._h, int(samples)))

    def set_fuse_group(self, chunks):
        """Batches: consecutive chunks that share one knot side of the fused levels (default 4)."""
        self._check(self._L.itd_set_fuse_group(self._h, int(chunks)))

    @property
    def fuse_repeats(self):
        """Calls of this engine that itd_get_summary had to repeat level by level because the fused levels reported a failure."""
        return self._L.itd_get_fuse_repeats(self._h)

    @property
    def last_fuse_level(self):
        """The first fused level of the last decomposition as enqueued (2, 3, ...), 0 = one launch per level / resident form."""
        return self._L.itd_get_last_fuse_level(self._h)

    @property
    def fuse_signal_repairs(self):
        """Single signals of batches that itd_get_summary re-ran on their own (at most one in eight of a batch refused the fused
        form: the rest of the batch kept its fused result)."""
        return int(self._L.itd_get_fuse_signal_repairs(self._h))

    def set_resident_mode(self, mode):
        """RESIDENT_AUTO (signals of <= 8192 samples run as one workgroup each in one launch, the signal resident in LDS; a call
        that meets a non-finite value is repeated level by level), RESIDENT_OFF, RESIDENT_ONLY (never repeat)."""
        self._check(self._L.itd_set_resident_mode(self._h, int(mode)))

    def set_resident_window(self, segments):
        """Knot-to-knot segments of a level the resident form holds in LDS per pass (0 = automatic); results do not depend on it."""
        self._check(self._L.itd_set_resident_window(self._h, int(segments)))

    @property
    def resident_repeats(self):
        """Resident calls of this engine that itd_get_summary had to repeat level by level so far."""
        return self._L.itd_get_resident_repeats(self._h)

    def set_batch_chunk(self, signals_per_chunk):
        """Signals per launch sequence of a batched decomposition (0 = automatic, about 2^24 samples per chunk)."""
        self._check(self._L.itd_set_batch_chunk(self._h, int(signals_per_chunk)))

    def set_batch_streams(self, streams):
        """1 or 2: the chunks of a batched decomposition alternate over that many streams."""
        self._check(self._L.itd_set_batch_streams(self._h, int(streams)))

    def set_batch_pipeline(self, on):
        """Fused batches: 1 = chunk k's pass over the samples runs on the engine's second stream beside chunk k + 1's knot side (behind a
        gate: include/pyitd_hip.h); 0 (default, measured faster on MI355X) = the chunks rotate over the streams."""
        self._check(self._L.itd_set_batch_pipeline(self._h, int(on)))

    def kernel_timing(self, which=TIME_EXTRACT):
        """(total ms, launches) of the recorded launches of class `which` (TIME_*)."""
        ms, cnt = ctypes.c_double(0), ctypes.c_int32(0)
        self._check(self._L.itd_get_kernel_timing(self._h, which, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # ---- numpy in -> numpy out ----------------------------------------------------------------
    def decompose_host(self, x, max_iteration, want_baselines=True, out=None):
        """want_baselines: True = copy them back now; False = none; "lazy" = leave them on the device: the result carries
        `n_baselines` and `fetch_baselines()`, valid until this engine's next host-form decomposition.
        out: a caller-owned float64 C-contiguous array of at least (max_iteration + 2, n) the rows are written into (a loop over
        calls then neither allocates nor releases 8 (max_iteration + 2) n bytes per call)."""
        x = np.ascontiguousarray(x)
        if x.dtype != np.float32:
            x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        R = max_iteration + 2
        if out is None:
            rows = np.empty((R, n), np.float64)
        else:
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.ndim == 2 and out.shape[0] >= R
                    and out.shape[1] == n and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]):
                raise ValueError("out must be a writable C-contiguous float64 array of shape (>= %d, %d)" % (R, n))
            rows = out
        lazy = want_baselines == "lazy"
        self._check(self._L.itd_set_host_keep_baselines(self._h, 1 if lazy else 0))
        bases = np.zeros((R, n), np.float64) if (want_baselines and not lazy) else None
        n_rows, n_b, stop = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        kc = np.zeros(MAX_ROWS + 1, np.int64)
        f = self._L.itd_decompose_host_f32 if x.dtype == np.float32 else self._L.itd_decompose_host_f64
        rc = self._check(f(self._h, _np_ptr(x), n, max_iteration, _np_ptr(rows), _np_ptr(bases), ctypes.byref(n_rows),
                           ctypes.byref(n_b), ctypes.byref(stop), _np_ptr(kc)), allow=(ITD_ERR_NONFINITE,))
        out = {"rows": rows[: n_rows.value], "stop": stop.value, "knot_counts": kc, "nonfinite": rc == ITD_ERR_NONFINITE}
        if lazy:
            nb = n_b.value
            out["n_baselines"] = nb

            def fetch_baselines():
                b = np.zeros((nb, n), np.float64)
                self._check(self._L.itd_get_last_baselines_host(self._h, _np_ptr(b), n, nb))
                return b
            out["fetch_baselines"] = fetch_baselines
        elif want_baselines:
            out["baselines"] = bases[: n_b.value]
        return out

    def baseline_extract_host(self, x, want_knots=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        rot, base = np.empty(n), np.empty(n)
        kn = np.empty(n, np.int64) if want_knots else None
        bk = np.empty(n + 2) if want_knots else None
        m = ctypes.c_int64(0)
        self._check(self._L.itd_baseline_extract_host_f64(self._h, _np_ptr(x), n, _np_ptr(rot), _np_ptr(base),
                                                          _np_ptr(kn), ctypes.byref(m), _np_ptr(bk)))
        if want_knots:
            return rot, base, kn[: m.value].copy(), bk[: m.value + 2].copy()
        return rot, base

    def detect_host(self, x, mode=DETECT_KNOTS):
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        idx = np.empty(n, np.int64)
        m = ctypes.c_int64(0)
        self._check(self._L.itd_detect_host_f64(self._h, _np_ptr(x), n, mode, _np_ptr(idx), ctypes.byref(m)))
        return idx[: m.value].copy()

    def knot_values_host(self, baseline_knots, x, extrema_indices):
        x = np.ascontiguousarray(x, dtype=np.float64)
        e = np.ascontiguousarray(extrema_indices, dtype=np.int64)
        bk = np.ascontiguousarray(baseline_knots, dtype=np.float64)
        self._check(self._L.itd_knot_values_host_f64(self._h, _np_ptr(x), x.shape[0], _np_ptr(e), e.shape[0] - 2,
                                                     _np_ptr(bk)))
        return bk

    def knot_values_dev(self, x_ptr, n, extrema_ptr, m, bk_ptr, stream=None):
        """baseline_knot_estimation on device buffers (itd_knot_values_f64): x float64[n], extrema int32[m+2], bk float64[m+2]
        (bk[1..m] are written); asynchronous on `stream`."""
        self._check(self._L.itd_knot_values_f64(self._h, x_ptr, n, extrema_ptr, m, bk_ptr, stream))

    # ---- cubic-spline baseline variant (include/pyitd_hip.h: itd_baseline_extract_cubic_*, itd_find_extrema_*) ----------
    def find_extrema_host(self, signal):
        """find_extrema (itd_fourier_decomposition.py:17-31): (extrema int64[n] zero padded, idx)."""
        s = np.ascontiguousarray(signal, dtype=np.float64)
        n = s.shape[0]
        ext = np.zeros(n, np.int64)
        idx = ctypes.c_int64(0)
        self._check(self._L.itd_find_extrema_host_f64(self._h, _np_ptr(s), n, _np_ptr(ext), ctypes.byref(idx)))
        return ext, int(idx.value)

    def cubic_extract_host(self, x, extrema=None, idx=0):
        """itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122).  extrema=None: the knots are detected with
        itd.cpp's predicate (itd.cpp:161-168).  Returns (baseline or None when fewer than 2 knots, knots int64, idx)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.shape[0]
        base = np.empty(n)
        got = ctypes.c_int64(0)
        if extrema is None:
            kn = np.zeros(n, np.int64)
            self._check(self._L.itd_baseline_extract_cubic_host_f64(self._h, _np_ptr(x), n, None, 0, _np_ptr(base),
                                                                    ctypes.byref(got), _np_ptr(kn)))
            return (base if got.value >= 2 else None), kn, int(got.value)
        e = np.ascontiguousarray(extrema, dtype=np.int64)
        if e.shape[0] < idx + 1:
            raise ValueError("extrema_input needs idx+1 entries")
        self._check(self._L.itd_baseline_extract_cubic_host_f64(self._h, _np_ptr(x), n, _np_ptr(e), int(idx), _np_ptr(base),
                                                                ctypes.byref(got), None))
        return base, e, int(idx)

    def iq_extract_host(self, z, extrema=None, idx=0):
        """itd_baseline_extract_iq (itd.cpp:58-154): ONE real natural-cubic baseline for complex data — knots where both components
        have an extremum (extrema=None) or the caller's, the operator on the components' mean.  Returns (baseline or None when fewer
        than 2 knots, knots int64, idx)."""
        z = np.ascontiguousarray(z, dtype=np.complex128)
        n = z.shape[0]
        iq = z.view(np.float64)                       # interleaved (re, im)
        base = np.empty(n)
        got = ctypes.c_int64(0)
        if extrema is None:
            kn = np.zeros(n, np.int64)
            self._check(self._L.itd_baseline_extract_iq_host_f64(self._h, _np_ptr(iq), n, None, 0, _np_ptr(base), ctypes.byref(got), _np_ptr(kn)))
            return (base if got.value >= 2 else None), kn, int(got.value)
        e = np.ascontiguousarray(extrema, dtype=np.int64)
        if e.shape[0] < idx + 1:
            raise ValueError("extrema needs idx+1 entries")
        self._check(self._L.itd_baseline_extract_iq_host_f64(self._h, _np_ptr(iq), n, _np_ptr(e), int(idx), _np_ptr(base), ctypes.byref(got), None))
        return base, e, int(idx)

    # ---- batched single-level operators on device buffers (asynchronous; include/pyitd_hip.h: *_batch_f64) ---------------
    def extract_batch_dev(self, x_ptr, n, batch, x_stride, rot_ptr, rot_stride, base_ptr, base_stride, info_ptr=None, stream=None):
        """itd_baseline_extract (ITD.py:79-121) of every row; info int32[batch]: knot count, -1 - count if the row holds a NaN."""
        self._check(self._L.itd_baseline_extract_batch_f64(self._h, x_ptr, n, batch, x_stride, rot_ptr, rot_stride, base_ptr,
                                                           base_stride, info_ptr, stream))

    def detect_batch_dev(self, x_ptr, n, batch, x_stride, mode=DETECT_KNOTS, idx_ptr=None, idx_stride=0, info_ptr=None, stream=None):
        """Knots of every row by predicate `mode`; idx_ptr None: counts only."""
        self._check(self._L.itd_detect_batch_f64(self._h, x_ptr, n, batch, x_stride, mode, idx_ptr, idx_stride, info_ptr, stream))

    def cubic_batch_dev(self, x_ptr, n, batch, x_stride, extrema_ptr, extrema_stride, idx, base_ptr, base_stride, info_ptr=None,
                        stream=None):
        """itd_baseline_extract_fast of every row; extrema_stride 0: one retained knot list for every row (itd.cpp:40-44)."""
        self._check(self._L.itd_baseline_extract_cubic_batch_f64(self._h, x_ptr, n, batch, x_stride, extrema_ptr, extrema_stride,
                                                                 idx, base_ptr, base_stride, info_ptr, stream))

    # ---- the FITPACK flavour of the baseline and its 2-D consumers (itd_baseline_extract_spline_*, itd_crossways_*) --------
    def spline_extract_host(self, x, min_extrema=10, want_rotation=False, want_baseline_knots=False):
        """x[B, n] float64 -> (baseline[B, n], rotation[B, n] or None, knots[B]) (numba_accelerated_itd.py:182-211);
        want_baseline_knots: a fourth item, the knot count of every produced baseline (counted on the device)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        B, n = x.shape
        base = np.empty((B, n))
        rot = np.empty((B, n)) if want_rotation else None
        knots = np.zeros(B, np.int32)
        bk = np.zeros(B, np.int32) if want_baseline_knots else None
        self._check(self._L.itd_baseline_extract_spline_host2_f64(self._h, _np_ptr(x), n, B, int(min_extrema), _np_ptr(base),
                                                                  _np_ptr(rot), _np_ptr(knots), _np_ptr(bk)))
        return (base, rot, knots, bk) if want_baseline_knots else (base, rot, knots)

    def set_spline_solver(self, solver):
        """SPLINE_AUTO (parallel in the knots for few long signals, FITPACK's serial sweep for many rows), SPLINE_SERIAL (bit-level
        against scipy), SPLINE_PARALLEL (not-a-knot moment form: equal to rounding)."""
        self._check(self._L.itd_set_spline_solver(self._h, int(solver)))

    def count_knots_host(self, x, mode=DETECT_KNOTS):
        """Knot counts of x[B, n] (or one signal [n]) under predicate `mode`: no index list is built or copied."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        x2 = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.zeros(x2.shape[0], np.int32)
        self._check(self._L.itd_count_knots_host_f64(self._h, _np_ptr(x2), x2.shape[1], x2.shape[0], int(mode), _np_ptr(out)),
                    allow=(ITD_ERR_NONFINITE,))
        return out

    # ---- MEITD's operators on device-resident signals (pointers are ints; only scalars come back) ----------------------------
    def count_knots_dev(self, x_ptr, n, mode=DETECT_KNOTS, stream=None):
        """The knot count of the n float64 samples at x_ptr (matlab_detect_peaks(x).size + matlab_detect_peaks(-x).size)."""
        out = np.zeros(1, np.int32)
        self._check(self._L.itd_count_knots_f64(self._h, x_ptr, n, 1, n, int(mode), _np_ptr(out), stream), allow=(ITD_ERR_NONFINITE,))
        return int(out[0])

    def wpe3_dev(self, x_ptr, n, want_knots=False, stream=None):
        """(weights[6], windows[6]) of the six order-3 permutation patterns of the n samples at x_ptr (MEITD.py:79-128), in
        numpy.unique's order of the reference's hash values; want_knots: a third item, x's knot count (same synchronisation)."""
        w = np.zeros(6, np.float64)
        c = np.zeros(6, np.int64)
        k = np.zeros(1, np.int32) if want_knots else None
        self._check(self._L.itd_wpe3_f64(self._h, x_ptr, n, _np_ptr(w), _np_ptr(c), _np_ptr(k), stream), allow=(ITD_ERR_NONFINITE,))
        return (w, c, int(k[0])) if want_knots else (w, c)

    def wpe_dev(self, x_ptr, n, order, stream=None):
        """(sums[order^order], windows[order^order]) by the reference's hash value, for any order 2 .. 5 (MEITD.py:79-128)."""
        nh = int(order) ** int(order)
        w = np.zeros(nh, np.float64)
        c = np.zeros(nh, np.int64)
        self._check(self._L.itd_wpe_f64(self._h, x_ptr, n, int(order), _np_ptr(w), _np_ptr(c), stream))
        return w, c

    def spline_extract_dev(self, x_ptr, n, base_ptr, rot_ptr=None, min_extrema=0, want_baseline_knots=False, stream=None):
        """The spline baseline of the n samples at x_ptr into base_ptr (x - baseline into rot_ptr): returns the signal's knot
        count, and the produced baseline's with want_baseline_knots."""
        k = np.zeros(1, np.int32)
        bk = np.zeros(1, np.int32) if want_baseline_knots else None
        self._check(self._L.itd_baseline_extract_spline2_f64(self._h, x_ptr, n, 1, n, int(min_extrema), base_ptr, n, rot_ptr, n,
                                                             _np_ptr(k), _np_ptr(bk), stream))
        return (int(k[0]), int(bk[0])) if want_baseline_knots else int(k[0])

    MEITD_PROBE = np.dtype([("w", np.float64, 6), ("c", np.int32, 6), ("count", np.int32), ("pad", np.int32), ("wpe", np.float64)])

    def meitd_small_dev(self, rows_ptr, n, wpemax, stream=None):
        """MEITD's selection loop on the device rows at rows_ptr (include/pyitd_hip.h: itd_meitd_small_f64; the signal in row 5):
        returns (result[24] = status, high rows, low rows, residual's row, probes, extractions, turns, 0, then the launch's time per
        operator kind in 10 ns units: probes, extractions, counts, copies; the probes' log)."""
        res = np.zeros(24, np.int32)
        log = np.zeros(1024, self.MEITD_PROBE)
        self._check(self._L.itd_meitd_small_f64(self._h, rows_ptr, n, float(wpemax), _np_ptr(res), _np_ptr(log), len(log), stream))
        return res, log[:min(int(res[4]), len(log))]

    def subtract_dev(self, a_ptr, b_ptr, out_ptr, count, stream=None):
        self._check(self._L.itd_subtract_f64(self._h, a_ptr, b_ptr, out_ptr, count, stream))

    def copy(self, dst, src, nbytes, kind, wait=False, stream=None):
        """A copy ordered on the engine's stream: kind 0 device -> host, 1 host -> device, 2 device -> device, 3 zero fill."""
        self._check(self._L.itd_copy(self._h, dst, src, nbytes, int(kind), 1 if wait else 0, stream))

    def crossways_host(self, images, min_extrema=10):
        """images[P, rows, cols] float64 -> crossways_itd_baseline_extract of every plane (siftED2D.ipynb cell 1)."""
        img = np.ascontiguousarray(images, dtype=np.float64)
        P, r, c = img.shape
        out = np.empty((P, r, c))
        self._check(self._L.itd_crossways_host_f64(self._h, _np_ptr(img), P, r, c, int(min_extrema), _np_ptr(out)))
        return out

    def instantaneous_host(self, rotation):
        """(amplitude, phase, frequency) of a proper rotation, float64[n] each (include/pyitd_hip.h: itd_instantaneous_*)."""
        x = np.ascontiguousarray(rotation, dtype=np.float64)
        n = x.shape[0]
        a, p, f = np.empty(n), np.empty(n), np.empty(n)
        self._check(self._L.itd_instantaneous_host_f64(self._h, _np_ptr(x), n, _np_ptr(a), _np_ptr(p), _np_ptr(f)))
        return a, p, f


class DeviceBuffer:
    """hipMalloc'd bytes on one GPU through the C ABI (itd_dev_alloc / itd_dev_copy / itd_dev_free): what the numpy
    entry points use for device-resident staging when the caller brings no allocator of its own (no torch needed)."""

    def __init__(self, nbytes, device=0):
        self._L = _lib.load()
        self.device, self.nbytes = int(device), int(nbytes)
        p = ctypes.c_void_p()
        rc = self._L.itd_dev_alloc(self.device, max(self.nbytes, 1), ctypes.byref(p))
        if rc:
            raise ITDError(rc, "itd_dev_alloc(%d bytes, device %d)" % (nbytes, device))
        self.ptr = p.value

    def upload(self, a, offset=0):
        a = np.ascontiguousarray(a)
        if offset < 0 or offset + a.nbytes > self.nbytes:
            raise ValueError("upload of %d bytes at offset %d into a buffer of %d" % (a.nbytes, offset, self.nbytes))
        rc = self._L.itd_dev_copy(self.device, self.ptr + offset, _np_ptr(a), a.nbytes, 1)
        if rc:
            raise ITDError(rc, "itd_dev_copy(host -> device)")

    def download(self, out, offset=0):
        if not out.flags["C_CONTIGUOUS"] or offset < 0 or offset + out.nbytes > self.nbytes:
            raise ValueError("download of %d bytes at offset %d from a buffer of %d (or a non-contiguous target)" % (out.nbytes, offset, self.nbytes))
        rc = self._L.itd_dev_copy(self.device, _np_ptr(out), self.ptr + offset, out.nbytes, 0)
        if rc:
            raise ITDError(rc, "itd_dev_copy(device -> host)")
        return out

    def free(self):
        if getattr(self, "ptr", None):
            self._L.itd_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
