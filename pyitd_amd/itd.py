"""Host-side mirror of the reference's ITD call surface, backed by the HIP engine.

Same names, argument meaning, return shapes/dtypes and error behaviour as
  ITD.py            class ITD (:123-465), itd_baseline_extract (:79-121), detect_peaks (:33-76), isin (:23-32)
  ITD_numba.py      free function itd(data, max_iteration=22) (:100-136)
  numba_accelerated_itd.py   matlab_detect_peaks (:17-59), baseline_knot_estimation (:167-178)
plus the north-star form  itd_levels(x, n_iters) -> (rotations[n, N], baseline[N]).

Every numeric result comes from libpyitd_hip.so (hand-written gfx950 kernels); there is no CPU
fallback — without the library or a GPU these functions raise.
"""
import numpy

from . import _lib
from .engine import DETECT_KNOTS, DETECT_PEAKS, DETECT_VALLEYS, STOP_TIMEOUT, Engine

_engines = {}
_pending = None    # weak reference to the ITD instance whose baselines are still on the device (at most one per process)


def _flush_pending():
    """Bring the baselines of the last ITD().itd() call to the host before their device buffer is reused or freed."""
    global _pending
    inst = _pending() if _pending is not None else None
    _pending = None
    if inst is not None:
        inst._fetch_baselines()


_MAX_N = 2 ** 31 - 2     # the ABI's limit: int32 knot indices


def _engine_for(n, device=0):
    """Single-signal engines are cached per device and grown by 1.5x (a power of two to start with), never beyond the
    ABI's sample limit."""
    key = int(device)
    eng = _engines.get(key)
    if eng is None or eng.max_n < n:
        cap = 1 << max(12, int(n - 1).bit_length()) if eng is None else max(n, eng.max_n + eng.max_n // 2)
        cap = n if cap > _MAX_N else cap
        if eng is not None:
            _flush_pending()
            eng.close()
        eng = Engine(cap, 1, device)
        _engines[key] = eng
    return eng


_batch_engines = {}


def _batch_engine_for(n, batch, device=0):
    """Batch engines are cached per device and reused while they are large enough (creating one allocates 24 B per
    sample and signal; destroying one synchronises the GPU)."""
    key = int(device)
    eng = _batch_engines.get(key)
    if eng is None or eng.max_n < n or eng.max_batch < batch:
        if eng is not None:
            eng.close()
        eng = Engine(n, batch, device)
        _batch_engines[key] = eng
    return eng


def release_engines():
    """Free every cached engine (their HBM workspaces)."""
    _flush_pending()
    for cache in (_engines, _batch_engines):
        for eng in cache.values():
            eng.close()
        cache.clear()


def _as_signal(data):
    """float32 stays float32 (widened exactly on the GPU, like numpy.asarray(.., float64) at ITD.py:389);
    everything else becomes float64."""
    a = numpy.asarray(data)
    if a.ndim != 1:
        # the reference's driver fails on anything but a 1-D signal: its buffers are [22, len(data)] and the first extraction's
        # result does not broadcast into them (ITD.py:384-389: ValueError from numpy, a typing error under numba)
        raise ValueError("expected a 1-D signal, got an array of shape %s (ITD.py:384-389 cannot broadcast it either)" % (a.shape,))
    if a.dtype != numpy.float32:
        a = numpy.asarray(a, dtype=numpy.float64)
    return numpy.ascontiguousarray(a)


def isin(a, b):
    """ITD.py:23-32 — boolean membership of int64 a[i] in b."""
    return numpy.isin(numpy.asarray(a, dtype=numpy.int64), numpy.asarray(b, dtype=numpy.int64))


def detect_peaks(x, device=0):
    """ITD.py:33-76 — indices i with dx[i] > 0 and dx[i-1] <= 0 (int64, ascending)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    if len(x) < 3:
        raise ValueError("detect_peaks needs at least 3 samples (the reference returns numpy.empty(1) garbage)")
    return _engine_for(len(x), device).detect_host(x, DETECT_VALLEYS)


def matlab_detect_peaks(x, device=0):
    """numba_accelerated_itd.py:17-59 — detect_peaks on the negated differences."""
    x = numpy.asarray(x, dtype=numpy.float64)
    if len(x) < 3:
        raise ValueError("matlab_detect_peaks needs at least 3 samples")
    return _engine_for(len(x), device).detect_host(x, DETECT_PEAKS)


def detect_knots(x, device=0):
    """ITD.py:87-98 — sort(unique(detect_peaks(x) U detect_peaks(-x)))."""
    x = numpy.asarray(x, dtype=numpy.float64)
    return _engine_for(len(x), device).detect_host(x, DETECT_KNOTS)


def baseline_knot_estimation(baseline_knots, x, extrema_indices, device=0):
    """numba_accelerated_itd.py:167-178 — fills baseline_knots[1:-1]; the end knots stay the caller's."""
    x = numpy.asarray(x, dtype=numpy.float64)
    out = _engine_for(len(x), device).knot_values_host(baseline_knots, x, extrema_indices)
    try:
        baseline_knots[:] = out  # the reference writes into its argument and returns it
        return baseline_knots
    except (TypeError, ValueError):
        return out


def itd_baseline_extract(data, device=0):
    """ITD.py:79-121 — (rotation, baseline), both float64[N]."""
    x = numpy.asarray(data, dtype=numpy.float64)
    if len(x) < 3:
        raise ValueError("itd_baseline_extract needs at least 3 samples")
    return _engine_for(len(x), device).baseline_extract_host(x)


class ITD:
    """Intrinsic Time-Scale Decomposition — drop-in for the reference class (ITD.py:123-465)."""

    def __init__(self, extrema_detection: str = "matlab", device: int = 0):
        self.extrema_detection = extrema_detection
        assert self.extrema_detection in (
            "simple",
            "parabol",
            "matlab",
        ), "Only 'simple', 'matlab', and 'parabol' values supported"  # ITD.py:177-181
        self.DTYPE = numpy.float64
        self.device = device
        self.rotations = None
        self._baselines = None
        self._fetch = None          # the last run's baselines are still on the device: fetched when first asked for
        self.knot_counts = None
        self.stop_reason = None

    def __call__(self, S, max_iterations: int = 12):
        # upstream passes a misspelt keyword here (ITD.py:189-190) and cannot run; the intent is clear
        return self.itd(S, max_iteration=max_iterations)

    def itd(self, data, max_iteration: int = 11, out=None):
        """ITD.py:351-432 — rows 0..c-1 are proper rotations, the last row is the residual.
        out (an addition to the reference's signature): a caller-owned float64 array of at least (max_iteration + 2, len(data))
        the result is written into; the returned array is a view of its first rows.  The reference allocates its [22, N] buffers
        per call (ITD.py:384-388); a loop over calls that passes `out` pays for that once."""
        x = _as_signal(data)
        self.DTYPE = numpy.asarray(data).dtype
        n = len(x)
        if n < 3:
            raise ValueError("ITD needs at least 3 samples")
        if max_iteration < 0:
            # counter > max_iteration holds at once: the first pending pair is summed (ITD.py:418-422)
            raise ValueError("max_iteration must be >= 0")
        m = min(int(max_iteration), _lib.MAX_ITERATION)
        self._fetch = None          # this instance's previous baselines are being replaced: nothing to bring home
        _flush_pending()            # another instance's may still sit in the engine's staging buffer: fetch those first
        res = _engine_for(n, self.device).decompose_host(x, m, want_baselines="lazy", out=out)
        if res["nonfinite"]:    # only an engine switched to NAN_INPUT_REJECT gets here
            raise ValueError("the input signal contains NaN")
        if max_iteration > _lib.MAX_ITERATION and res["stop"] == STOP_TIMEOUT:
            # the reference's buffers hold 22 rows (ITD.py:384-385): row 22 does not exist
            raise IndexError("index 22 is out of bounds for axis 0 with size 22")
        self.rotations = res["rows"]
        # the reference stores the baselines on the instance (ITD.py:413-414,423-424); here they stay on the GPU until somebody
        # asks (`baselines`, get_baselines()) or the engine's staging buffer is needed again — half the PCIe traffic of a call
        self._baselines = None
        self._fetch = res["fetch_baselines"]
        global _pending
        import weakref
        _pending = weakref.ref(self)
        self.knot_counts = res["knot_counts"]
        self.stop_reason = "timeout" if res["stop"] == STOP_TIMEOUT else "natural"
        return self.rotations

    def _fetch_baselines(self):
        if self._fetch is not None:
            f, self._fetch = self._fetch, None
            self._baselines = f()

    @property
    def baselines(self):
        self._fetch_baselines()
        return self._baselines

    @baselines.setter
    def baselines(self, value):
        self._fetch = None
        self._baselines = value

    def __del__(self):
        self._fetch = None

    def get_baselines(self):
        if self.baselines is None:
            raise ValueError("No baselines found. Please, run ITD method or its variant first.")
        return self.baselines

    def get_rotations(self):
        if self.rotations is None:
            raise ValueError("No IPR found. Please, run ITD method or its variant first.")
        return self.rotations


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


def itd_batch(x, max_iteration: int = 11, keep_baselines: bool = False, device=None):
    """Decompose a batch of independent signals x[B, N] in one call (device resident, one engine launch sequence
    for the whole batch — the batched form of ITD.itd the reference only has as `numba.prange` over rows,
    siftED2D.ipynb cell 1).

    x: numpy array (staged through hipMalloc'd buffers of the C ABI; torch is not needed) or a torch CUDA tensor (used in
    place), float32 or float64.
    Returns a dict: rows [B, max_iteration+2, N] float64 (numpy, or a torch CUDA tensor when x is one), n_rows [B],
    stop [B] (0 natural / 1 timeout), knot_counts [B, 23], and baselines [B, max_iteration+2, N] + n_baselines [B]
    when keep_baselines.  Row r of signal b is valid for r < n_rows[b].
    """
    if max_iteration < 0 or max_iteration > _lib.MAX_ITERATION:
        raise ValueError("max_iteration must be in 0..20 (the reference's buffers hold 22 rows, ITD.py:384-385)")
    R = max_iteration + 2
    if _is_torch(x):
        import torch
        if not x.is_cuda or x.dim() != 2:
            raise ValueError("expected a 2-D CUDA tensor")
        xt = x if x.dtype in (torch.float32, torch.float64) else x.double()
        if xt.stride(1) != 1:
            xt = xt.contiguous()
        dev = xt.device.index
        B, n = xt.shape
        if n < 3:
            raise ValueError("ITD needs at least 3 samples")
        rows = torch.empty((B, R, n), dtype=torch.float64, device=xt.device)
        bases = torch.zeros((B, R, n), dtype=torch.float64, device=xt.device) if keep_baselines else None
        eng = _batch_engine_for(n, B, dev)
        torch.cuda.synchronize(xt.device)   # the engine runs on its own stream
        eng.decompose_dev(xt.data_ptr(), numpy.float32 if xt.dtype == torch.float32 else numpy.float64, n, B,
                          xt.stride(0), max_iteration, rows.data_ptr(), bases.data_ptr() if keep_baselines else None, None)
        s = eng.summary(B)
        if (s["nan_levels"] == -2).any():
            raise ValueError("an input signal contains NaN")
        out = {"n_rows": s["n_rows"], "stop": s["stop"], "knot_counts": s["knot_counts"], "rows": rows}
        if keep_baselines:
            out["baselines"] = bases
            out["n_baselines"] = s["n_baselines"]
        return out
    from .engine import DeviceBuffer
    a = numpy.asarray(x)
    if a.ndim != 2:
        raise ValueError("expected x[B, N]")
    if a.dtype != numpy.float32:
        a = numpy.asarray(a, dtype=numpy.float64)
    a = numpy.ascontiguousarray(a)
    dev = 0 if device is None else int(device)
    B, n = a.shape
    if n < 3:
        raise ValueError("ITD needs at least 3 samples")
    d_x = DeviceBuffer(a.nbytes, dev)
    d_rows = DeviceBuffer(B * R * n * 8, dev)
    d_bases = DeviceBuffer(B * R * n * 8, dev) if keep_baselines else None
    try:
        d_x.upload(a)
        eng = _batch_engine_for(n, B, dev)
        eng.decompose_dev(d_x.ptr, a.dtype, n, B, n, max_iteration, d_rows.ptr, d_bases.ptr if keep_baselines else None, None)
        s = eng.summary(B)
        if (s["nan_levels"] == -2).any():
            raise ValueError("an input signal contains NaN")
        out = {"n_rows": s["n_rows"], "stop": s["stop"], "knot_counts": s["knot_counts"],
               "rows": d_rows.download(numpy.empty((B, R, n), numpy.float64))}
        if keep_baselines:
            out["baselines"] = d_bases.download(numpy.empty((B, R, n), numpy.float64))
            out["n_baselines"] = s["n_baselines"]
        return out
    finally:
        d_x.free()
        d_rows.free()
        if d_bases is not None:
            d_bases.free()


# ---- cubic-spline baseline variant with externally supplied knots: itd_fourier_decomposition.py / itd.cpp ------------------
def generate_sine_wave(freq, sample_rate, duration):
    """itd_fourier_decomposition.py:11-14 (host numpy: the knot positions depend on the sign of every sample, so the sine is
    generated exactly as the reference generates it)."""
    t = numpy.arange(0, duration, 1 / sample_rate)
    return numpy.sin(2 * numpy.pi * freq * t)


def find_extrema(signal, device=0):
    """itd_fourier_decomposition.py:17-31 — (extrema int64[n], idx): 0, the sign changes signal[i] -> signal[i+1], one
    extrapolated index; the rest of the array is zero like the reference's numpy.zeros."""
    s = numpy.asarray(signal, dtype=numpy.float64)
    if len(s) < 3:
        raise ValueError("find_extrema needs at least 3 samples")
    return _engine_for(len(s), device).find_extrema_host(s)


def itd_baseline_extract_fast(I, extrema_input, idx, device=0):
    """itd_fourier_decomposition.py:49-122 — natural-cubic baseline through the knot values at the caller's knots
    (float64[n]).  The Python twin of itd.cpp:156-239 with compute_extrema = false."""
    x = numpy.asarray(I, dtype=numpy.float64)
    if len(x) < 3:
        raise ValueError("itd_baseline_extract_fast needs at least 3 samples")
    ext = numpy.asarray(extrema_input)
    over = ext[: int(idx) + 1][ext[: int(idx) + 1] >= len(x)]
    if over.size:   # find_extrema's extrapolated last knot can lie beyond the signal: the reference's numpy form raises here
        raise IndexError("index %d is out of bounds for axis 0 with size %d" % (int(over[0]), len(x)))
    base, _, _ = _engine_for(len(x), device).cubic_extract_host(x, extrema_input, int(idx))
    return base


def itd_baseline_extract_cubic(x, device=0, want_knots=False):
    """itd.cpp:156-239 with compute_extrema = true: knots by the file's own 3-point predicate (:161-168), then the same
    spline.  Returns the baseline (the input itself, unchanged, when fewer than 2 knots exist — itd.cpp:170-172 leaves the
    caller's buffer alone); with want_knots also the knot indices."""
    x = numpy.asarray(x, dtype=numpy.float64)
    if len(x) < 3:
        raise ValueError("needs at least 3 samples")
    base, kn, idx = _engine_for(len(x), device).cubic_extract_host(x)
    out = x.copy() if base is None else base
    return (out, kn[:idx].copy()) if want_knots else out


def itd_baseline_extract_iq(data, extrema=None, idx=None, device=0, want_knots=False):
    """itd.cpp:58-154 (`itd_baseline_extract_iq`) — the common baseline of complex (I/Q) data: knots where BOTH components have an
    extremum under the file's 3-point predicate (:74-80) — or the caller's retained knots (`extrema`, `idx`: itd.cpp:40-44) —, then the
    natural-cubic operator of itd_baseline_extract_fast on the components' mean (I + Q) / 2 (:96-108).  Returns the real baseline
    float64[n] (the mean series itself, unchanged, when fewer than 2 knots exist: the file leaves the caller's buffer alone, :85-87);
    with want_knots also (knots, idx)."""
    z = numpy.asarray(data, dtype=numpy.complex128)
    if z.ndim != 1 or len(z) < 3:
        raise ValueError("itd_baseline_extract_iq needs a 1-D signal of at least 3 complex samples")
    eng = _engine_for(len(z), device)
    if extrema is None:
        base, kn, got = eng.iq_extract_host(z)
    else:
        base, kn, got = eng.iq_extract_host(z, extrema, int(idx))
    out = (z.real + z.imag) / 2.0 if base is None else base
    return (out, kn[:got].copy(), got) if want_knots else out


def itd_sine_wrapper(signal, sample_rate, device=0):
    """itd_fourier_decomposition.py:33-47 — peel frequency-governed bands: for each synthetic sine (descending
    frequencies) the sine's zero crossings are the knots of one cubic baseline extraction."""
    problem = numpy.array(signal, dtype=numpy.float64)
    duration = len(problem) / sample_rate
    frequencies = numpy.arange(2, sample_rate // 2 - 1, 96)[::-1]
    products = []
    for k in range(1, frequencies.size):
        sine_wave = generate_sine_wave(frequencies[k], sample_rate, duration)
        extrema, idx = find_extrema(sine_wave, device)
        baseline = itd_baseline_extract_fast(problem, extrema, idx, device)
        rotation = problem - baseline
        products.append(rotation)
        problem = problem - rotation
    products.append(problem)
    return products


def instantaneous(rotation, device=0):
    """Instantaneous amplitude, phase (radians, 0..2 pi per wave) and frequency (cycles per sample) of a proper rotation —
    the time-frequency-energy step README.md:13-21, 41-55 describes and the reference does not implement (definitions: Frei &
    Osorio 2007, single-wave analysis; see pyitd_amd/csrc/itd_tfe.hpp).  Returns three float64[N] arrays."""
    x = numpy.asarray(rotation, dtype=numpy.float64)
    if x.ndim != 1 or len(x) < 3:
        raise ValueError("expected a 1-D rotation with at least 3 samples")
    return _engine_for(len(x), device).instantaneous_host(x)


def itd(data, max_iteration: int = 22, device=0):
    """ITD_numba.py:100-136 — free-function driver (its default of 22 overruns the 22-row buffer upstream;
    the usable range is 0..20)."""
    return ITD(device=device).itd(data, max_iteration=max_iteration)


def itd_levels(x, n_iters, device=0):
    """North-star form: n_iters proper rotations and the final baseline.
    Equals rows[:-1], rows[-1] of ITD().itd(x, max_iteration=n_iters-1)."""
    rows = ITD(device=device).itd(x, max_iteration=int(n_iters) - 1)
    return rows[:-1], rows[-1]
