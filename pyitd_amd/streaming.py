"""Block-wise (streaming) operation for unbounded signals — host handle over itd_stream_* (include/pyitd_hip.h).

The reference gives the recipe in a comment only (itd.cpp:31-38): "use a circular buffer with modulous tracking to rotate
the samples / re-assess extrema in the entire buffer every iteration / use from the last extrema in the first buffer to the
first extrema in the last buffer / set the first and last baseline knots manually to said values / update the j array /
compute only the baseline[i] array for the inner third of the buffer overall / rotate buffers, rinse and repeat", and the
reuse of retained extrema along channels (itd.cpp:40-44).  Everything numeric runs on the GPU inside the library: the ring
of the last three blocks, the window's extrema, the knot selection and the operator (pyitd_amd/csrc/itd_stream.hpp,
itd_engine_batch.inc); this module only hands blocks in and out.  The recipe's exact statement (window geometry, knot selection) is in include/pyitd_hip.h
and DESIGN.md section 7; the GPU tests hold these classes to an independent CPU statement of it.

    st = Stream(4096, kind="linear")             # tier-1 operator, ITD.py:79-121
    for block in blocks:
        out = st.push(block)                     # None for the first block, then (rotation, baseline) of the PREVIOUS block
    last = st.flush()

kind "cubic": itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122) on the extrema from `margin` in front of the
emitted block to margin + 2 behind it (the recipe's literal choice is margin 1); blocks with fewer than 4 such extrema are
returned unchanged (itd.cpp:170-172).  `shared_knots`: channel 0's extrema serve every channel (itd.cpp:40-44).
kind "linear": itd_baseline_extract on the window — away from a couple of knots at the window's ends bit-identical to the
whole-signal operator, so a stream whose blocks hold a few knots each reproduces the whole-signal rows exactly.
"""
import ctypes

import numpy

from . import _lib
from ._lib import ITDError

KINDS = {"cubic": 0, "linear": 1}


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Stream:
    """One stream = `channels` channels in lock step, blocks of `block` samples."""

    def __init__(self, block, channels=1, kind="cubic", margin=8, shared_knots=False, device=0):
        if kind not in KINDS:
            raise ValueError("kind must be 'cubic' or 'linear'")
        if block < 8 or channels < 1 or (kind == "cubic" and margin < 1):
            raise ValueError("block >= 8 samples, channels >= 1, margin >= 1 extremum")
        if kind == "linear" and shared_knots:
            raise ValueError("the tier-1 operator's knots are the signal's own (ITD.py:87-98): shared_knots needs kind='cubic'")
        self._L = _lib.load()
        h = ctypes.c_void_p()
        rc = self._L.itd_stream_create(ctypes.byref(h), int(device), int(block), int(channels), KINDS[kind], int(margin),
                                       1 if shared_knots else 0)
        if rc:
            raise ITDError(rc, "itd_stream_create(block=%d, channels=%d, kind=%s)" % (block, channels, kind))
        self._h = h
        self.block, self.channels, self.kind, self.margin, self.device = int(block), int(channels), kind, int(margin), int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.itd_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise ITDError(rc, self._L.itd_stream_last_error(self._h).decode())

    @property
    def blocks_held(self):
        return self._L.itd_stream_blocks(self._h)

    def reset(self):
        self._check(self._L.itd_stream_reset(self._h))

    def status(self):
        """Synchronises the device; 0, or 2 if some window since the last reset held a NaN."""
        v = ctypes.c_int32(0)
        self._check(self._L.itd_stream_status(self._h, ctypes.byref(v)))
        return v.value

    # ---- device buffers (raw pointers; asynchronous on `stream`) ----------------------------------------------------
    def push_dev(self, block_ptr, in_stride, base_ptr, base_stride, rot_ptr=None, rot_stride=0, stream=None):
        """Store one block of every channel; True if the previous block's baseline (rotation) was written."""
        em = ctypes.c_int32(0)
        self._check(self._L.itd_stream_push_f64(self._h, block_ptr, in_stride, base_ptr, base_stride, rot_ptr, rot_stride,
                                                ctypes.byref(em), stream))
        return bool(em.value)

    def flush_dev(self, base_ptr, base_stride, rot_ptr=None, rot_stride=0, stream=None):
        em = ctypes.c_int32(0)
        self._check(self._L.itd_stream_flush_f64(self._h, base_ptr, base_stride, rot_ptr, rot_stride, ctypes.byref(em), stream))
        return bool(em.value)

    # ---- numpy in -> numpy out (one synchronisation per call) ----------------------------------------------------------
    def _shape_out(self, a, flat):
        return a[0] if flat else a

    def push(self, samples):
        """samples [channels, block] (or [block] for one channel).  Returns None for the first block; afterwards the PREVIOUS
        block's baseline (cubic) or (rotation, baseline) (linear), shaped like the input."""
        x = numpy.ascontiguousarray(samples, dtype=numpy.float64)
        flat = x.ndim == 1
        x2 = x.reshape(1, -1) if flat else x
        if x2.shape != (self.channels, self.block):
            raise ValueError("expected blocks of shape (%d, %d)" % (self.channels, self.block))
        base = numpy.empty_like(x2)
        rot = numpy.empty_like(x2) if self.kind == "linear" else None
        em = ctypes.c_int32(0)
        self._check(self._L.itd_stream_push_host_f64(self._h, _np_ptr(x2), _np_ptr(base), _np_ptr(rot), ctypes.byref(em)))
        if not em.value:
            return None
        if self.kind == "linear":
            return self._shape_out(rot, flat), self._shape_out(base, flat)
        return self._shape_out(base, flat)

    def flush(self, flat=None):
        """The last block's result (it has no successor); None if the stream holds nothing."""
        base = numpy.empty((self.channels, self.block))
        rot = numpy.empty_like(base) if self.kind == "linear" else None
        em = ctypes.c_int32(0)
        self._check(self._L.itd_stream_flush_host_f64(self._h, _np_ptr(base), _np_ptr(rot), ctypes.byref(em)))
        if not em.value:
            return None
        flat = (self.channels == 1) if flat is None else flat
        if self.kind == "linear":
            return self._shape_out(rot, flat), self._shape_out(base, flat)
        return self._shape_out(base, flat)


class BlockwiseCubic(Stream):
    """One channel of the cubic recipe: push(block) -> baseline of the previous block."""

    def __init__(self, block, margin=8, device=0):
        super().__init__(block, 1, "cubic", margin, False, device)


class BlockwiseLinear(Stream):
    """One channel of the tier-1 operator: push(block) -> (rotation, baseline) of the previous block."""

    def __init__(self, block, device=0):
        super().__init__(block, 1, "linear", 1, False, device)


def blockwise(x, block, kind="cubic", margin=8, shared_knots=False, device=0):
    """Run a whole array x[channels, n_blocks * block] (or [n]) through a stream: what a caller with an unbounded source
    does block by block.  Returns baseline (cubic) or (rotation, baseline) (linear)."""
    x = numpy.asarray(x, dtype=numpy.float64)
    flat = x.ndim == 1
    x2 = x.reshape(1, -1) if flat else x
    C, n = x2.shape
    if n % block or n < block:
        raise ValueError("the length must be a multiple of the block")
    st = Stream(block, C, kind, margin, shared_knots, device)
    outs = []
    try:
        for k in range(n // block):
            r = st.push(x2[:, k * block:(k + 1) * block])
            if r is not None:
                outs.append(r)
        outs.append(st.flush(flat=False))
    finally:
        st.close()
    if kind == "linear":
        rot = numpy.concatenate([o[0] for o in outs], axis=1)
        base = numpy.concatenate([o[1] for o in outs], axis=1)
        return (rot[0], base[0]) if flat else (rot, base)
    base = numpy.concatenate(outs, axis=1)
    return base[0] if flat else base
