"""Block-wise (streaming) use of the cubic-spline baseline operator — the recipe of the comment at itd.cpp:31-44:

    "use a circular buffer with modulous tracking to rotate the samples.  re-assess extrema in the entire buffer every
     iteration.  use from the last extrema in the first buffer to the first extrema in the last buffer.  set the first and
     last baseline knots manually to said values.  update the j array.  compute only the baseline[i] array for the inner
     third of the buffer overall.  rotate buffers, rinse and repeat"

There is no reference code for it (the comment is all there is), so this is a reading of the recipe on top of the
external-knot operator `itd_baseline_extract_fast` (itd_fourier_decomposition.py:49-122 = itd.cpp:156-239, compute_extrema =
false), every numeric step of which runs on the GPU (pyitd_amd/csrc/itd_cubic.hpp):

  * three blocks of `block` samples are kept; when a new block arrives the extrema of the whole buffer are re-assessed with
    itd.cpp's own predicate (:161-168);
  * the spline is built on the extrema from `margin` extrema before the inner third to `margin` extrema after it (the
    recipe's literal choice is margin = 1: "the last extrema in the first buffer to the first extrema in the last buffer";
    the operator pins the first and last knot value to the data there, as the recipe asks.  Because the operator never
    computes the knot value of its second-to-last knot (K[idx-1] stays 0, a quirk of the reference), two further extrema are
    always taken behind the inner third so that the quirk's segments lie outside what is emitted — its influence still
    decays only by ~0.27 per knot, which is why the default margin is 8 extrema rather than the recipe's 1);
  * only the inner third is emitted.

`push(block)` returns the baseline of the PREVIOUS block (latency: one block), or None while the buffer is filling;
`flush()` returns the last block's baseline.  Blocks with too few extrema around them return the block unchanged
(itd.cpp:170-172: "break early").
"""
import numpy

from .itd import _engine_for


class BlockwiseCubic:
    def __init__(self, block, margin=8, device=0):
        if block < 8 or margin < 1:
            raise ValueError("block >= 8 samples, margin >= 1 extremum")
        self.block, self.margin, self.device = int(block), int(margin), int(device)
        self._buf = numpy.zeros(3 * self.block)
        self._filled = 0            # blocks pushed so far

    def _extract_inner(self, buf, lo, hi):
        """baseline of buf[lo:hi] from the spline through the extrema around it"""
        eng = _engine_for(len(buf), self.device)
        _, knots, m = eng.cubic_extract_host(buf)               # re-assess the extrema of the entire buffer (GPU)
        knots = knots[:m]
        a = int(numpy.searchsorted(knots, lo, side="left"))     # first extremum inside the emitted part
        b = int(numpy.searchsorted(knots, hi, side="left"))     # first extremum behind it
        first = max(a - self.margin, 0)
        last = min(b + self.margin + 2, m)                      # + 2: keep the K[idx-1] = 0 quirk outside the emitted part
        sel = knots[first:last]
        if len(sel) < 4:
            return buf[lo:hi].copy()                            # too few extrema: the block is its own baseline
        base, _, _ = eng.cubic_extract_host(buf, sel, len(sel) - 1)
        return base[lo:hi]

    def push(self, samples):
        x = numpy.asarray(samples, dtype=numpy.float64)
        if x.shape != (self.block,):
            raise ValueError("expected a block of %d samples" % self.block)
        L = self.block
        self._buf[:2 * L] = self._buf[L:]                       # rotate
        self._buf[2 * L:] = x
        self._filled += 1
        if self._filled < 2:
            return None
        if self._filled == 2:                                    # the first block has no predecessor: the buffer starts with it
            return self._extract_inner(self._buf[L:], 0, L)
        return self._extract_inner(self._buf, L, 2 * L)

    def flush(self):
        """baseline of the last block pushed (it has no successor)"""
        if self._filled == 0:
            return None
        L = self.block
        if self._filled == 1:
            return self._extract_inner(self._buf[2 * L:], 0, L)
        return self._extract_inner(self._buf[L:], L, 2 * L)
