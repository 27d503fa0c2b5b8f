// itd_stream.hpp — device side of the block-wise (streaming) operators, itd_stream_* in include/pyitd_hip.h.
//
// The reference describes block-wise operation in a comment only (itd.cpp:31-38: "use a circular buffer with modulous
// tracking ... re-assess extrema in the entire buffer every iteration ... use from the last extrema in the first buffer to the
// first extrema in the last buffer ... compute only the baseline[i] array for the inner third ... rotate buffers") and the
// reuse of retained extrema along channels (itd.cpp:40-44).  The recipe as built here: include/pyitd_hip.h, DESIGN.md section 7.
//
// The ring: per channel 5 slots of `block` samples.  Block k lives in slot k % 3 and — slots 0 and 1 — a second time in slot
// 3 + k % 3, so the three most recent blocks are always one CONTIGUOUS window (first slot (k - 2) % 3) and every kernel of the
// whole-signal operators runs on it unchanged: modulus tracking without a modulus in the kernels' addressing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "itd_cubic.hpp"

namespace itd {

// one incoming block of every channel into its slot (and the slot's mirror)
__global__ void k_stream_store(const double *__restrict__ blk, int64_t in_stride, double *__restrict__ ring, int64_t ring_stride,
                               int64_t L, int slot)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int c = blockIdx.y;
    const double v = blk[(int64_t)c * in_stride + i];
    double *r = ring + (int64_t)c * ring_stride;
    r[(int64_t)slot * L + i] = v;
    if (slot < 2) r[(int64_t)(slot + 3) * L + i] = v;
}

// The knots a window's spline is built on: from `margin` extrema in front of the
// emitted part [lo, hi) to margin + 2 behind it; fewer than 4: no spline (the block is emitted unchanged, itd.cpp:170-172).
// kidx[b]: [0, the window's knots (totals[2b] of them), tail] as k_compact leaves them.  One thread per list.
__global__ void k_stream_select(const int32_t *__restrict__ kidx, int64_t kidx_stride, const int32_t *__restrict__ totals, int n_lists,
                                int lo, int hi, int margin, CubicJob *__restrict__ jobs, int32_t *__restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_lists) return;
    const int32_t *k = kidx + (int64_t)b * kidx_stride + 1;
    const int m = totals[2 * b];
    auto lower = [&](int v) {            // first position with k[pos] >= v (numpy.searchsorted, side = "left")
        int l = 0, h = m;
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (k[mid] < v) l = mid + 1; else h = mid;
        }
        return l;
    };
    const int a = lower(lo), bb = lower(hi);
    const int first = max(a - margin, 0), last = min(bb + margin + 2, m);
    const int cnt = last - first;
    CubicJob j;
    j.first = 1 + first;
    j.idx = cnt - 1;
    j.status = totals[2 * b + 1] ? 2 : 0;
    j.valid = cnt >= 4 && j.status == 0;
    jobs[b] = j;
    if (j.status) atomicOr(status, j.status);
}

// the emitted part of the tier-1 operator's window results
__global__ void k_stream_emit2(const double *__restrict__ rot_w, const double *__restrict__ base_w, int64_t w_stride, int64_t lo,
                               int64_t L, double *__restrict__ rot, int64_t rot_stride, double *__restrict__ base, int64_t base_stride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int c = blockIdx.y;
    if (base) base[(int64_t)c * base_stride + i] = base_w[(int64_t)c * w_stride + lo + i];
    if (rot) rot[(int64_t)c * rot_stride + i] = rot_w[(int64_t)c * w_stride + lo + i];
}

}  // namespace itd
