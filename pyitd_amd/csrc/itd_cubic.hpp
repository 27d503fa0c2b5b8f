// itd_cubic.hpp — the cubic-spline baseline variant with externally supplied knots:
//     itd_baseline_extract_fast(I, extrema_input, idx)      itd_fourier_decomposition.py:49-122
//   = itd_baseline_extract(data, baseline, length, &idx, compute_extrema)   itd.cpp:156-239 (float32 fragment; its Python twin
//     above is the form that can be pinned, and the one restated here in float64)
// plus the two knot sources the reference feeds it with: find_extrema (zero crossings of a synthetic sine,
// itd_fourier_decomposition.py:17-31) and itd.cpp's own 3-point predicate (itd.cpp:161-168).
//
// The reference is three serial loops over the knots (knot values, the "Thomas" sweep as written, the back substitution)
// and one over the samples.  Here:
//   k_cubic_knots   knot values K[k] (:61-83) and spacings h[k] (:85-86), one thread per knot;
//   k_cubic_rhs     u, the right-hand side b0 and the pivots d[i] = 2 - u[i] * v[i-1] (:88-96; d does not depend on the sweep);
//   affine scans    the forward sweep  b[i] = (b0[i] - u[i] b[i-1]) / d[i]  (:93-98) and the back substitution
//                   b[i] = b[i] - v[i] b[i+1]  (:100-101) are first-order linear recurrences y -> a_i + c_i y: the maps are
//                   composed in order by wave scans inside a workgroup and across workgroups (three launches: reduce, scan of
//                   the workgroup aggregates, apply); every element is its inclusive map applied to the carry-in, so the
//                   association differs from the reference's serial loop (|c_i| < 1: the composed maps contract, rounding
//                   differences do not grow);
//   k_cubic_eval    one wavefront per 512 samples: segment of each sample from a bitmap of the tile's knots (the j_lookup
//                   loop, :107-111, is "number of knots e[1..idx-1] at or before the sample"), then the spline formula
//                   (:113-120) with t*t*t where the reference has numpy's float64 ** 3 (libm pow; numba multiplies).
// Float parity is therefore a tolerance, not bit equality (tests: 1e-9 of the signal's scale); knot indices are exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace itd {

// ---- knot values and spacings ----------------------------------------------------------------------------------------
// e: idx+1 knots (int32, device); K: idx+1 values; h: idx spacings.  K[idx-1] = 0 (never computed by the reference).
__global__ void k_cubic_knots(const double *__restrict__ I, const int32_t *__restrict__ e, int64_t idx, double *__restrict__ K,
                              double *__restrict__ h)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > idx) return;
    double v;
    if (k == 0 || k == idx) v = I[e[k]];                      // :83
    else if (k == idx - 1) v = 0.0;                           // the loop at :61 stops at idx-2
    else {
        const int32_t ep = e[k - 1], ec = e[k], en = e[k + 1];
        const double ap = I[ep], ac = I[ec], an = I[en];
        const double weight = (double)(ec - ep) / (double)(en - ep);                 // :77
        v = 0.5 * (ap + weight * (an - ap)) + (1 - 0.5) * ac;                          // :80
    }
    K[k] = v;
    if (k < idx) h[k] = (double)(e[k + 1] - e[k]);                                     // :86
}

// ---- u, b0, d for i = 1 .. idx-1 (:88-96); entry 0 of each array is 0 like the reference's zeros ----------------------
__global__ void k_cubic_rhs(const double *__restrict__ K, const double *__restrict__ h, int64_t idx, double *__restrict__ u,
                            double *__restrict__ b0, double *__restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= idx) return;
    if (i == 0) { u[0] = 0.0; b0[0] = 0.0; d[0] = 0.0; return; }
    const double hm = h[i - 1], hi = h[i];
    const double ui = hm / (hm + hi);                                                   // :89
    double vm = 0.0;                                                                    // v[0] = 0
    if (i >= 2) { const double hmm = h[i - 2]; vm = 1 - hmm / (hmm + hm); }            // v[i-1] = 1 - u[i-1], :90
    u[i] = ui;
    b0[i] = 6 * ((K[i + 1] - K[i]) / hi - (K[i] - K[i - 1]) / hm) / (hm + hi);          // :91
    d[i] = 2 - ui * vm;                                                                 // :94,96 (original u and v)
}

// ---- first-order linear recurrences  y[i] = f_i(y[i-1])  with affine f_i -----------------------------------------------
struct Affine { double a, c; };   // y -> a + c * y
__device__ __forceinline__ Affine compose(Affine later, Affine earlier)   // later o earlier
{
    return Affine{later.a + later.c * earlier.a, later.c * earlier.c};
}
constexpr int kScanChunk = 16, kScanThreads = 256, kScanBlockElems = kScanChunk * kScanThreads;

// FWD: element q (q = 0 .. count-1) is knot i = 1 + q:  y = (b0[i] - u[i] * y_prev) / d[i]
// BWD: element q is knot i = idx-2-q (running downwards):  y = bf[i] - v[i] * y_prev,  v[i] = 1 - u[i]
template <bool FWD>
struct Recur {
    const double *b0_or_bf, *u, *d;
    int64_t idx;
    __device__ __forceinline__ int64_t knot(int64_t q) const { return FWD ? 1 + q : idx - 2 - q; }
    __device__ __forceinline__ Affine map(int64_t q) const
    {
        const int64_t i = knot(q);
        if (FWD) { const double di = d[i]; return Affine{b0_or_bf[i] / di, -(u[i] / di)}; }
        const double vi = (i == 0) ? 0.0 : 1 - u[i];
        return Affine{b0_or_bf[i], -vi};
    }
};

// The scans run over consecutive elements in COALESCED rounds: a workgroup owns kScanBlockElems consecutive elements and walks
// them in rounds of kScanThreads (thread t takes element round * kScanThreads + t: lane-contiguous loads and stores; the first
// version gave every thread 16 consecutive elements, i.e. a 128-byte stride between lanes — every load instruction touched 64
// cache lines and the launches ran at a twentieth of the memory rate).  Inside a round the maps are combined by an ordered
// wave scan (shuffles), the four waves through LDS, the rounds through a running carry.
__device__ __forceinline__ Affine wave_inclusive(Affine inc, int lane)
{
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const double pa = __shfl_up(inc.a, dlt), pc = __shfl_up(inc.c, dlt);
        if (lane >= dlt) inc = compose(inc, Affine{pa, pc});
    }
    return inc;
}

// reduce: the composed map of the workgroup's elements, in order
template <bool FWD>
__global__ __launch_bounds__(kScanThreads) void k_recur_reduce(Recur<FWD> r, int64_t count, Affine *__restrict__ block_maps)
{
    __shared__ Affine s_wave[kScanThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanBlockElems;
    Affine tot{0.0, 1.0};   // thread 0's copy is the one that counts
    for (int round = 0; round < kScanChunk; ++round) {
        const int64_t q = base + (int64_t)round * kScanThreads + threadIdx.x;
        const Affine m = q < count ? r.map(q) : Affine{0.0, 1.0};
        const Affine inc = wave_inclusive(m, lane);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int w = 0; w < kScanThreads / 64; ++w) tot = compose(s_wave[w], tot);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) block_maps[blockIdx.x] = tot;
}

// carry-in of every workgroup: y in front of its first element.  One wavefront: lane l composes its run of consecutive
// workgroup maps, the runs are scanned across the lanes, and every lane then walks its run from its own carry-in.
__global__ __launch_bounds__(64) void k_recur_carries(const Affine *__restrict__ block_maps, int n_blocks,
                                                      const double *__restrict__ y_init /* nullptr: 0 */, double *__restrict__ carry)
{
    const int lane = threadIdx.x;
    const int per = (n_blocks + 63) / 64;
    const int b0 = lane * per, b1 = min(n_blocks, b0 + per);
    Affine m{0.0, 1.0};
    for (int b = b0; b < b1; ++b) m = compose(block_maps[b], m);
    const Affine inc = wave_inclusive(m, lane);
    const double ea = __shfl_up(inc.a, 1), ec = __shfl_up(inc.c, 1);
    const Affine ex = lane == 0 ? Affine{0.0, 1.0} : Affine{ea, ec};
    double y = ex.a + ex.c * (y_init ? *y_init : 0.0);
    for (int b = b0; b < b1; ++b) {
        carry[b] = y;
        y = block_maps[b].a + block_maps[b].c * y;
    }
}

// apply: every element from the running carry through the inclusive maps of its round
template <bool FWD>
__global__ __launch_bounds__(kScanThreads) void k_recur_apply(Recur<FWD> r, int64_t count, const double *__restrict__ carry,
                                                               double *__restrict__ out)
{
    __shared__ Affine s_wave[kScanThreads / 64];
    __shared__ double s_y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanBlockElems;
    double y_in = carry[blockIdx.x];      // y in front of the round's first element
    for (int round = 0; round < kScanChunk; ++round) {
        const int64_t q = base + (int64_t)round * kScanThreads + threadIdx.x;
        const Affine m = q < count ? r.map(q) : Affine{0.0, 1.0};
        const Affine inc = wave_inclusive(m, lane);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        Affine pre{0.0, 1.0};             // the waves of this round in front of mine
        for (int w = 0; w < wave; ++w) pre = compose(s_wave[w], pre);
        const double y_wave = pre.a + pre.c * y_in;          // y in front of my wave
        const double y = inc.a + inc.c * y_wave;
        if (q < count) out[r.knot(q)] = y;
        if (threadIdx.x == kScanThreads - 1) s_y = y;       // the round's last element (identity maps carry it on)
        __syncthreads();
        y_in = s_y;
    }
}

// b[0] = 0, b[idx-1] = 0 (:104-105), b[idx] = 0 (never written by the reference: zeros)
__global__ void k_cubic_fix_ends(double *__restrict__ b, int64_t idx)
{
    b[0] = 0.0;
    b[idx - 1] = 0.0;
    b[idx] = 0.0;
}

// ---- evaluation: one wavefront per TW samples ------------------------------------------------------------------------
// j(i) = number of k in [1, idx-1] with e[k] <= i (the j_lookup loop :107-111 for increasing knots), t = (i - e[j]) / h[j].
template <int TW>
__global__ __launch_bounds__(64) void k_cubic_eval(const int32_t *__restrict__ e, const double *__restrict__ K,
                                                   const double *__restrict__ b, const double *__restrict__ h, int64_t idx,
                                                   int64_t n, double *__restrict__ baseline)
{
    static_assert(TW == 512, "8 bitmap words of 64 positions");
    __shared__ unsigned long long s_bits[TW / 64];
    const int lane = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * TW;
    // j0 = number of knots e[1..idx-1] strictly in front of the tile (<= s-1): lower bound of s, wave-uniform
    int64_t lo = 1, hi = idx;           // first k in [1, idx) with e[k] >= s
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (e[mid] < s) lo = mid + 1; else hi = mid;
    }
    const int64_t kfirst = lo;          // knots kfirst, kfirst+1, ... lie at or behind s
    if (lane < TW / 64) s_bits[lane] = 0ull;
    __syncthreads();
    for (int64_t k = kfirst + lane; k < idx; k += 64) {     // at most TW of them fall into the tile
        const int64_t p = (int64_t)e[k] - s;
        const bool in = p < TW;
        if (in) atomicOr(&s_bits[p >> 6], 1ull << (p & 63));
        if (!__any(in)) break;
    }
    __syncthreads();
    unsigned long long w[TW / 64];
    int pre[TW / 64];
    int acc = (int)(kfirst - 1);        // knots in front of the tile
#pragma unroll
    for (int q = 0; q < TW / 64; ++q) {
        w[q] = s_bits[q];
        pre[q] = acc;
        acc += __popcll(w[q]);
    }
#pragma unroll
    for (int q = 0; q < TW / 64; ++q) {
        const int p = q * 64 + lane;
        const int64_t i = s + p;
        if (i >= n) continue;
        const unsigned long long upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int64_t j = pre[q] + __popcll(w[q] & upto);        // knots at or before the sample
        const double hj = h[j];
        const double t = (double)(i - (int64_t)e[j]) / hj;                          // :115
        const double Kj = K[j], Kn = K[j + 1];
        double v;
        if (j == idx - 2) {
            v = (1 - t) * Kj + t * Kn;                                              // :117
        } else {
            const double omt = 1 - t;
            const double c1 = hj * hj / 6 * ((omt * omt * omt - 1) + t) * b[j];     // :119 ((1-t)**3 - 1 + t)
            const double c2 = hj * hj / 6 * (t * t * t - t) * b[j + 1];
            v = ((omt * Kj + t * Kn) + c1) + c2;
        }
        baseline[i] = v;
    }
}

// knots handed in by the caller: e[0..idx-1] strictly increasing and inside the signal, e[idx] inside the signal
__global__ void k_cubic_validate(const int32_t *__restrict__ e, int64_t idx, int64_t n, int32_t *__restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > idx) return;
    const int64_t v = e[k];
    bool ok = v >= 0 && v < n;
    if (k >= 1 && k < idx) ok = ok && v > (int64_t)e[k - 1];
    if (!ok) *bad = 1;
}

// find_extrema's tail (itd_fourier_decomposition.py:29-30): e[m+1] = 2 e[m] - e[m-1]; python's e[-1] (= 0) when m = 0
__global__ void k_zero_cross_tail(int32_t *__restrict__ e, const int32_t *__restrict__ total)
{
    const int m = total[0];
    const int a = e[m], bq = m >= 1 ? e[m - 1] : 0;
    e[m + 1] = 2 * a - bq;
}

__global__ void k_narrow_idx(const int64_t *__restrict__ src, int32_t *__restrict__ dst, int64_t cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) dst[i] = (int32_t)src[i];
}

__global__ void k_widen_f32(const float *__restrict__ src, double *__restrict__ dst, int64_t cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) dst[i] = (double)src[i];
}

}  // namespace itd
