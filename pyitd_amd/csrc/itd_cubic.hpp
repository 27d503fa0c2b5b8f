// itd_cubic.hpp — the cubic-spline baseline variant with externally supplied knots:
//     itd_baseline_extract_fast(I, extrema_input, idx)      itd_fourier_decomposition.py:49-122
//   = itd_baseline_extract(data, baseline, length, &idx, compute_extrema)   itd.cpp:156-239 (float32 fragment; its Python twin
//     above is the form that can be pinned, and the one restated here in float64)
// plus the two knot sources the reference feeds it with: find_extrema (zero crossings of a synthetic sine,
// itd_fourier_decomposition.py:17-31) and itd.cpp's own 3-point predicate (itd.cpp:161-168).
//
// The reference is three serial loops over the knots (knot values, the "Thomas" sweep as written, the back substitution)
// and one over the samples.  Here everything is driven by a per-signal CubicJob in DEVICE memory (which knots, how many,
// valid or not), so no knot count ever crosses to the host in the middle of a call, the launches are batched over signals /
// channels (grid.y; one knot list for all of them or one each: itd.cpp:40-44 "retain the extrema ... along multiple channels")
// and a call can be captured into a graph:
//   k_cubic_sweep<true>   knot values K[k] (:61-83), the right-hand side and pivots (:88-96) computed on the fly, and the
//                         forward sweep  b[i] = (b0[i] - u[i] b[i-1]) / d[i]  (:93-98);
//   k_cubic_sweep<false>  the back substitution  b[i] = b[i] - v[i] b[i+1]  (:100-101), natural ends (:104-105).
//       Both are first-order linear recurrences y -> a_i + c_i y.  A workgroup owns 4096 consecutive elements and walks them in
//       coalesced rounds of 256: the maps of a round are composed in order by wave scans (shuffles), across the four waves through
//       LDS, across rounds by a running carry.  ACROSS workgroups nothing is exchanged: a workgroup starts kWarm = 256 elements
//       early from y = 0.  That is exact to far below rounding because the maps contract — forward |c_i c_{i+1}| <= 1/2 for
//       every pair (c_i = u_i / (2 - u_i v_{i-1}) <= u_i / (2 - u_i), c_{i+1} <= 1 / (1 + u_i)), so 256 elements damp the
//       unknown start value by 2^-128; backward c_i = v_i = h_i / (h_{i-1} + h_i) with integer spacings below 2^31, whose
//       product over 256 elements is below 1e-70.  (The first version ran reduce / carries / apply launches per direction over
//       six intermediate arrays: 198 of its 528 us at 2^24 samples.)
//   k_cubic_eval    one wavefront per 512 samples: segment of each sample from a bitmap of the tile's knots (the j_lookup
//                   loop, :107-111, is "number of knots e[1..idx-1] at or before the sample"), then the spline formula
//                   (:113-120) with t*t*t where the reference has numpy's float64 ** 3 (libm pow; numba multiplies).
// The association of the sweeps differs from the reference's serial loops, so float parity is a tolerance, not bit equality
// (tests: 1e-9 of the signal's scale, measured ~4e-16); knot indices are exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace itd {

// What one signal's spline is built on.  Written on the device: by k_cubic_jobs (a caller's list / the detected knots) or by
// the block-wise stream's knot selection (itd_stream.hpp).
struct CubicJob {
    int32_t first;    // the signal's knots are list[first .. first + idx] (idx + 1 entries, the reference's extrema_input)
    int32_t idx;      // the reference's idx
    int32_t valid;    // 0: nothing is built; k_cubic_eval leaves the baseline untouched (itd.cpp:170-172) or copies the signal
    int32_t status;   // 0 ok, 1 = the caller's list is not strictly increasing / not inside the signal, 2 = NaN in the signal
};

struct CubicArgs {
    const double *x; int64_t x_stride;        // signals
    int64_t n;
    const int32_t *e; int64_t e_stride;       // knot lists (e_stride = 0: one list for every signal)
    const CubicJob *jobs; int job_stride;     // 0: one job for every signal (shared knots), 1: one each
    double *K, *bf, *b; int64_t a_stride;     // per-signal knot arrays: values, forward sweep, final second derivatives
};

constexpr int kScanThreads = 256, kScanRounds = 16, kScanBlockElems = kScanRounds * kScanThreads, kWarm = kScanThreads;

// ---- jobs -------------------------------------------------------------------------------------------------------------
// mode 0: the caller's list (idx given; validated by k_cubic_validate afterwards); mode 1: the detected knots of signal b,
// totals[2b] of them behind a leading slot (k_compact), totals[2b+1] = the signal holds a NaN
__global__ void k_cubic_jobs(CubicJob *__restrict__ jobs, int n_jobs, int mode, int64_t idx, const int32_t *__restrict__ totals)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_jobs) return;
    CubicJob j;
    if (mode == 0) { j.first = 0; j.idx = (int32_t)idx; j.valid = idx >= 2; j.status = 0; }
    else {
        j.first = 1;
        j.idx = totals[2 * b];
        j.status = totals[2 * b + 1] ? 2 : 0;
        j.valid = j.idx >= 2 && j.status == 0;      // itd.cpp:170-172: fewer than 2 knots, break early
    }
    jobs[b] = j;
}

// knots handed in by the caller: e[0..idx-1] strictly increasing and inside the signal, e[idx] inside the signal
__global__ void k_cubic_validate(const int32_t *__restrict__ e, int64_t e_stride, int64_t idx, int64_t n, CubicJob *__restrict__ jobs)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > idx) return;
    const int32_t *el = e + (int64_t)blockIdx.y * e_stride;
    const int64_t v = el[k];
    bool ok = v >= 0 && v < n;
    if (k >= 1 && k < idx) ok = ok && v > (int64_t)el[k - 1];
    if (!ok) { jobs[blockIdx.y].valid = 0; jobs[blockIdx.y].status = 1; }
}

// ---- first-order linear recurrences  y[i] = f_i(y[i-1])  with affine f_i -----------------------------------------------
struct Affine { double a, c; };   // y -> a + c * y
__device__ __forceinline__ Affine compose(Affine later, Affine earlier)   // later o earlier
{
    return Affine{later.a + later.c * earlier.a, later.c * earlier.c};
}
__device__ __forceinline__ Affine wave_inclusive(Affine inc, int lane)
{
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const double pa = __shfl_up(inc.a, dlt), pc = __shfl_up(inc.c, dlt);
        if (lane >= dlt) inc = compose(inc, Affine{pa, pc});
    }
    return inc;
}

// knot value K[k] (:61-83): the end values are the data, K[idx-1] = 0 (the loop at :61 stops at idx-2)
__device__ __forceinline__ double cubic_knot_value(const double *__restrict__ I, const int32_t *__restrict__ e, int k, int idx)
{
    if (k == 0 || k == idx) return I[e[k]];                  // :83
    if (k == idx - 1) return 0.0;
    const int32_t ep = e[k - 1], ec = e[k], en = e[k + 1];
    const double ap = I[ep], ac = I[ec], an = I[en];
    const double weight = (double)(ec - ep) / (double)(en - ep);                 // :77
    return 0.5 * (ap + weight * (an - ap)) + (1 - 0.5) * ac;                      // :80
}

// FWD: element q (q = 0 .. idx-2) is knot i = 1 + q:  y = (b0[i] - u[i] * y_prev) / d[i], y in front of knot 1 = b[0] = 0
// BWD: element q is knot i = idx-2-q (running downwards):  y = bf[i] - v[i] * y_prev, v[i] = 1 - u[i]; y in front = bf[idx-1]
template <bool FWD>
__global__ __launch_bounds__(kScanThreads) void k_cubic_sweep(CubicArgs A)
{
    __shared__ Affine s_wave[kScanThreads / 64];
    __shared__ double s_y;
    __shared__ double s_K[kScanThreads + 2];
    const int sig = blockIdx.y;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    if (!job.valid) return;
    const int idx = job.idx, count = idx - 1;
    const int base = blockIdx.x * kScanBlockElems;
    if (base >= count) return;
    const int32_t *e = A.e + (int64_t)sig * A.e_stride + job.first;
    const double *I = A.x + (int64_t)sig * A.x_stride;
    double *K = A.K + (int64_t)sig * A.a_stride, *bf = A.bf + (int64_t)sig * A.a_stride, *bo = A.b + (int64_t)sig * A.a_stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r_first = base >= kWarm ? -1 : 0;        // one warm-up round in front of the workgroup's own elements
    double y_in = 0.0;                                  // FWD at the very front: b[0] = 0
    if (!FWD && r_first == 0) y_in = bf[idx - 1];       // BWD at the very front: the forward sweep's last value
    for (int round = r_first; round < kScanRounds; ++round) {
        const int q0 = base + round * kScanThreads;     // uniform
        if (q0 >= count) break;
        const int q = q0 + tid;
        Affine m{0.0, 1.0};
        if (FWD) {
            // knot values of the round's knots 1+q0 .. 1+q0+255 and the two next to them, through LDS
            const int i = 1 + q;
            if (i <= idx) s_K[tid + 1] = cubic_knot_value(I, e, i, idx);
            if (tid == 0) s_K[0] = cubic_knot_value(I, e, q0, idx);
            if (tid == 1 && q0 + kScanThreads + 1 <= idx) s_K[kScanThreads + 1] = cubic_knot_value(I, e, q0 + kScanThreads + 1, idx);
            __syncthreads();
            if (q < count) {
                const double Km = s_K[tid], Kc = s_K[tid + 1], Kp = s_K[tid + 2];
                const double hm = (double)(e[i] - e[i - 1]), hi = (double)(e[i + 1] - e[i]);           // :86
                const double ui = hm / (hm + hi);                                                       // :89
                double vm = 0.0;                                                                        // v[0] = 0
                if (i >= 2) { const double hmm = (double)(e[i - 1] - e[i - 2]); vm = 1 - hmm / (hmm + hm); }   // :90
                const double b0 = 6 * ((Kp - Kc) / hi - (Kc - Km) / hm) / (hm + hi);                    // :91
                const double di = 2 - ui * vm;                                                          // :94,96 (original u and v)
                m = Affine{b0 / di, -(ui / di)};
                if (round >= 0) {
                    K[i] = Kc;
                    if (q == 0) K[0] = Km;
                    if (q == count - 1) K[idx] = Kp;
                }
            }
        } else if (q < count) {
            const int i = idx - 2 - q;
            m = Affine{0.0, 0.0};                                                   // i = 0: v[0] = 0 and b[0] is forced to 0 (:104)
            if (i >= 1) {
                const double hm = (double)(e[i] - e[i - 1]), hi = (double)(e[i + 1] - e[i]);
                m = Affine{bf[i], -(1 - hm / (hm + hi))};                           // v[i] = 1 - u[i], :90, :101
            }
        }
        const Affine inc = wave_inclusive(m, lane);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        Affine pre{0.0, 1.0};             // the waves of this round in front of mine
        for (int w = 0; w < wave; ++w) pre = compose(s_wave[w], pre);
        const double y_wave = pre.a + pre.c * y_in;          // y in front of my wave
        const double y = inc.a + inc.c * y_wave;
        if (q < count && round >= 0) {
            if (FWD) bf[1 + q] = y;
            else { const int i = idx - 2 - q; bo[i] = i == 0 ? 0.0 : y; }           // :104 b[0] = 0
        }
        if (tid == kScanThreads - 1) s_y = y;                // the round's last element (identity maps carry it on)
        __syncthreads();
        y_in = s_y;
    }
    if (!FWD && base == 0 && tid == 0) { bo[idx - 1] = 0.0; bo[idx] = 0.0; }   // :105; b[idx] is the reference's untouched zero
}

// ---- evaluation: one wavefront per TW samples of [lo, hi) ------------------------------------------------------------
// j(i) = number of k in [1, idx-1] with e[k] <= i (the j_lookup loop :107-111 for increasing knots), t = (i - e[j]) / h[j].
// out holds the samples lo .. hi-1 of every signal (out[sig * out_stride + i - lo]); a signal without a valid job is left
// untouched (itd.cpp:170-172) or, with copy_invalid, copied (the block-wise recipe: "the block is its own baseline").
// NAK = the interpolating not-a-knot spline of itd_nak.hpp on the same representation (values K, second derivatives b at the
// sites e[0..idx]): every segment is cubic — the reference operator's linear last segment (:117) is the natural variant's alone.
template <int TW, bool NAK = false>
__global__ __launch_bounds__(64) void k_cubic_eval(CubicArgs A, int64_t lo, int64_t hi, double *__restrict__ out, int64_t out_stride,
                                                   int copy_invalid, double *__restrict__ rot = nullptr, int64_t rot_stride = 0)
{
    static_assert(TW == 512, "8 bitmap words of 64 positions");
    __shared__ unsigned long long s_bits[TW / 64];
    const int lane = threadIdx.x;
    const int sig = blockIdx.y;
    const int64_t s = lo + (int64_t)blockIdx.x * TW;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    double *o = out + (int64_t)sig * out_stride - lo;
    double *ro = rot ? rot + (int64_t)sig * rot_stride - lo : nullptr;     // optional: signal - baseline
    const double *xs = A.x + (int64_t)sig * A.x_stride;
    if (!job.valid) {
        if (copy_invalid)
            for (int p = lane; p < TW; p += 64)
                if (s + p < hi) {
                    o[s + p] = xs[s + p];
                    if (ro) ro[s + p] = 0.0;
                }
        return;
    }
    const int64_t idx = job.idx;
    const int32_t *e = A.e + (int64_t)sig * A.e_stride + job.first;
    const double *K = A.K + (int64_t)sig * A.a_stride, *b = A.b + (int64_t)sig * A.a_stride;
    // j0 = number of knots e[1..idx-1] strictly in front of the tile (<= s-1): lower bound of s, wave-uniform
    int64_t l = 1, h = idx;             // first k in [1, idx) with e[k] >= s
    while (l < h) {
        const int64_t mid = (l + h) >> 1;
        if (e[mid] < s) l = mid + 1; else h = mid;
    }
    const int64_t kfirst = l;           // knots kfirst, kfirst+1, ... lie at or behind s
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
        if (i >= hi) continue;
        const unsigned long long upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int64_t j = pre[q] + __popcll(w[q] & upto);        // knots at or before the sample
        const int32_t ej = e[j];
        const double hj = (double)(e[j + 1] - ej);                                  // :86
        const double t = (double)(i - (int64_t)ej) / hj;                            // :115
        const double Kj = K[j], Kn = K[j + 1];
        double v;
        if (!NAK && j == idx - 2) {
            v = (1 - t) * Kj + t * Kn;                                              // :117
        } else {
            const double omt = 1 - t;
            const double c1 = hj * hj / 6 * ((omt * omt * omt - 1) + t) * b[j];     // :119 ((1-t)**3 - 1 + t)
            const double c2 = hj * hj / 6 * (t * t * t - t) * b[j + 1];
            v = ((omt * Kj + t * Kn) + c1) + c2;
        }
        o[i] = v;
        if (ro) ro[i] = xs[i] - v;
    }
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
