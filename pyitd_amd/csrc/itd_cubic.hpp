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
//       Both are first-order linear recurrences y -> a_i + c_i y.  A workgroup owns 1792 consecutive elements and walks them in
//       rounds of 1024 (four per thread): the maps of a round are composed in order by wave scans (shuffles), across the four waves through
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
#include <type_traits>

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
    const int32_t *tbase = nullptr; int64_t tb_stride = 0;   // optional: detected knots in front of every 512-sample tile (k_compact), per list
};

#ifndef ITD_SCAN_ROUNDS
#define ITD_SCAN_ROUNDS 2      // rounds of 1024 elements a sweep workgroup walks, its 256 warm-up elements included (A/B builds)
#endif
constexpr int kScanThreads = 256, kScanPer = 4, kScanRoundElems = kScanThreads * kScanPer, kScanRounds = ITD_SCAN_ROUNDS, kWarm = 256;
constexpr int kScanBlockElems = kScanRounds * kScanRoundElems - kWarm;   // elements a workgroup owns
constexpr int kScanStage = kScanRounds * kScanRoundElems + 4;            // knots a forward workgroup stages: its span and two either side

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

// 1 / x for the sweeps' divisors (knot spacings: integers below 2^31; pivots 2 - u v in [1.5, 2]): hardware estimate + two Newton
// steps, ~1 ulp.  The sweeps were bound by IEEE divisions (8 per knot, ~40 instructions each); their results are held to a
// tolerance anyway (the recurrences run as scans), so the reciprocal form costs no parity.
__device__ __forceinline__ double sweep_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
}

// FWD: element q (q = 0 .. idx-2) is knot i = 1 + q:  y = (b0[i] - u[i] * y_prev) / d[i], y in front of knot 1 = b[0] = 0
// BWD: element q is knot i = idx-2-q (running downwards):  y = bf[i] - v[i] * y_prev, v[i] = 1 - u[i]; y in front = bf[idx-1]
// A thread owns kScanPer = 4 consecutive elements of a round (composed serially: the wave scan runs once per 256 elements), and
// everything a workgroup's rounds read is requested BEFORE the first scan: FWD stages the knots' positions and the signal's values
// there for its whole span in LDS (one coalesced load + one gather per knot, all in flight together) and turns them into knot
// values in place; BWD computes its rounds' maps into registers first.  (One element per thread with the loads inside the rounds:
// 100 + 52 us at 6.7 M knots — 17 dependent round trips per workgroup, then bound by the scans' and divisions' instruction count.)
template <bool FWD>
__global__ __launch_bounds__(kScanThreads) void k_cubic_sweep(CubicArgs A)
{
    constexpr int NR = kScanRounds;
    constexpr int NI = (kScanStage + kScanThreads - 1) / kScanThreads;
    __shared__ Affine s_wave[kScanThreads / 64];
    __shared__ double s_y;
    __shared__ int32_t s_E[FWD ? kScanStage : 1];
    __shared__ double s_V[FWD ? kScanStage : 1];                     // the signal at the staged knots; then the knot values K
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
    const int q_first = base >= kWarm ? base - kWarm : base;       // the warm-up elements in front of the workgroup's own
    const int q_end = min(base + kScanBlockElems, count);          // one past the last element walked
    double y_in = 0.0;                                              // FWD at the very front: b[0] = 0
    if (!FWD && q_first == 0) y_in = bf[idx - 1];                   // BWD at the very front: the forward sweep's last value
    Affine mb[FWD ? 1 : NR * kScanPer];
    const int k_start = q_first - 1;                                // FWD: staged knot j is knot k_start + j (two either side of the span)
    if (FWD) {
        int32_t ev[NI];
        double vv[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int j = tid + it * kScanThreads, kk = k_start + j;
            ev[it] = (j < kScanStage && kk >= 0 && kk <= idx) ? e[kk] : -1;
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) vv[it] = ev[it] >= 0 ? I[ev[it]] : 0.0;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int j = tid + it * kScanThreads;
            if (j < kScanStage) { s_E[j] = ev[it]; s_V[j] = vv[it]; }
        }
        __syncthreads();
        // knot values K[k] (:61-83) of the staged knots 1 .. kScanStage-2 (what the span's maps read)
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int j = tid + it * kScanThreads, kk = k_start + j;
            double kv = 0.0;
            if (j >= 1 && j <= kScanStage - 2 && kk >= 0 && kk <= idx) {
                if (kk == 0 || kk == idx) kv = s_V[j];                           // :83 the end values are the data
                else if (kk == idx - 1) kv = 0.0;                                // the loop at :61 stops at idx-2
                else {
                    const int32_t ep = s_E[j - 1], ec = s_E[j], en = s_E[j + 1];
                    const double ap = s_V[j - 1], ac = s_V[j], an = s_V[j + 1];
                    const double weight = (double)(ec - ep) * sweep_rcp((double)(en - ep));   // :77
                    kv = 0.5 * (ap + weight * (an - ap)) + (1 - 0.5) * ac;        // :80
                }
            }
            vv[it] = kv;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int j = tid + it * kScanThreads;
            if (j < kScanStage) s_V[j] = vv[it];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
#pragma unroll
            for (int c = 0; c < kScanPer; ++c) {
                const int q = q_first + r * kScanRoundElems + tid * kScanPer + c;
                Affine m{0.0, 1.0};
                if (q < q_end) {
                    const int i = idx - 2 - q;
                    m = Affine{0.0, 0.0};                                         // i = 0: v[0] = 0 and b[0] is forced to 0 (:104)
                    if (i >= 1) {
                        const double hm = (double)(e[i] - e[i - 1]), hi = (double)(e[i + 1] - e[i]);
                        m = Affine{bf[i], -(1 - hm * sweep_rcp(hm + hi))};        // v[i] = 1 - u[i], :90, :101
                    }
                }
                mb[r * kScanPer + c] = m;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int q0 = q_first + r * kScanRoundElems;    // uniform
        if (q0 >= q_end) break;
        const int qt = q0 + tid * kScanPer;               // the thread's first element of this round
        Affine loc[kScanPer];                             // loc[c] = m_c o ... o m_0 (the thread's elements composed in order)
#pragma unroll
        for (int c = 0; c < kScanPer; ++c) {
            const int q = qt + c;
            Affine m{0.0, 1.0};
            if (FWD) {
                if (q < q_end) {
                    const int i = 1 + q, j = q - q_first + 2;
                    const double Km = s_V[j - 1], Kc = s_V[j], Kp = s_V[j + 1];
                    const double hm = (double)(s_E[j] - s_E[j - 1]), hi = (double)(s_E[j + 1] - s_E[j]);   // :86
                    const double rhm = sweep_rcp(hm), rhi = sweep_rcp(hi), rs = sweep_rcp(hm + hi);
                    const double ui = hm * rs;                                                              // :89
                    double vm = 0.0;                                                                        // v[0] = 0
                    if (i >= 2) { const double hmm = (double)(s_E[j - 1] - s_E[j - 2]); vm = 1 - hmm * sweep_rcp(hmm + hm); }   // :90
                    const double b0 = 6 * ((Kp - Kc) * rhi - (Kc - Km) * rhm) * rs;                         // :91
                    const double rdi = sweep_rcp(2 - ui * vm);                                              // :94,96 (original u and v)
                    m = Affine{b0 * rdi, -(ui * rdi)};
                    if (q >= base) {
                        K[i] = Kc;
                        if (q == 0) K[0] = Km;
                        if (q == count - 1) K[idx] = Kp;
                    }
                }
            } else {
                m = mb[r * kScanPer + c];
            }
            loc[c] = c == 0 ? m : compose(m, loc[c > 0 ? c - 1 : 0]);
        }
        const Affine inc = wave_inclusive(loc[kScanPer - 1], lane);
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        Affine pre{0.0, 1.0};             // the waves of this round in front of mine
        for (int w = 0; w < wave; ++w) pre = compose(s_wave[w], pre);
        const double y_wave = pre.a + pre.c * y_in;          // y in front of my wave
        const double exa = __shfl_up(inc.a, 1), exc = __shfl_up(inc.c, 1);
        const double y_thr = lane == 0 ? y_wave : exa + exc * y_wave;   // y in front of my elements
        double yv[kScanPer];
#pragma unroll
        for (int c = 0; c < kScanPer; ++c) yv[c] = loc[c].a + loc[c].c * y_thr;
#pragma unroll
        for (int c = 0; c < kScanPer; ++c) {
            const int q = qt + c;
            if (q < q_end && q >= base) {
                if (FWD) bf[1 + q] = yv[c];
                else { const int i = idx - 2 - q; bo[i] = i == 0 ? 0.0 : yv[c]; }   // :104 b[0] = 0
            }
        }
        if (tid == kScanThreads - 1) s_y = yv[kScanPer - 1];  // the round's last element (identity maps carry it on)
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
    // Slots for 320 knots (tiles of ordinary signals hold ~200 at most: 0.4 knots per sample; 6.5 KB of LDS keeps 24 wavefronts
    // per CU in flight); a denser tile reads its knots from global memory (the same values).
    constexpr int SL = 320 + 2;
    __shared__ int32_t s_e[SL];
    __shared__ double s_K[SL], s_b[SL];
    if (lane < TW / 64) s_bits[lane] = 0ull;
    __syncthreads();
    bool dense = false, staged = false;
    int64_t kfirst = 1;                 // first k in [1, idx) with e[k] >= s: knots kfirst, kfirst+1, ... lie at or behind s
    const int64_t n_wt = (A.n + TW - 1) / TW;
    const bool have_tb = A.tbase && (s % TW) == 0;
    if (have_tb) {
        // detected knots: the compaction left the knots in front of every tile, so the tile's part of the list is known at once and
        // everything is requested in ONE round trip (a wavefront used to walk the list 64 knots at a time, each step waiting for
        // its loads: the launch was a chain of 5-6 dependent round trips per wavefront at 24 wavefronts per CU, 81-100 us)
        const int32_t *tb = A.tbase + (int64_t)sig * A.tb_stride;
        const int64_t t = s / TW;
        const int64_t g = (int64_t)tb[t] - job.first + 1;          // (list entry first + k is detected knot first + k - 1)
        kfirst = g < 1 ? 1 : (g > idx ? idx : g);
        if (t + 1 < n_wt) {
            const int64_t g1 = (int64_t)tb[t + 1] - job.first + 1;
            const int64_t k_hi = g1 < kfirst ? kfirst : (g1 > idx ? idx : g1);   // the tile's knots are kfirst .. k_hi - 1; k_hi is staged too
            if (k_hi - (kfirst - 1) + 1 <= 320) {
                int32_t ek[5];
                double Kk[5], bk[5];
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int64_t k = kfirst - 1 + it * 64 + lane;
                    const bool ok = k <= k_hi;
                    ek[it] = ok ? e[k] : 0; Kk[it] = ok ? K[k] : 0.0; bk[it] = ok ? b[k] : 0.0;
                }
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int64_t k = kfirst - 1 + it * 64 + lane;
                    if (k <= k_hi) {
                        const int r = it * 64 + lane;
                        s_e[r] = ek[it]; s_K[r] = Kk[it]; s_b[r] = bk[it];
                        const int64_t p = (int64_t)ek[it] - s;
                        if (k >= kfirst && k < k_hi && p >= 0 && p < TW) atomicOr(&s_bits[p >> 6], 1ull << (p & 63));
                    }
                }
                staged = true;
            }
        }
    } else {
        int64_t l = 1, h = idx;         // a caller's list: searched
        while (l < h) {
            const int64_t mid = (l + h) >> 1;
            if (e[mid] < s) l = mid + 1; else h = mid;
        }
        kfirst = l;
    }
    if (!staged) {
        for (int64_t k0 = kfirst - 1; k0 <= idx; k0 += 64) {
            const int64_t k = k0 + lane;
            int64_t p = (int64_t)TW;                      // (beyond the list: behaves like a knot behind the tile)
            if (k <= idx) {
                const int32_t ek = e[k];
                const int r = (int)(k - (kfirst - 1));
                if (r < SL) { s_e[r] = ek; s_K[r] = K[k]; s_b[r] = b[k]; }
                p = (int64_t)ek - s;
                if (k >= kfirst && k < idx && p < TW) {
                    atomicOr(&s_bits[p >> 6], 1ull << (p & 63));
                    dense = dense || r + 1 >= SL;         // this knot's segment needs slot r + 1
                }
            }
            if (__any(k >= kfirst && (p >= TW || k >= idx))) break;   // the first knot behind the tile (or the list's end) is staged
        }
    }
    dense = __any(dense);
    __syncthreads();
    unsigned long long w[TW / 64];
    int pre[TW / 64];
    int acc = 0;                        // slot of the knot at or before the sample = the tile's knots at or before it
#pragma unroll
    for (int q = 0; q < TW / 64; ++q) {
        w[q] = s_bits[q];
        pre[q] = acc;
        acc += __popcll(w[q]);
    }
    // (wave-uniform cases hoisted out of the sample loop, the natural variant's linear last segment as a select: the loop body is
    //  straight-line code; the rows leave with nontemporal stores)
    auto eval = [&](auto dense_c, auto full_c) {
        constexpr bool DENSE = decltype(dense_c)::value, FULL = decltype(full_c)::value;
#pragma unroll
        for (int q = 0; q < TW / 64; ++q) {
            const int p = q * 64 + lane;
            const int64_t i = s + p;
            if (!FULL && i >= hi) continue;
            const unsigned long long upto = (lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull);
            const int r = pre[q] + __popcll(w[q] & upto);           // slot; the knot itself is j = kfirst - 1 + r
            const int64_t j = kfirst - 1 + r;
            int32_t ej, en;
            double Kj, Kn, bj, bn;
            if (!DENSE) { ej = s_e[r]; en = s_e[r + 1]; Kj = s_K[r]; Kn = s_K[r + 1]; bj = s_b[r]; bn = s_b[r + 1]; }
            else { ej = e[j]; en = e[j + 1]; Kj = K[j]; Kn = K[j + 1]; bj = b[j]; bn = b[j + 1]; }
            const double hj = (double)(en - ej);                                        // :86
            const double t = (double)(i - (int64_t)ej) * sweep_rcp(hj);                 // :115
            const double omt = 1 - t;
            const double h26 = hj * hj * (1.0 / 6);
            double c1 = h26 * ((omt * omt * omt - 1) + t) * bj;                         // :119 ((1-t)**3 - 1 + t)
            double c2 = h26 * (t * t * t - t) * bn;
            if (!NAK && j == idx - 2) { c1 = 0.0; c2 = 0.0; }                           // :117 the natural variant's linear last segment
            const double v = ((omt * Kj + t * Kn) + c1) + c2;
            __builtin_nontemporal_store(v, &o[i]);
            if (ro) __builtin_nontemporal_store(xs[i] - v, &ro[i]);
        }
    };
    const bool full = s + TW <= hi;
    if (dense) { if (full) eval(std::true_type{}, std::true_type{}); else eval(std::true_type{}, std::false_type{}); }
    else { if (full) eval(std::false_type{}, std::true_type{}); else eval(std::false_type{}, std::false_type{}); }
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
