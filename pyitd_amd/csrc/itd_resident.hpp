// itd_resident.hpp — the WHOLE decomposition of a short signal in ONE launch, the signal resident in LDS.
//
// The level-by-level engine (itd_kernels.hpp) streams a level's input from HBM and is launch bound below ~2^16 samples
// (10 dependent launches, ~6.5 us each) and traffic bound at 20 + 24 B per sample and level.  A signal of n <= 4096
// samples fits into one workgroup's LDS together with its knot arrays, so ONE workgroup runs the reference's driver loop
// (ITD.itd, ITD.py:384-432) for it from the first extraction to the stop rule: the signal is read once (4 or 8 B per
// sample), every result row is written once (8 B per sample and row), nothing else touches HBM, and a batch of short
// signals is one launch with one workgroup per signal.
//
// Per level, four workgroup barriers (a 64-thread workgroup's barriers are free):
//   scan    knot predicate of detect_peaks(x) U detect_peaks(-x) (ITD.py:44-59, 87-98) per sample, one __ballot word per
//           64 samples; every wavefront totals the words itself (exclusive prefix of their popcounts)
//   knots   ordered knot list by rank: index e_k and value x[e_k]; e_0 = 0, e_{m+1} = n-1 (ITD.py:93-98)
//   values  B_k in the reference's association (ITD.py:100-110) and the segment slopes (ITD.py:115-116) in one phase
//   map     baseline = B_k + slope_k * (x - x[e_k]) with k = knots at or before the sample (ITD.py:114-117),
//           baseline[n-1] = 0 (ITD.py:112: never written); the baseline replaces the signal in LDS, the thread keeps the
//           level's input in registers; rotation = x - baseline (ITD.py:119) leaves for row c at once
//   verdict the stop test of the pending baseline is the next level's scan (ITD.py:400-404): < 2 knots -> row c is
//           rewritten with baselines[c-1] (zeros at c = 0, ITD.py:404-416); c > max_iteration -> row c = rotation +
//           baseline (ITD.py:418-426); otherwise the baseline becomes the next extraction's input (ITD.py:428-432)
//
// The kernel handles FINITE data only.  A NaN or infinity in the input or in any baseline (0/0 on a leading or trailing
// plateau, ITD.py:115-116) raises SigState::res_fail and the workgroup leaves; itd_get_summary then repeats the call
// through the level-by-level engine, whose kernels carry the reference's NaN rules (nan_rules, k_nan_level0).
//
// Arithmetic: binary64, the reference's association, no contraction, IEEE division — bit-identical to k_extract.
#pragma once
#include "itd_kernels.hpp"

#pragma clang fp contract(off)

// timing-only ablations of k_resident (results are wrong by construction): 1 no knot values / slopes, 2 no prefix over the
// words, 4 no state resets, 8 no row / baseline stores, 16 no knot list, 32 no by-rank reads in the map, 64 never stop naturally
// (every build that is compared with another needs it: the level count must not depend on the values), 128 no scan loads
#ifndef ITD_RES_ABL
#define ITD_RES_ABL 0
#endif
namespace itd {

constexpr int kResidentMax = 4096;        // samples: 34 B of LDS per sample, 160 KB per workgroup on gfx950

// dynamic LDS of a workgroup that holds an n-sample signal: the signal, three by-rank knot arrays (value, B, slope), the
// knots' indices (16 bit), one ballot word and one prefix per 64 samples, three control words
__host__ __device__ constexpr int resident_pad(int n) { return (n + 63) & ~63; }
__host__ __device__ constexpr size_t resident_lds_bytes(int n)
{
    return (size_t)resident_pad(n) * (4 * 8 + 2) + (size_t)(resident_pad(n) / 64) * 12 + 16 + 8;
}

template <typename Tin, int TH, int SPT>
__global__ __launch_bounds__(TH) void k_resident(const Tin *__restrict__ xin, int64_t x_stride, int n, int M,
                                                   double *__restrict__ rows, int64_t rows_stride,
                                                   double *__restrict__ bases, int64_t bases_stride,
                                                   SigState *__restrict__ state, SigState *__restrict__ other_state)
{
    static_assert(TH % 64 == 0 && TH * SPT <= kResidentMax, "geometry");
    constexpr int W = TH / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int sig = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np2 = resident_pad(n), Q = np2 >> 6;
    double *xs = reinterpret_cast<double *>(lds_raw);          // the level's input; after the map: its baseline
    double *Xk = xs + np2;                                       // by rank: value of the level's input at knot k
    double *Bk = Xk + np2;                                       // knot value B_k
    double *Sk = Bk + np2;                                       // slope of the segment that starts at knot k
    unsigned long long *bal = reinterpret_cast<unsigned long long *>(Sk + np2);   // knot flags, one word per 64 samples
    int *ctl = reinterpret_cast<int *>(bal + Q) + Q + 1;         // [0]: a non-finite sample was seen
    unsigned short *ek = reinterpret_cast<unsigned short *>(ctl + 3);   // by rank: the knots' sample indices

    SigState *st = state + sig;
    const Tin *x = xin + (int64_t)sig * x_stride;
    double *rows_s = rows + (int64_t)sig * rows_stride;
    double *bases_s = bases ? bases + (int64_t)sig * bases_stride : nullptr;
    const double inf = __builtin_huge_val();

    if (tid == 0) {
        if (!(ITD_RES_ABL & 4)) {
            sig_state_reset(st);
            if (other_state) sig_state_reset(other_state + sig);   // the engine's other set, as k_finalize leaves it
        }
        ctl[0] = 0;
    }

    // the thread's samples: word q = wave + W j, sample 64 q + lane — consecutive lanes, consecutive samples (coalesced
    // rows, conflict-free LDS).  In registers: xr[] the input of the extraction in flight, xb[] its baseline, and per word
    // (wave-uniform) the knot flags of the level's input and the number of knots in front of the word
    double xr[SPT], xb[SPT];
    unsigned long long bm[SPT];
    int pr[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
        const int q = wave + W * j, i = q * 64 + lane;
        xr[j] = 0.0;
        xb[j] = 0.0;
        if (q < Q) {
            if (i < n) xr[j] = (double)x[i];
            xs[i] = xr[j];
        }
    }
    __syncthreads();

    // Knot flags of the signal in xs (cur[] = the thread's own samples of it; interior samples 1 .. n-2, ITD.py:70-73) into
    // bm[], the words' exclusive prefix into pr[]; returns the knot total.  One barrier: every wavefront totals the words
    // itself.  ctl[0] is raised if a sample is not finite.
    auto scan = [&](const double (&cur)[SPT]) -> int {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j, i = q * 64 + lane;
            bm[j] = 0;
            if (q < Q) {
                bool f = false, bad = false;
                if (i < n) {
                    const double c = cur[j];
                    bad = !(__builtin_fabs(c) < inf);
                    if (i >= 1 && i <= n - 2) {
                        const double d0 = c - ((ITD_RES_ABL & 128) ? 0.5 : xs[i - 1]), d1 = ((ITD_RES_ABL & 128) ? 0.25 : xs[i + 1]) - c;   // dx[i-1], dx[i] (ITD.py:44)
                        f = (d1 > 0.0 && d0 <= 0.0) || (d1 < 0.0 && d0 >= 0.0);   // valley of x or of -x (ITD.py:59, 87-88)
                    }
                }
                bm[j] = __ballot(f);
                const unsigned long long bb = __ballot(bad);
                if (lane == 0) {
                    bal[q] = bm[j];
                    if (bb) ctl[0] = 1;
                }
            }
        }
        __syncthreads();
        const int cnt = lane < Q ? __popcll(bal[lane]) : 0;
        // inclusive prefix over the 64 lanes on the DPP path (no LDS round trips): Hillis-Steele inside the rows of 16 lanes
        // (row_shr 1, 2, 4, 8: a lane without a source inside its row keeps the 0 it is given), then row 0's / rows 0-1's
        // totals into the rows behind them (row_bcast:15 to rows 1 and 3, row_bcast:31 to rows 2 and 3)
        int inc = cnt;
        if (!(ITD_RES_ABL & 2)) {
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);
        }
        const int excl = inc - cnt;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j;
            pr[j] = q < Q ? __builtin_amdgcn_readlane(excl, q) : 0;
        }
        return __builtin_amdgcn_readlane(inc, 63);
    };

    int m = scan(xr);
    if (tid == 0) st->m[0] = m;
    if (ctl[0]) {
        if (tid == 0) st->res_fail = 1;
        return;
    }

    for (int c = 0;; ++c) {
        // ---- ordered knot list of the level's input (knots by rank 1 .. m, the two ends at 0 and m+1) ----------------
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j, i = q * 64 + lane;
            if (q < Q && ((bm[j] >> lane) & 1ull) && !(ITD_RES_ABL & 16)) {
                const int k = pr[j] + __popcll(bm[j] & ((1ull << lane) - 1ull)) + 1;
                ek[k] = (unsigned short)i;
                Xk[k] = xr[j];
            }
        }
        if (tid == 0) {
            ek[0] = 0;                       // ITD.py:96
            Xk[0] = xs[0];
            ek[m + 1] = (unsigned short)(n - 1);   // ITD.py:98
            Xk[m + 1] = xs[n - 1];
        }
        __syncthreads();
        // ---- knot values (ITD.py:100-110) and slopes (ITD.py:115-116; 0/0 on an end segment is the reference's NaN: the
        //      next scan catches it).  A wavefront takes 64 consecutive knots and the 63 segments between them — B_{k+1} comes
        //      from the next lane — so consecutive chunks overlap by one knot and no barrier separates values from slopes ------
        for (int cidx = wave; 63 * cidx <= m + 1 && !(ITD_RES_ABL & 1); cidx += W) {
            const int k = 63 * cidx + lane;
            double v = 0.0, x0 = 0.0, xp1 = 0.0;
            if (k <= m + 1) {
                x0 = Xk[k];
                if (k == 0) {
                    v = (xs[0] + xs[1]) / 2.0;                   // numpy.mean(x[:2])
                    xp1 = Xk[1];
                } else if (k == m + 1) {
                    v = (xs[n - 2] + xs[n - 1]) / 2.0;           // numpy.mean(x[-2:])
                } else {
                    const int e0 = ek[k - 1], e1 = ek[k], e2 = ek[k + 1];
                    const double xm1 = Xk[k - 1];
                    xp1 = Xk[k + 1];
                    const double frac = (double)(e1 - e0) / (double)(e2 - e0);
                    const double t = frac * (xp1 - xm1);
                    const double u = xm1 + t;
                    v = 0.5 * u + 0.5 * x0;
                }
                Bk[k] = v;
            }
            const double vn = __shfl_down(v, 1, 64);
            if (lane < 63 && k <= m) Sk[k] = (vn - v) / (xp1 - x0);
        }
        __syncthreads();
        // ---- map (ITD.py:112-117): the baseline replaces the signal in LDS; rotation = x - baseline (ITD.py:119) leaves at
        //      once unless this is the extraction behind the last requested one (its row is never the rotation) -----------
        double *row = rows_s + (int64_t)c * n;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j, i = q * 64 + lane;
            if (q < Q && i < n) {
                const int k = pr[j] + __popcll(bm[j] & ((2ull << lane) - 1ull));   // knots at or before the sample
                const double b = (ITD_RES_ABL & 32) ? xr[j] * 0.75 + k : (i == n - 1) ? 0.0 : Bk[k] + Sk[k] * (xr[j] - Xk[k]);
                xb[j] = b;
                xs[i] = b;
                if (c <= M && !(ITD_RES_ABL & 8)) {
                    row[i] = xr[j] - b;
                    if (bases_s) bases_s[(int64_t)c * n + i] = b;
                }
            }
        }
        __syncthreads();
        // ---- the pending baseline's stop test = the next level's scan (ITD.py:400-404) -------------------------------
        m = scan(xb);
        if (tid == 0) st->m[c + 1] = m;
        if (ctl[0] && !(ITD_RES_ABL & 64)) {
            if (tid == 0) st->res_fail = 1;
            return;
        }
        if (m < 2 && !(ITD_RES_ABL & 64)) {            // "No more decompositions possible": row c = baselines[c-1], the zero row at c = 0
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q && i < n) row[i] = c == 0 ? 0.0 : xr[j];
            }
            if (tid == 0) {
                st->stop_level = c + 1;
                st->fin_stopped = 1;
                st->fin_stop_level = c + 1;
            }
            return;
        }
        if (c > M) {            // "Out of time!": row c = rotation + baseline (ITD.py:418-426)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q && i < n) {
                    const double r = xr[j] - xb[j];
                    row[i] = r + xb[j];
                }
            }
            return;             // fin_stopped stays 0
        }
#pragma unroll
        for (int j = 0; j < SPT; ++j) xr[j] = xb[j];   // the baseline is the next extraction's input (ITD.py:428-432)
    }
}

}  // namespace itd
