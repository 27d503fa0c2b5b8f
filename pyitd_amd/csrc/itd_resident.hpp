// itd_resident.hpp — the WHOLE decomposition of a short signal in ONE launch, the signal resident in LDS.
//
// The level-by-level engine (itd_kernels.hpp) streams a level's input from HBM and is launch bound below ~2^16 samples
// (10 dependent launches, ~6.5 us each) and traffic bound at 20 + 24 B per sample and level.  A signal of n <= 8192
// samples fits into one workgroup's LDS together with a window of its knot arrays, so ONE workgroup runs the reference's driver loop
// (ITD.itd, ITD.py:384-432) for it from the first extraction to the stop rule: the signal is read once (4 or 8 B per
// sample), every result row is written once (8 B per sample and row), nothing else touches HBM, and a batch of short
// signals is one launch with one workgroup per signal.
//
// Per level, one workgroup barrier for the scan and three per pass over the level's ranks (one pass unless the level has more
// knots than the window holds; a 64-thread workgroup's barriers are free):
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
// Baselines that go NaN (0/0 on a leading or trailing plateau, ITD.py:115-116) follow the reference's NaN rules right here
// (scan: the NaN branch of detect_peaks, the in-place NaN -> +inf write, ITD.py:46-51, 64-68); infinities are plain data.
// A NaN in the CALLER'S signal follows them as well (its knots: NaN-branch valleys + peaks of the mutated array, ITD.py:87-95);
// only an engine told to reject NaN input gets SigState::res_fail: itd_get_summary then repeats the call level by level.
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
// wavefronts per SIMD the four-samples-per-thread instances are compiled for (8: 64 VGPRs)
#ifndef ITD_RES_MINW
#define ITD_RES_MINW 8
#endif
namespace itd {

constexpr int kResidentMax = 8192;        // samples: the signal (8 B per sample) + a window of by-rank knot arrays in 160 KB of LDS

// The by-rank knot arrays hold a WINDOW of `cw` consecutive segments (+ the knots either side that their knot values need); a
// level with more knots than that takes several passes over its ranks (knot list, values, map per pass).  The automatic window
// is 0.4 knots per sample: one pass for every level of ordinary signals (sines + noise: 0.40 / 0.12 / 0.04 ... knots per
// sample), two for the first level of white noise (0.67) — and 18 instead of 34 B of LDS per sample, i.e. about twice the
// workgroups per CU, which is what a latency-bound launch is short of.
__host__ __device__ constexpr int resident_pad(int n) { return (n + 63) & ~63; }
__host__ __device__ constexpr int resident_auto_window(int n)
{
    return resident_pad(n) * 2 / 5 < 64 ? 64 : resident_pad(n) * 2 / 5;
}
// dynamic LDS: the signal, three by-rank arrays of cw + 3 doubles (value, B, slope), the knots' 16-bit indices, one ballot word
// per 64 samples, the level's four end samples, control words
__host__ __device__ constexpr int resident_cap(int cw) { return (cw + 4 + 3) & ~3; }   // by-rank slots of a window of cw segments
__host__ __device__ constexpr size_t resident_lds_bytes(int n, int cw)
{
    return (size_t)resident_pad(n) * 8 + (size_t)resident_cap(cw) * (3 * 8 + 2) + (size_t)(resident_pad(n) / 64) * 12 + 4 * 8 + 32 + 8;
}
constexpr size_t kResidentLdsMax = 160 * 1024;

// 8-byte row stores / input loads through raw buffer descriptors (base = the row, extent = n samples): the lanes of the last,
// partly filled 64-sample word need no guard — what lies beyond the row is not written and reads as 0
__device__ __forceinline__ void res_store(__amdgpu_buffer_rsrc_t r, int i, double v)
{
    using U2 = unsigned __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(U2, v), r, i * 8, 0, kBufNT);   // result rows are never read again
}
__device__ __forceinline__ double res_load(__amdgpu_buffer_rsrc_t r, int i, const double *)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, i * 8, 0, 0));
}
__device__ __forceinline__ double res_load(__amdgpu_buffer_rsrc_t r, int i, const float *)
{
    return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, i * 4, 0, 0));
}

template <typename Tin, int TH, int SPT>
__global__ __launch_bounds__(TH, SPT == 4 ? ITD_RES_MINW : 4) void k_resident(const Tin *__restrict__ xin, int64_t x_stride, int n, int M, int cw,
                                                   int nan_follow /* a NaN in the caller's signal: 1 = the reference's rules (below), 0 = leave (res_fail) */,
                                                   double *__restrict__ rows, int64_t rows_stride,
                                                   double *__restrict__ bases, int64_t bases_stride,
                                                   SigState *__restrict__ state, SigState *__restrict__ other_state)
{
    static_assert(TH % 64 == 0 && TH * SPT <= kResidentMax, "geometry");
    constexpr int W = TH / 64;
    constexpr int WPL = (TH * SPT / 64 + 63) / 64;   // ballot words a lane totals in the prefix (1 up to 4096 samples, 2 up to 8192)
    static_assert(WPL == 1 || WPL == 2, "prefix geometry");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int sig = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np2 = resident_pad(n), Q = np2 >> 6;
    const int cap = resident_cap(cw);                            // by-rank slots (slot 0 = the knot in front of the window)
    double *xs = reinterpret_cast<double *>(lds_raw);          // the level's input; after the map: its baseline
    double *Xk = xs + np2;                                       // by slot: value of the level's input at the knot
    double *Bk = Xk + cap;                                       // knot value B_k
    double *Sk = Bk + cap;                                       // slope of the segment that starts at the knot
    double *ends = Sk + cap;                                     // x[0], x[1], x[n-2], x[n-1] of the level's input (ITD.py:101-102)
    unsigned long long *bal = reinterpret_cast<unsigned long long *>(ends + 4);   // knot flags, one word per 64 samples
    int *pre = reinterpret_cast<int *>(bal + Q);                 // knots in front of word q (written and read by the word's wavefront)
    int *ctl = pre + Q;                                          // two sets (by scan parity) of: -, a NaN was seen, the NaN branch's two counts
    unsigned short *ek = reinterpret_cast<unsigned short *>(ctl + 8);   // by slot: the knots' sample indices

    SigState *st = state + sig;
    const Tin *x = xin + (int64_t)sig * x_stride;
    double *rows_s = rows + (int64_t)sig * rows_stride;
    double *bases_s = bases ? bases + (int64_t)sig * bases_stride : nullptr;
    const double inf = __builtin_huge_val();

    // the signal's state as a decomposition expects to find it (sig_state_reset, spread over the first wavefront's lanes) — and
    // the engine's other set, as k_finalize leaves it for the call after this one
    if (tid < kMaxLevels && !(ITD_RES_ABL & 4)) {
        SigState *const os = other_state ? other_state + sig : nullptr;
        st->m[tid] = -1;
        st->c_delta[tid] = 0;
        if (os) { os->m[tid] = -1; os->c_delta[tid] = 0; }
        if (tid == 0) {
            st->stop_level = -1; st->nan_mask = 0; st->in_nan = 0; st->l0_fail = 0;
            st->fin_stopped = 0; st->fin_stop_level = -1; st->kf_fail = 0; st->res_fail = 0; st->skip = 0;
            if (os) {
                os->stop_level = -1; os->nan_mask = 0; os->in_nan = 0; os->l0_fail = 0;
                os->fin_stopped = 0; os->fin_stop_level = -1; os->kf_fail = 0; os->res_fail = 0; os->skip = 0;
            }
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) ctl[k] = 0;
    }

    // the thread's samples: word q = wave + W j, sample 64 q + lane — consecutive lanes, consecutive samples (coalesced
    // rows, conflict-free LDS).  Registers hold only xr[], the input of the extraction in flight: the words' knot flags and
    // prefixes live in LDS (bal[], pre[]), the baseline in xs — 64 VGPRs, so that the CU's wavefront slots, not registers,
    // bound the workgroups per CU once the rank window has cut the LDS per sample
    // Samples of the padding (n .. np2-1) are 0 in registers and LDS throughout: they flag nothing and are never stored.
    double xr[SPT];
    {
        const __amdgpu_buffer_rsrc_t rx = tile_rsrc(x, (int64_t)n * (int64_t)sizeof(Tin));
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j, i = q * 64 + lane;
            xr[j] = 0.0;
            if (q < Q) {
                xr[j] = res_load(rx, i, x);
                xs[i] = xr[j];
            }
        }
    }
    __syncthreads();

    // Knot flags of the signal in xs (interior samples 1 .. n-2, ITD.py:70-73) into bal[], the exclusive prefix of the
    // wavefront's own words into pre[]; returns the number of flags (the knots of the next extraction) and leaves in m_stop
    // what the reference's stop test counts (ITD.py:400-402).  The two differ only if the array holds a NaN — a baseline does
    // after a leading / trailing plateau (0/0 at ITD.py:115-116): detect_peaks(b) then takes its NaN branch (NaN differences
    // count as +inf, NaN samples and their neighbours are no peaks, ITD.py:46-51, 64-68) and OVERWRITES the NaNs with +inf in
    // place (:50; no copy at :41), detect_peaks(-b) and the next extraction see the mutated array (plain rules: no NaN left;
    // infinities just flow through the IEEE arithmetic).  `input`: the caller's signal — there the first extraction's knots
    // themselves are detect_peaks(x) under the NaN branch + detect_peaks(-x) of the mutated array (ITD.py:87-95), and the
    // extraction works on the mutated values (the caller's array is not written); without nan_follow: returns -1.
    // One barrier (four if a NaN was seen); every wavefront totals the words itself.
    int m_stop = 0, n_scans = 0;
    auto scan = [&](bool input, __amdgpu_buffer_rsrc_t r_mut, bool store_mut) -> int {
        int *cs = ctl + 4 * (n_scans & 1), *co = ctl + 4 * ((n_scans & 1) ^ 1);   // this scan's counters, the previous scan's
        ++n_scans;
        if (tid == 0) co[1] = co[2] = co[3] = 0;   // last read before the barriers of the phases in between
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int q = wave + W * j, i = q * 64 + lane;
            if (q < Q) {
                const double c = xs[i];
                // dx[i-1], dx[i] (ITD.py:44); the reads next to the signal's ends stay inside xs (word 0 starts at sample 0, the
                // padding ends the last word) and only interior samples can flag
                const double d0 = c - ((ITD_RES_ABL & 128) ? 0.5 : xs[i > 0 ? i - 1 : 0]), d1 = ((ITD_RES_ABL & 128) ? 0.25 : xs[i < np2 - 1 ? i + 1 : i]) - c;
                const bool f = ((d1 > 0.0 && d0 <= 0.0) || (d1 < 0.0 && d0 >= 0.0)) && i >= 1 && i <= n - 2;   // valley of x or of -x (ITD.py:59, 87-88)
                const unsigned long long bw = __ballot(f);
                const unsigned long long bb = __ballot(c != c);
                if (lane == 0) {
                    bal[q] = bw;
                    if (bb) cs[1] = 1;
                }
            }
        }
        __syncthreads();
        const bool has_nan = cs[1] != 0;
        if (has_nan) {
            if (input && !nan_follow) return -1;
            // detect_peaks(b), NaN branch, on the array as it is (ITD.py:44-68): valleys among the samples whose three values are
            // no NaN; a difference that is NaN all the same (inf - inf) counts as +inf
            int acc = 0;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q) {
                    const double a = xs[i > 0 ? i - 1 : 0], c = xs[i], b = xs[i < np2 - 1 ? i + 1 : i];
                    double vil = b - c, vix = c - a;
                    if (vil != vil) vil = inf;
                    if (vix != vix) vix = inf;
                    const bool ok = vil > 0.0 && vix <= 0.0 && a == a && c == c && b == b && i >= 1 && i <= n - 2;
                    const unsigned long long okw = __ballot(ok);
                    acc += __popcll(okw);
                    if (input && lane == 0) bal[q] = okw;   // the caller's signal: these ARE the valleys of its knot list (ITD.py:87)
                }
            }
            if (lane == 0 && acc) atomicAdd(&cs[2], acc);
            __syncthreads();
            // the mutation (ITD.py:50), also in the stored baseline row: baselines[counter] is copied after it (ITD.py:429)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q) {
                    const double c = xs[i];
                    if (c != c) {
                        xs[i] = inf;
                        if (store_mut) res_store(r_mut, i, inf);
                    }
                }
            }
            __syncthreads();
            // detect_peaks(-b) on the mutated array, plain rules (ITD.py:401) — and the flags of the next extraction
            acc = 0;
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q) {
                    const double c = xs[i];
                    const double d0 = c - xs[i > 0 ? i - 1 : 0], d1 = xs[i < np2 - 1 ? i + 1 : i] - c;
                    const bool inner = i >= 1 && i <= n - 2;
                    const bool pk = d1 < 0.0 && d0 >= 0.0 && inner;
                    const bool f = ((d1 > 0.0 && d0 <= 0.0) || pk) && inner;
                    const unsigned long long bw = __ballot(f), pw = __ballot(pk);
                    acc += __popcll(pw);
                    // a baseline: the next extraction finds its knots on the mutated array, plain rules; the caller's signal:
                    // valleys from the NaN branch (above) + peaks of the mutated array (ITD.py:87-95)
                    if (lane == 0) bal[q] = input ? (bal[q] | pw) : bw;
                }
            }
            if (lane == 0 && acc) atomicAdd(&cs[3], acc);
            __syncthreads();
        }
        if (tid == 0) {   // read by this thread (knot list) and, behind the knot list's barrier, by everybody (knot values)
            ends[0] = xs[0];
            ends[1] = xs[1];
            ends[2] = xs[n - 2];
            ends[3] = xs[n - 1];
        }
        // inclusive prefix over the 64 lanes on the DPP path (no LDS round trips): Hillis-Steele inside the rows of 16 lanes
        // (row_shr 1, 2, 4, 8: a lane without a source inside its row keeps the 0 it is given), then row 0's / rows 0-1's
        // totals into the rows behind them (row_bcast:15 to rows 1 and 3, row_bcast:31 to rows 2 and 3).  A lane totals WPL words.
        const int c0 = WPL * lane < Q ? __popcll(bal[WPL * lane]) : 0;
        const int c1 = (WPL == 2 && 2 * lane + 1 < Q) ? __popcll(bal[2 * lane + 1]) : 0;
        const int cnt = c0 + c1;
        int inc = cnt;
        if (!(ITD_RES_ABL & 2)) {
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x142, 0xa, 0xf, false);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x143, 0xc, 0xf, false);
        }
        // the prefixes of the wavefront's own words (nobody else reads them: no barrier)
        const int excl = inc - cnt;
        const int w0 = WPL * lane;
        if (w0 < Q && (w0 % W) == wave) pre[w0] = excl;
        if (WPL == 2 && w0 + 1 < Q && ((w0 + 1) % W) == wave) pre[w0 + 1] = excl + c0;
        const int total = __builtin_amdgcn_readlane(inc, 63);
        m_stop = has_nan ? cs[2] + cs[3] : total;
        return total;
    };

    const __amdgpu_buffer_rsrc_t r_none = tile_rsrc(nullptr, 0);
    int m = scan(true, r_none, false);
    if (m < 0) {   // a NaN in the caller's signal and the engine was told to reject it: the level-by-level engine reports it
        if (tid == 0) st->res_fail = 1;
        return;
    }
    if (tid == 0) st->m[0] = m;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {   // the mutated values, if the scan met a NaN
        const int q = wave + W * j, i = q * 64 + lane;
        if (q < Q) xr[j] = xs[i];
    }

    for (int c = 0;; ++c) {
        const __amdgpu_buffer_rsrc_t r_row = tile_rsrc(rows_s + (int64_t)c * n, (int64_t)n * 8);
        const __amdgpu_buffer_rsrc_t r_base = tile_rsrc(bases_s ? bases_s + (int64_t)(c <= M ? c : 0) * n : nullptr, bases_s ? (int64_t)n * 8 : 0);
        // ---- passes over the level's ranks: segments k0 .. kend-1 (segment k = samples e_k <= i < e_{k+1}); the window's
        //      slots hold the knots k0-1 .. kend+1, slot = k - k0 + 1 -------------------------------------------------------
        for (int k0 = 0; k0 <= m; k0 += cw) {
            const int kend = k0 + cw < m + 1 ? k0 + cw : m + 1;
            // nothing derived from the lane / wavefront index is carried across the passes (registers)
            int lane_p = lane, wave_p = wave;
            asm volatile("" : "+v"(lane_p), "+s"(wave_p));
            // -- ordered knot list (knots by rank 1 .. m; the two ends are knots 0 and m+1, ITD.py:93-98)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave_p + W * j, i = q * 64 + lane_p;
                if (q < Q && !(ITD_RES_ABL & 16)) {
                    const unsigned long long bw = bal[q];
                    if ((bw >> lane_p) & 1ull) {
                        const int k = pre[q] + __popcll(bw & ((1ull << lane_p) - 1ull)) + 1;
                        if (k >= k0 - 1 && k <= kend + 1) {
                            ek[k - k0 + 1] = (unsigned short)i;
                            Xk[k - k0 + 1] = xr[j];
                        }
                    }
                }
            }
            if (tid == 0) {
                if (k0 <= 1) {                   // knot 0: ITD.py:96
                    ek[1 - k0] = 0;
                    Xk[1 - k0] = ends[0];
                }
                if (kend + 1 >= m + 1) {         // knot m+1: ITD.py:98
                    ek[m + 1 - k0 + 1] = (unsigned short)(n - 1);
                    Xk[m + 1 - k0 + 1] = ends[3];
                }
            }
            __syncthreads();
            // -- knot values (ITD.py:100-110) and slopes (ITD.py:115-116; 0/0 on an end segment is the reference's NaN: the
            //    next scan catches it).  A wavefront takes 64 consecutive knots and the 63 segments between them — B_{k+1} comes
            //    from the next lane — so consecutive chunks overlap by one knot and no barrier separates values from slopes
            for (int cidx = wave_p; k0 + 63 * cidx <= kend && !(ITD_RES_ABL & 1); cidx += W) {
                const int k = k0 + 63 * cidx + lane_p, sl = k - k0 + 1;
                double v = 0.0, x0 = 0.0, xp1 = 0.0;
                if (k <= kend) {
                    x0 = Xk[sl];
                    if (k == 0) {
                        v = (ends[0] + ends[1]) / 2.0;               // numpy.mean(x[:2])
                        xp1 = Xk[sl + 1];
                    } else if (k == m + 1) {
                        v = (ends[2] + ends[3]) / 2.0;               // numpy.mean(x[-2:])
                    } else {
                        const int e0 = ek[sl - 1], e1 = ek[sl], e2 = ek[sl + 1];
                        const double xm1 = Xk[sl - 1];
                        xp1 = Xk[sl + 1];
                        const double frac = (double)(e1 - e0) / (double)(e2 - e0);      // (int_ratio here: no difference, profiles/r05/experiments)
                        const double t = frac * (xp1 - xm1);
                        const double u = xm1 + t;
                        v = 0.5 * u + 0.5 * x0;
                    }
                    Bk[sl] = v;
                }
                const double vn = __shfl_down(v, 1, 64);
                if (lane_p < 63 && k < kend) Sk[sl] = (vn - v) / (xp1 - x0);
            }
            __syncthreads();
            // -- map (ITD.py:112-117) of the samples whose segment lies in the window: the baseline replaces the signal in LDS;
            //    rotation = x - baseline (ITD.py:119) leaves at once unless this is the extraction behind the last requested
            //    one (its row is never the rotation)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave_p + W * j, i = q * 64 + lane_p;
                if (q < Q) {
                    const int k = pre[q] + __popcll(bal[q] & ((2ull << lane_p) - 1ull));   // knots at or before the sample
                    if (k >= k0 && k < kend) {
                        const int sl = k - k0 + 1;
                        const double b = (ITD_RES_ABL & 32) ? xr[j] * 0.75 + k : (i >= n - 1) ? 0.0 : Bk[sl] + Sk[sl] * (xr[j] - Xk[sl]);
                        xs[i] = b;
                        if (c <= M && !(ITD_RES_ABL & 8)) {
                            res_store(r_row, i, xr[j] - b);
                            if (bases_s) res_store(r_base, i, b);
                        }
                    }
                }
            }
            __syncthreads();
        }
        // ---- the pending baseline's stop test = the next level's scan (ITD.py:400-404) -------------------------------
        m = scan(false, r_base, bases_s && c <= M);
        if (tid == 0) st->m[c + 1] = m_stop;
        if (m_stop < 2 && !(ITD_RES_ABL & 64)) {            // "No more decompositions possible": row c = baselines[c-1], the zero row at c = 0
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int q = wave + W * j, i = q * 64 + lane;
                if (q < Q) res_store(r_row, i, c == 0 ? 0.0 : xr[j]);
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
                if (q < Q) {
                    const double b = xs[i];
                    const double r = xr[j] - b;
                    res_store(r_row, i, r + b);
                }
            }
            return;             // fin_stopped stays 0
        }
#pragma unroll
        for (int j = 0; j < SPT; ++j) {   // the baseline is the next extraction's input (ITD.py:428-432)
            const int q = wave + W * j, i = q * 64 + lane;
            if (q < Q) xr[j] = xs[i];
        }
    }
}

}  // namespace itd
