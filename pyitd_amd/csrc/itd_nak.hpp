// itd_nak.hpp — the FITPACK flavour of the baseline (numba_accelerated_itd.py:182-211, MEITD.py:303-338), solved PARALLEL IN
// THE KNOTS for long single signals.
//
// What the reference computes there is scipy.interpolate.splrep(x, y, k=3) with its default s = 0 (numba_accelerated_itd.py:84):
// the cubic spline that INTERPOLATES the baseline knot values at the data sites e = [0, knots, n-1] with FITPACK's knot choice
// t = [x0 x4, x2 .. x_{m-3}, x_{m-1} x4] — i.e. the not-a-knot interpolant (third derivative continuous at x1 and x_{m-2}) —
// evaluated at every sample (numba_splev, :89-164; for equally spaced sites its interval formula picks the true interval,
// thanks to its clamps).  itd_fitpack.hpp restates FITPACK's own row-by-row Givens sweep (serial in the knots, one GPU thread
// per signal, bit-level against scipy): right for thousands of image rows, a single serial lane for ONE long signal
// (MEITD.py:344-549 calls the operator on single 1-D signals in a loop).
//
// Here the same spline comes from its second derivatives M_i at the sites (the classical moment form):
//     mu_i M_{i-1} + 2 M_i + lambda_i M_{i+1} = 6 f[x_{i-1}, x_i, x_{i+1}],   mu_i = h_{i-1} / (h_{i-1} + h_i),  i = 1 .. m-2
// with the not-a-knot ends folded into the first and last row ((2 + r) M_1 + (1 - r) M_2 = d_1, r = h_0 / h_1; mirrored at the
// other end) — a strictly diagonally dominant tridiagonal system.  Thomas' two recurrences (pivots + forward, then back
// substitution) contract, so a thread solves a run of kNakRun consecutive unknowns exactly (to far below rounding) by starting
// kNakWarm steps early from zero — no inter-thread exchange, every thread independent, two launches.  How far below: a step passes a
// perturbation of dp on with the factor mu_j / den_j and one of M (going down) with lambda_j / den_j, den_j >= 3/2; mu_j =
// 1 / (1 + rho_j) with rho_j = h_j / h_{j-1}, and over k consecutive steps the rho's multiply to h_last / h_first >= 1 / n (the
// spacings are whole numbers of samples), so prod(1 + rho_j) >= (1 + n^(-1/k))^k.  For k = kNakWarm = 64 and n = 2^31 that is
// 2^49.8, times (3/2)^64 = 2^37.4 from the pivots: a start error is damped by 2^-87 at least whatever the knots' spacings (the
// pivots' own recurrence contracts by 1/9 per step) — rounding is 2^-53.  (Rounds 3-4 warmed up over 128 steps on the cruder
// bound of 1/2 per pair of steps.)  The result equals FITPACK's to rounding (tests: 1e-10 of the signal's
// scale against the reference-generated vectors; the north star allows 1e-6), not bit for bit: the serial form stays the one
// the bit-level tests and the image sweeps use (itd_set_spline_solver).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "itd_cubic.hpp"
#include "itd_spline.hpp"

#pragma clang fp contract(off)

namespace itd {

#ifndef ITD_NAK_RUN
#define ITD_NAK_RUN 16
#endif
#ifndef ITD_NAK_WARM
#define ITD_NAK_WARM 64
#endif
#ifndef ITD_NAK_DIV
#define ITD_NAK_DIV 0            // 1 (diagnostic builds): the forward step's two quotients as IEEE divisions
#endif
constexpr int kNakRun = ITD_NAK_RUN, kNakWarm = ITD_NAK_WARM;

// jobs of the batch from the detected knots: sites e = kidx[b][0 .. knots+1] (first = 0, idx = knots + 1); fewer than
// max(min_extrema, 2) knots: invalid = the signal is its own baseline (numba_accelerated_itd.py:188-190)
__global__ void k_nak_jobs(CubicJob *__restrict__ jobs, int batch, const int32_t *__restrict__ totals, int min_extrema)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    CubicJob j;
    const int knots = totals[2 * b];
    j.first = 0;
    j.idx = knots + 1;
    j.status = totals[2 * b + 1] ? 2 : 0;
    j.valid = knots >= min_extrema && knots >= 2;
    jobs[b] = j;
}

// baseline knot values at the sites (numba_accelerated_itd.py:196-206), into A.K
__global__ void k_nak_values(CubicArgs A)
{
    const int sig = blockIdx.y;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    if (!job.valid) return;
    const int m = job.idx + 1;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    A.K[(int64_t)sig * A.a_stride + k] = spline_knot_value(A.x + (int64_t)sig * A.x_stride, A.n, A.e + (int64_t)sig * A.e_stride, m, k);
}

// row j (1 <= j <= m-2) of the system: sub M_{j-1} + diag M_j + sup M_{j+1} = rhs
struct NakRow { double sub, diag, sup, rhs; };
__device__ __forceinline__ NakRow nak_row(const int32_t *__restrict__ e, const double *__restrict__ S, int m, int j)
{
    const double hm = (double)(e[j] - e[j - 1]), hi = (double)(e[j + 1] - e[j]);
    NakRow r;
    r.rhs = 6 * ((S[j + 1] - S[j]) / hi - (S[j] - S[j - 1]) / hm) / (hm + hi);
    r.sub = hm / (hm + hi);
    r.sup = 1 - r.sub;
    r.diag = 2;
    // m >= 4 (at least 2 knots), so the first row (j = 1) and the last (j = m-2) are different rows
    if (j == 1) { const double q = hm / hi; r.diag = 2 + q; r.sup = 1 - q; r.sub = 0; }              // M_0 = (1 + q) M_1 - q M_2
    if (j == m - 2) { const double q = hi / hm; r.sub = 1 - q; r.diag = 2 + q; r.sup = 0; }          // M_{m-1} = (1 + q) M_{m-2} - q M_{m-3}
    return r;
}

// One step of the forward elimination: den = diag - sub cp_{j-1}; cp_j = sup / den; dp_j = (rhs - sub dp_{j-1}) / den.  The pivot lies in
// [1.5, 2 + q] for every row (cp stays in (-1, 1/2]; the not-a-knot rows' q = a ratio of knot spacings > 0): its reciprocal comes from
// v_rcp_f64 and two Newton steps (~1 ulp, no scaling, no fix-up) and serves both quotients — the two IEEE divisions were ~60 dependent
// instructions of every step of a chain of kNakWarm + kNakRun of them, the whole time of an extraction of a short signal
// (51 -> ~15 us for 3000 samples inside MEITD's loop).  The recurrences contract, so the step's rounding does not accumulate.
__device__ __forceinline__ void nak_step(const NakRow &r, double &cp, double &dp)
{
#if ITD_NAK_DIV
    const double den = r.diag - r.sub * cp;
    cp = r.sup / den;
    dp = (r.rhs - r.sub * dp) / den;
#else
    const double inv = sweep_rcp(r.diag - r.sub * cp);
    cp = r.sup * inv;
    dp = (r.rhs - r.sub * dp) * inv;
#endif
}

// the rows of the system, one thread per row, once: every row is met by kNakWarm / kNakRun + 1 = 5 forward runs, and a row costs
// five divisions.  sub and rhs are stored; sup = 1 - sub and diag = 2 except in the two end rows, which the runs rebuild
__global__ __launch_bounds__(256) void k_nak_rows(CubicArgs A, double *__restrict__ sub_ws, double *__restrict__ rhs_ws)
{
    const int sig = blockIdx.y;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    if (!job.valid) return;
    const int m = job.idx + 1;
    const int j = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (j > m - 2) return;
    const NakRow r = nak_row(A.e + (int64_t)sig * A.e_stride, A.K + (int64_t)sig * A.a_stride, m, j);
    sub_ws[(int64_t)sig * A.a_stride + j] = r.sub;
    rhs_ws[(int64_t)sig * A.a_stride + j] = r.rhs;
}

// forward elimination of a run of kNakRun rows per thread: cp_j = sup / den, dp_j = (rhs - sub dp_{j-1}) / den,
// den = diag - sub cp_{j-1}; into A.bf (dp) and A.b (cp, overwritten by the back substitution's M afterwards: separate array)
__global__ __launch_bounds__(64) void k_nak_forward(CubicArgs A, double *__restrict__ cp_ws, const double *__restrict__ sub_ws,
                                                    const double *__restrict__ rhs_ws)
{
    const int sig = blockIdx.y;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    if (!job.valid) return;
    const int m = job.idx + 1;
    const int s = 1 + (blockIdx.x * blockDim.x + threadIdx.x) * kNakRun;
    if (s > m - 2) return;
    const int32_t *e = A.e + (int64_t)sig * A.e_stride;
    const double *S = A.K + (int64_t)sig * A.a_stride;
    const double *subv = sub_ws + (int64_t)sig * A.a_stride, *rhsv = rhs_ws + (int64_t)sig * A.a_stride;
    double *dpv = A.bf + (int64_t)sig * A.a_stride, *cpv = cp_ws + (int64_t)sig * A.a_stride;
    const int last = min(s + kNakRun - 1, m - 2);
    double cp = 0.0, dp = 0.0;
    for (int j = max(1, s - kNakWarm); j <= last; ++j) {
        NakRow r;
        if (j == 1 || j == m - 2) r = nak_row(e, S, m, j);       // (the not-a-knot rows: their own diag / sup)
        else { r.sub = subv[j]; r.rhs = rhsv[j]; r.sup = 1 - r.sub; r.diag = 2; }
        nak_step(r, cp, dp);
        if (j >= s) { cpv[j] = cp; dpv[j] = dp; }
    }
}

// back substitution M_j = dp_j - cp_j M_{j+1} of a run per thread, the two end moments from the not-a-knot conditions
__global__ __launch_bounds__(64) void k_nak_backward(CubicArgs A, const double *__restrict__ cp_ws)
{
    const int sig = blockIdx.y;
    const CubicJob job = A.jobs[(size_t)sig * A.job_stride];
    if (!job.valid) return;
    const int m = job.idx + 1;
    const int s = 1 + (blockIdx.x * blockDim.x + threadIdx.x) * kNakRun;
    if (s > m - 2) return;
    const int32_t *e = A.e + (int64_t)sig * A.e_stride;
    const double *dpv = A.bf + (int64_t)sig * A.a_stride, *cpv = cp_ws + (int64_t)sig * A.a_stride;
    double *M = A.b + (int64_t)sig * A.a_stride;
    const int top = min(s + kNakRun - 1, m - 2);
    double y = 0.0, y_next = 0.0;                      // M_{j+1}, M_{j+2} while walking down
    for (int j = min(m - 2, top + kNakWarm); j >= s; --j) {
        y_next = y;
        y = dpv[j] - cpv[j] * y;
        if (j <= top) M[j] = y;
    }
    if (s == 1) {                                      // M_0 from M_1 (= y) and M_2 (= y_next: the loop ran at least twice, m >= 4)
        const double q = (double)(e[1] - e[0]) / (double)(e[2] - e[1]);
        M[0] = (1 + q) * y - q * y_next;
    }
    if (top == m - 2) {                                // M_{m-1} from M_{m-2} and M_{m-3}
        const double q = (double)(e[m - 1] - e[m - 2]) / (double)(e[m - 2] - e[m - 3]);
        const double mt = dpv[m - 2];                  // cp_{m-2} = 0: M_{m-2} = dp_{m-2}
        const double mb = dpv[m - 3] - cpv[m - 3] * mt;
        M[m - 1] = (1 + q) * mt - q * mb;
    }
}

// ---- ONE signal of up to kNakSmallMax samples: the whole operator in ONE launch, one workgroup ---------------------------------
// MEITD / XITD (MEITD.py:344-549) call the operator on one short signal after the other, every call waiting for the host's
// decision on the previous one: as the batch form's nine launches (knots, compaction, jobs, values, rows, forward, backward,
// evaluation, + the produced baseline's knot count) an extraction of a 3000-sample signal took ~130 us, all of it launch boundaries.
// Here the same phases — the same expressions in the same order, hence the same bits as the launches above — run behind one another
// in one workgroup with a barrier in between (the arrays are global memory, a few ten KB: L2): knots + ordered compaction (a thread
// owns a run of consecutive samples, one block scan), knot values, rows, the forward and backward runs, evaluation, and the
// knot count of the baseline it has just written.
// LDS = true: the rows of the system and the forward sweep's two arrays (4 x (n + 2) doubles) live in dynamic LDS — the two serial
// recurrences (160 dependent steps per thread) then wait for LDS, not for L2, in every step: 104 -> ~45 us per extraction of a
// 3000-sample signal; signals whose worst case does not fit (n > ~4900) keep them in global memory.
constexpr int kNakSmallMax = 8192, kNakSmallThreads = 1024, kNakSmallPer = kNakSmallMax / kNakSmallThreads;
constexpr size_t kNakSmallLdsMax = 156 * 1024;
// The operator's body, for the kNakSmallThreads threads of one workgroup (k_nak_small below; k_meitd_small, itd_meitd.hpp, runs it
// once per extraction of its loop).  s_dyn: 4 x (n + 2) doubles of LDS when LDS.  Comes back with res = {knots, NaN seen, valid,
// knots of the produced baseline (want_bcount)} in every thread.  The caller puts a barrier between two uses.
template <bool LDS>
__device__ __forceinline__ void nak_small_body(const double *__restrict__ x, int n, int min_extrema, int32_t *__restrict__ e,
                                               double *__restrict__ K, double *__restrict__ dpg, double *__restrict__ M,
                                               double *__restrict__ cpg, double *__restrict__ subg, double *__restrict__ rhsg,
                                               double *__restrict__ base, double *__restrict__ rot, int want_bcount,
                                               double *__restrict__ s_dyn, int (&res)[4], long long *prof = nullptr)
{
    double *const subv = LDS ? s_dyn : subg, *const rhsv = LDS ? s_dyn + (n + 2) : rhsg;
    double *const cpv = LDS ? s_dyn + 2 * (n + 2) : cpg, *const dpv = LDS ? s_dyn + 3 * (n + 2) : dpg;
    __shared__ int s_wave[kNakSmallThreads / 64], s_flag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + kNakSmallThreads - 1) / kNakSmallThreads;          // consecutive samples a thread owns (<= kNakSmallPer)
    const int lo = tid * per, hi = min(n, lo + per);
    if (tid < 4) s_flag[tid] = 0;
    __syncthreads();
    // the knots of a row held in global memory: ITD.py:59 on x and on -x (raw differences), never the first or the last sample
    auto knot_mask = [&](const double *__restrict__ v, bool &has_nan) {
        unsigned mask = 0;
        for (int i = lo; i < hi; ++i) {
            const double x0 = v[i];
            has_nan = has_nan || x0 != x0;
            if (i >= 1 && i <= n - 2) {
                const double vil = v[i + 1] - x0, vix = x0 - v[i - 1];
                if (((vil > 0.0) && (vix <= 0.0)) || ((vil < 0.0) && (vix >= 0.0))) mask |= 1u << (i - lo);
            }
        }
        return mask;
    };
    // exclusive prefix of c over the workgroup's threads and the total
    auto block_scan = [&](int c, int &total) {
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        __syncthreads();
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int wb = 0, tot = 0;
        for (int k = 0; k < kNakSmallThreads / 64; ++k) { const int v = s_wave[k]; if (k < wave) wb += v; tot += v; }
        total = tot;
        return wb + inc - c;
    };
    long long tp = prof ? (long long)wall_clock64() : 0;
    auto mark = [&](int k) { if (prof) { const long long t = (long long)wall_clock64(); prof[k] += t - tp; tp = t; } };
    bool has_nan = false;
    const unsigned mask = knot_mask(x, has_nan);
    if (has_nan) s_flag[0] = 1;
    int knots;
    const int off = block_scan(__popc(mask), knots);
    {
        int k = 1 + off;
        for (unsigned mm = mask; mm; mm &= mm - 1) e[k++] = lo + __builtin_ctz(mm);
        if (tid == 0) { e[0] = 0; e[knots + 1] = n - 1; }
    }
    __syncthreads();                                      // (also: s_flag[0] is complete)
    const int m = knots + 2;
    const bool valid = knots >= min_extrema && knots >= 2;
    res[0] = knots; res[1] = s_flag[0]; res[2] = valid ? 1 : 0; res[3] = 0;
    if (!valid) {                                         // fewer knots than the operator needs: the signal is its own baseline
        for (int i = lo; i < hi; ++i) { base[i] = x[i]; if (rot) rot[i] = 0.0; }
        if (want_bcount) res[3] = knots;                  // (the baseline IS the signal)
        return;
    }
    mark(0);
    for (int k = tid; k < m; k += kNakSmallThreads) K[k] = spline_knot_value(x, n, e, m, k);
    __syncthreads();
    mark(1);
    for (int j = 1 + tid; j <= m - 2; j += kNakSmallThreads) { const NakRow r = nak_row(e, K, m, j); subv[j] = r.sub; rhsv[j] = r.rhs; }
    __syncthreads();
    mark(2);
    // The two sweeps: a thread's run is a chain of kNakWarm + kNakRun dependent steps (one reciprocal's latency each going up, one
    // multiply-add's going down) — the whole time of the operator for a short signal.  The rows of the next eight steps are therefore
    // requested before the current eight are chained (with one step's row fetched per step the sweeps waited for LDS every step:
    // 26 + 15 us per extraction of a 3000-sample signal), and the two not-a-knot rows are taken out of the loop.
    for (int s = 1 + tid * kNakRun; s <= m - 2; s += kNakSmallThreads * kNakRun) {          // k_nak_forward's run
        const int last = min(s + kNakRun - 1, m - 2), j0 = max(1, s - kNakWarm);
        const int plain_end = last == m - 2 ? last - 1 : last;                               // (>= 1: m >= 4)
        double cp = 0.0, dp = 0.0;
        int j = j0;
        if (j == 1) {
            const NakRow r = nak_row(e, K, m, 1);
            nak_step(r, cp, dp);
            if (s == 1) { cpv[1] = cp; dpv[1] = dp; }
            ++j;
        }
        // plain rows j .. plain_end: the warm-up (nothing stored) up to s - 1, then the run
        auto chain = [&](int from, int to, bool store) {
            int jj0 = from;
            if (jj0 > to) return;
            double sb[8], rb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { sb[q] = subv[min(jj0 + q, to)]; rb[q] = rhsv[min(jj0 + q, to)]; }
            for (; jj0 + 8 <= to + 1; jj0 += 8) {                    // whole blocks: no test inside
                double sn[8], rn[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { sn[q] = subv[min(jj0 + 8 + q, to)]; rn[q] = rhsv[min(jj0 + 8 + q, to)]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    NakRow r;
                    r.sub = sb[q]; r.rhs = rb[q]; r.sup = 1 - r.sub; r.diag = 2;
                    nak_step(r, cp, dp);
                    if (store) { cpv[jj0 + q] = cp; dpv[jj0 + q] = dp; }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { sb[q] = sn[q]; rb[q] = rn[q]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {                             // the rest of a block
                if (jj0 + q <= to) {
                    NakRow r;
                    r.sub = sb[q]; r.rhs = rb[q]; r.sup = 1 - r.sub; r.diag = 2;
                    nak_step(r, cp, dp);
                    if (store) { cpv[jj0 + q] = cp; dpv[jj0 + q] = dp; }
                }
            }
        };
        chain(j, min(s - 1, plain_end), false);
        chain(max(j, s), plain_end, true);
        if (last == m - 2) {
            const NakRow r = nak_row(e, K, m, m - 2);
            nak_step(r, cp, dp);
            cpv[m - 2] = cp; dpv[m - 2] = dp;
        }
    }
    __syncthreads();
    mark(3);
    for (int s = 1 + tid * kNakRun; s <= m - 2; s += kNakSmallThreads * kNakRun) {          // k_nak_backward's run
        const int top = min(s + kNakRun - 1, m - 2), jt = min(m - 2, top + kNakWarm);
        double y = 0.0, y_next = 0.0;
        // rows jt .. s going down: the warm-up (nothing stored) down to top + 1, then the run
        auto chain = [&](int from, int to, bool store) {
            int jj0 = from;
            if (jj0 < to) return;
            double db[8], cb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { db[q] = dpv[max(jj0 - q, to)]; cb[q] = cpv[max(jj0 - q, to)]; }
            for (; jj0 - 8 >= to - 1; jj0 -= 8) {
                double dn[8], cn[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { dn[q] = dpv[max(jj0 - 8 - q, to)]; cn[q] = cpv[max(jj0 - 8 - q, to)]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    y_next = y;
                    y = db[q] - cb[q] * y;
                    if (store) M[jj0 - q] = y;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { db[q] = dn[q]; cb[q] = cn[q]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (jj0 - q >= to) {
                    y_next = y;
                    y = db[q] - cb[q] * y;
                    if (store) M[jj0 - q] = y;
                }
            }
        };
        chain(jt, top + 1, false);
        chain(top, s, true);
        if (s == 1) {
            const double q = (double)(e[1] - e[0]) / (double)(e[2] - e[1]);
            M[0] = (1 + q) * y - q * y_next;
        }
        if (top == m - 2) {
            const double q = (double)(e[m - 1] - e[m - 2]) / (double)(e[m - 2] - e[m - 3]);
            const double mt = dpv[m - 2];
            const double mb = dpv[m - 3] - cpv[m - 3] * mt;
            M[m - 1] = (1 + q) * mt - q * mb;
        }
    }
    __syncthreads();
    mark(4);
    {   // evaluation (k_cubic_eval<NAK>): the segment of a sample = the knots at or before it
        int j = off;                                      // knots in front of the thread's run
        for (int i = lo; i < hi; ++i) {
            j += (mask >> (i - lo)) & 1u;
            const int32_t ej = e[j], en = e[j + 1];
            const double Kj = K[j], Kn = K[j + 1], bj = M[j], bn = M[j + 1];
            const double hj = (double)(en - ej);
            const double t = (double)(i - ej) * sweep_rcp(hj);
            const double omt = 1 - t;
            const double h26 = hj * hj * (1.0 / 6);
            const double c1 = h26 * ((omt * omt * omt - 1) + t) * bj;
            const double c2 = h26 * (t * t * t - t) * bn;
            const double v = ((omt * Kj + t * Kn) + c1) + c2;
            base[i] = v;
            if (rot) rot[i] = x[i] - v;
        }
    }
    mark(5);
    if (want_bcount) {                                    // MEITD.py:362-363, :497-505: the extrema count of the produced baseline
        __syncthreads();
        bool nn = false;
        int total;
        (void)block_scan(__popc(knot_mask(base, nn)), total);
        res[3] = total;
    }
}

template <bool LDS>
__global__ __launch_bounds__(kNakSmallThreads) void k_nak_small(const double *__restrict__ x, int n, int min_extrema, int32_t *__restrict__ e,
                                                               double *__restrict__ K, double *__restrict__ dpg, double *__restrict__ M,
                                                               double *__restrict__ cpg, double *__restrict__ subg, double *__restrict__ rhsg,
                                                               double *__restrict__ base, double *__restrict__ rot, int want_bcount,
                                                               int32_t *__restrict__ out /* knots, NaN flag, valid, knots of the produced baseline */,
                                                               unsigned long long *__restrict__ done_words = nullptr, int32_t done_seq = 0
                                                               /* done_words (host-mapped): the same four values as self-validating words (small_put) — the host polls them */)
{
    extern __shared__ double s_dyn[];
    int res[4];
    nak_small_body<LDS>(x, n, min_extrema, e, K, dpg, M, cpg, subg, rhsg, base, rot, want_bcount, s_dyn, res);
    if (threadIdx.x == 0) {
        if (done_words) { for (int q = 0; q < 4; ++q) small_put(done_words, q, (uint32_t)res[q], (uint32_t)done_seq); }
        else { out[0] = res[0]; out[1] = res[1]; out[2] = res[2]; out[3] = res[3]; }
    }
}

}  // namespace itd
