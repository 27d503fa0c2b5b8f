// itd_meitd.hpp — MEITD's selection loop (MEITD.py:395-534, with retrieve_proper_rotation :344-368 and
// determine_if_first_is_proper_rotation :371-392) for ONE signal of up to kNakSmallMax samples as ONE launch of one workgroup.
//
// The loop is control logic around three operators — the spline baseline extraction (MEITD.py:303-338: nak_small_body, itd_nak.hpp),
// the extrema count and the weighted permutation entropy (MEITD.py:79-128: wpe3_pass, itd_wpe.hpp) — and every turn of it waits for
// the decision drawn from the previous operator's scalars.  Driven from the host (pyitd_amd/meitd.py, one launch per operator, the
// scalars through mapped host words) a call on a 3000-sample signal is ~107 launch round trips of 26-60 us: 5.4 ms, of which the
// operators' own work is less than half.  Here the decisions are drawn where the scalars are: the workgroup runs the operators
// behind one another on rows of one device allocation (L2-resident), thread 0 draws the entropy and every thread follows the same
// branch; the host sees one launch.
//
// What keeps the selections the reference's: counts are integers; the extraction is the launch form's, bit for bit; the entropy's
// six sums are the launch form's, bit for bit; only the six logarithms differ in who takes them (the device's log2 here, numpy's
// there).  Every probe's sums, counts and entropy are therefore LOGGED (MeitdProbe), and the caller re-draws each
// entropy with numpy and compares the threshold test — the only thing an entropy is ever used for (`_proper`, MEITD.py:364 / :387);
// one differing test and the call is repeated by the host-driven loop.  The same fallback serves everything this kernel does not
// model: a NaN in any row, an extraction with fewer than two knots (scipy raises there), a full log.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "itd_nak.hpp"
#include "itd_wpe.hpp"

#pragma clang fp contract(off)

namespace itd {

constexpr int kMeitdWork = 6, kMeitdKept = 22;          // rows: 6 working rows, then the kept "high" rows, then the kept "low" rows
constexpr int kMeitdLogCap = 1024, kMeitdMaxSteps = 1 << 16;
constexpr int kMeitdOk = 0, kMeitdTooFew = 1 /* fewer than 4 extrema: MEITD.py:411-413 */, kMeitdNaN = 2, kMeitdFewKnots = 3, kMeitdLogFull = 4;
struct MeitdProbe { double w[6]; int32_t c[6]; int32_t count, pad; double wpe; };   // 88 bytes
struct MeitdOut { int32_t status, n_high, n_low, x_row, probes, extractions, steps, pad; int32_t ticks[16]; /* 10 ns units per operator kind: probe, extract, count, copies */ };
constexpr size_t kMeitdWpeLds = (2 * (size_t)kWpeChunk + 2) * sizeof(double);

// what the operators share: the rows, the extraction's workspace, the log, the loop's counters — one object in LDS.  The operators
// are real functions (inlined at their twenty call sites they left the kernel with 250 spilled vector registers and every phase of
// every operator ran 1.5-3 times slower); what the loop branches on comes back in registers.
struct MeitdCtx {
    double *rows;
    int n;
    double wpemax;
    int32_t *e;
    double *K, *dpg, *M, *cpg, *subg, *rhsg;
    double *s_dyn;
    MeitdProbe *log;
    int probes, extractions;       // (written by thread 0, read behind a barrier)
    long long t_op[16];            // thread 0's wall clock per operator kind and phase: a diagnostic in the result
    __device__ __forceinline__ double *row(int r) const { return rows + (size_t)r * (size_t)n; }
};
struct MeitdProbed { double wpe; int count, status; };

// (normalised entropy, extrema count) of a row: MEITD.py:346-351 / :373-378
__device__ __noinline__ MeitdProbed meitd_probe(MeitdCtx &c, int r)
{
    constexpr int NT = kNakSmallThreads;
    __shared__ double s_w[6], s_wpe;
    __shared__ long long s_c[6];
    __shared__ int s_k[2];
    const int tid = threadIdx.x;
    const long long t0 = (long long)wall_clock64();
    __syncthreads();
    if (tid < 2) s_k[tid] = 0;
    double s;
    long long cw;
    int kn, nanf;
    long long t_sum = 0;
    const int probes = c.probes;
    wpe3_pass<NT>(c.row(r), 0, (int64_t)c.n - 2, c.s_dyn, c.s_dyn + kWpeChunk + 2, s, cw, kn, nanf, &t_sum);
    const long long t1 = (long long)wall_clock64();
    if (tid < 6) { s_w[tid] = s; s_c[tid] = cw; }
    if (kn) atomicAdd(&s_k[0], kn);
    if (nanf) s_k[1] = 1;
    __syncthreads();
    if (tid < 64) {
        // MEITD.py:119-128 on the patterns that occur, in numpy's order of operations (sums of fewer than 8 items: left to right);
        // lane q takes pattern q's logarithm, lane 0 adds the terms up in order
        double tot = 0.0;
        bool first = true;
        for (int q = 0; q < 6; ++q) if (s_c[q] > 0) { tot = first ? s_w[q] : tot + s_w[q]; first = false; }
        const int present = tid < 6 && s_c[tid < 6 ? tid : 0] > 0 ? 1 : 0;
        double t = 0.0;
        if (present) { const double p = s_w[tid] / tot; t = p * log2(p); }
        double acc = 0.0;
        first = true;
        for (int q = 0; q < 6; ++q) {
            const double tq = __shfl(t, q);
            if (__shfl(present, q)) { acc = first ? tq : acc + tq; first = false; }
        }
        const double pe = -acc / 2.584962500721156;          // numpy.log2(factorial(3))
        if (tid == 0) {
            s_wpe = pe;
            if (probes < kMeitdLogCap) {
                MeitdProbe &L = c.log[probes];
                for (int q = 0; q < 6; ++q) { L.w[q] = s_w[q]; L.c[q] = (int32_t)s_c[q]; }
                L.count = s_k[0]; L.pad = 0; L.wpe = pe;
            }
        }
    }
    __syncthreads();
    MeitdProbed out;
    out.wpe = s_wpe;
    out.count = s_k[0];
    out.status = s_k[1] ? kMeitdNaN : probes >= kMeitdLogCap ? kMeitdLogFull : kMeitdOk;
    if (tid == 0) {
        const long long t2 = (long long)wall_clock64();
        c.probes = probes + 1;
        c.t_op[4] += t_sum; c.t_op[5] += t2 - t1; c.t_op[0] += t2 - t0;
    }
    return out;
}

// matlab_detect_peaks(v).size + matlab_detect_peaks(-v).size: ITD.py:59 on v and on -v, raw differences, samples 1 .. n-2
__device__ __noinline__ int meitd_count(MeitdCtx &c, int r)
{
    constexpr int NT = kNakSmallThreads;
    __shared__ int s_k[2];
    const int tid = threadIdx.x, n = c.n;
    const long long t0 = (long long)wall_clock64();
    __syncthreads();
    if (tid < 2) s_k[tid] = 0;
    __syncthreads();
    const double *__restrict__ v = c.row(r);
    int kn = 0, nanf = 0;
    for (int i = tid; i < n; i += NT) {
        const double x0 = v[i];
        nanf |= x0 != x0 ? 1 : 0;
        if (i >= 1 && i <= n - 2) {
            const double vil = v[i + 1] - x0, vix = x0 - v[i - 1];
            kn += (((vil > 0.0) && (vix <= 0.0)) || ((vil < 0.0) && (vix >= 0.0))) ? 1 : 0;
        }
    }
    if (kn) atomicAdd(&s_k[0], kn);
    if (nanf) s_k[1] = 1;
    __syncthreads();
    if (tid == 0) c.t_op[2] += (long long)wall_clock64() - t0;
    return s_k[1] ? -kMeitdNaN : s_k[0];                      // (a failure comes back as minus its status)
}

// itd_baseline_extract (MEITD.py:303-338): src -> (rot, base; rot < 0: not wanted); returns the knot count of the produced baseline if asked for
template <bool LDS>
__device__ __noinline__ int meitd_extract(MeitdCtx &c, int src, int base, int rot, int want_bcount)
{
    const long long t0 = (long long)wall_clock64();
    __syncthreads();
    int res[4];
    long long t_ph[8] = {};
    nak_small_body<LDS>(c.row(src), c.n, 0, c.e, c.K, c.dpg, c.M, c.cpg, c.subg, c.rhsg, c.row(base), rot >= 0 ? c.row(rot) : nullptr, want_bcount,
                        c.s_dyn, res, t_ph);
    __syncthreads();
    if (threadIdx.x == 0) {
        ++c.extractions;
        for (int q = 0; q < 6; ++q) c.t_op[8 + q] += t_ph[q];
        c.t_op[1] += (long long)wall_clock64() - t0;
    }
    if (res[1]) return -kMeitdNaN;
    if (res[0] < 2) return -kMeitdFewKnots;                  // "m > k must hold": what scipy.interpolate.splrep raises
    return res[3];
}

// dst = a - b (b < 0: dst = a; a < 0: dst = 0): MEITD.py:453 and the rows' assignments
__device__ __noinline__ void meitd_rowop(MeitdCtx &c, int dst, int a, int b)
{
    constexpr int NT = kNakSmallThreads;
    const long long t0 = (long long)wall_clock64();
    __syncthreads();
    double *d = c.row(dst);
    const double *pa = a >= 0 ? c.row(a) : nullptr, *pb = b >= 0 ? c.row(b) : nullptr;
    for (int i = threadIdx.x; i < c.n; i += NT) d[i] = pa ? (pb ? pa[i] - pb[i] : pa[i]) : 0.0;
    if (threadIdx.x == 0) c.t_op[3] += (long long)wall_clock64() - t0;
}

template <bool LDS>
__global__ __launch_bounds__(kNakSmallThreads) void k_meitd_small(double *__restrict__ rows, int n, double wpemax, int32_t *__restrict__ e,
                                                                 double *__restrict__ K, double *__restrict__ dpg, double *__restrict__ M,
                                                                 double *__restrict__ cpg, double *__restrict__ subg, double *__restrict__ rhsg,
                                                                 MeitdProbe *__restrict__ log, MeitdOut *__restrict__ out /* both device memory: the host copies them
                                                                                                                          behind the launch (milliseconds long) */)
{
    extern __shared__ double s_dyn[];
    __shared__ MeitdCtx c;
    if (threadIdx.x == 0) {
        c.rows = rows; c.n = n; c.wpemax = wpemax; c.e = e; c.K = K; c.dpg = dpg; c.M = M; c.cpg = cpg; c.subg = subg; c.rhsg = rhsg;
        c.s_dyn = s_dyn; c.log = log; c.probes = 0; c.extractions = 0;
        for (int q = 0; q < 16; ++q) c.t_op[q] = 0;
    }
    __syncthreads();
    // the free working rows, four bits each (meitd.py's list: pop() takes the last)
    unsigned free_rows = 0x543210u;
    int nfree = kMeitdWork;
    int status = kMeitdOk;
    auto take = [&]() { --nfree; return (int)((free_rows >> (4 * nfree)) & 15u); };
    auto give = [&](int r) { free_rows = (free_rows & ~(15u << (4 * nfree))) | ((unsigned)r << (4 * nfree)); ++nfree; };
    auto proper_of = [&](double wpe) { return (wpe < wpemax && !(wpe < 0.2)) ? 1 : 0; };     // MEITD.py:364 / :387
    auto probe = [&](int r, double &wpe, int &count) {
        const MeitdProbed p = meitd_probe(c, r);
        wpe = p.wpe; count = p.count;
        if (p.status) status = p.status;
    };
    auto count_knots = [&](int r) { const int k = meitd_count(c, r); if (k < 0) status = -k; return k < 0 ? 0 : k; };
    auto extract = [&](int src, int base, int rot, bool want_bcount) {
        const int k = meitd_extract<LDS>(c, src, base, rot, want_bcount ? 1 : 0);
        if (k < 0) status = -k;
        return k < 0 ? 0 : k;
    };
    auto assign = [&](int dst, int src) { meitd_rowop(c, dst, src, -1); };
    auto zero = [&](int dst) { meitd_rowop(c, dst, -1, -1); };
    auto subtract_into = [&](int a, int b) { meitd_rowop(c, a, a, b); };                     // a = a - b (MEITD.py:453)
    // determine_if_first_is_proper_rotation(src) into the rows rot / base (base < 0: not wanted); (wpe, cnt) = probe(src)
    auto determine = [&](int src, int rot, int base, double wpe, int cnt) {
        if (cnt < 5) {
            assign(rot, src);
            if (base >= 0) zero(base);
            return 0;
        }
        const int tmp = base >= 0 ? base : take();
        (void)extract(src, tmp, rot, false);
        if (base < 0) give(tmp);
        return proper_of(wpe);
    };

    int x = take(), rotation = take(), baseline = take();         // the signal is in row 5 (meitd.py uploads it there)
    int n_high = 0, n_low = 0, steps = 0;
    double wpe;
    int cnt;
    probe(x, wpe, cnt);
    int proper = status ? 0 : determine(x, rotation, baseline, wpe, cnt);
    bool changed = false, on_signal = true;
    int digs = 1, count = cnt;
    if (!status && count < 4) status = kMeitdTooFew;
    while (!status && count > 5) {
        if (++steps > kMeitdMaxSteps) { status = kMeitdLogFull; break; }
        if (n_high + n_low > 20) break;
        if (proper == 0) {                                        // retrieve_proper_rotation on the row `rotation`
            probe(rotation, wpe, cnt);
            if (status) break;
            if (cnt > 5 && proper_of(wpe)) {
                const int base = take(), res = take();
                (void)extract(rotation, base, res, false);
                give(base);
                give(rotation);
                rotation = res;
                proper = 1;
            }
            if (status) break;
        }
        if (proper == 1) {
            if (on_signal) assign(kMeitdWork + n_high++, rotation);
            else assign(kMeitdWork + kMeitdKept + n_low++, rotation);
            digs = 0;
            subtract_into(x, rotation);
            changed = true;
        }
        if (changed && on_signal) {
            count = count_knots(x);
            if (status || count < 5) continue;
            (void)extract(x, baseline, -1, false);
            if (status) break;
            probe(baseline, wpe, cnt);
            if (status) break;
            proper = determine(baseline, rotation, -1, wpe, cnt);
            changed = false; on_signal = false;
            continue;
        } else if (on_signal) {
            probe(baseline, wpe, cnt);
            if (status) break;
            proper = determine(baseline, rotation, -1, wpe, cnt);
            on_signal = false;
            continue;
        }
        if (changed) {                                            // (and not on the signal)
            probe(x, wpe, cnt);
            count = cnt;
            if (status || count < 5) continue;
            proper = determine(x, rotation, baseline, wpe, cnt);
            changed = false; on_signal = true;
            continue;
        }
        if (digs == 0) {
            (void)extract(x, baseline, rotation, false);
            if (status) break;
            digs = 1;
        }
        count = count_knots(baseline);
        if (status || count < 5) continue;
        for (int d = 0; d < digs; ++d) {
            const int deeper = take();
            count = extract(baseline, deeper, rotation, true);
            give(baseline);
            baseline = deeper;
            if (status || count < 5) break;
        }
        ++digs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out->status = status; out->n_high = n_high; out->n_low = n_low; out->x_row = x;
        out->probes = c.probes; out->extractions = c.extractions; out->steps = steps; out->pad = 0;
        for (int q = 0; q < 16; ++q) out->ticks[q] = (int32_t)c.t_op[q];
    }
}

}  // namespace itd
