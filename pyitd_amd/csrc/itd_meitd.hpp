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
// there).  Every probe's sums, counts and entropy are therefore LOGGED (MeitdProbe, host-mapped), and the caller re-draws each
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
struct MeitdOut { int32_t status, n_high, n_low, x_row, probes, extractions, steps, pad; };
constexpr size_t kMeitdWpeLds = (2 * (size_t)kWpeChunk + 2) * sizeof(double);

template <bool LDS>
__global__ __launch_bounds__(kNakSmallThreads) void k_meitd_small(double *__restrict__ rows, int n, double wpemax, int32_t *__restrict__ e,
                                                                 double *__restrict__ K, double *__restrict__ dpg, double *__restrict__ M,
                                                                 double *__restrict__ cpg, double *__restrict__ subg, double *__restrict__ rhsg,
                                                                 MeitdProbe *__restrict__ log, MeitdOut *__restrict__ out,
                                                                 int32_t *__restrict__ done_flag, int32_t done_seq)
{
    extern __shared__ double s_dyn[];
    __shared__ double s_w[6], s_wpe;
    __shared__ long long s_c[6];
    __shared__ int s_k[2];
    constexpr int NT = kNakSmallThreads;
    const int tid = threadIdx.x;
    auto row = [&](int r) { return rows + (size_t)r * (size_t)n; };
    // the free working rows, four bits each (meitd.py's list: pop() takes the last)
    unsigned free_rows = 0x543210u;
    int nfree = kMeitdWork;
    auto take = [&]() { --nfree; return (int)((free_rows >> (4 * nfree)) & 15u); };
    auto give = [&](int r) { free_rows = (free_rows & ~(15u << (4 * nfree))) | ((unsigned)r << (4 * nfree)); ++nfree; };
    int status = kMeitdOk, probes = 0, extractions = 0;

    auto proper_of = [&](double wpe) { return (wpe < wpemax && !(wpe < 0.2)) ? 1 : 0; };     // MEITD.py:364 / :387
    // (normalised entropy, extrema count) of a row: MEITD.py:346-351 / :373-378
    auto probe = [&](int r, double &wpe, int &count) {
        __syncthreads();
        if (tid < 2) s_k[tid] = 0;
        double s;
        long long c;
        int kn, nanf;
        wpe3_pass<NT>(row(r), 0, (int64_t)n - 2, s_dyn, s_dyn + kWpeChunk + 2, s, c, kn, nanf);
        if (tid < 6) { s_w[tid] = s; s_c[tid] = c; }
        if (kn) atomicAdd(&s_k[0], kn);
        if (nanf) s_k[1] = 1;
        __syncthreads();
        if (tid == 0) {
            // MEITD.py:119-128 on the patterns that occur, in numpy's order of operations (sums of fewer than 8 items: left to right)
            double tot = 0.0, acc = 0.0;
            bool first = true;
            for (int q = 0; q < 6; ++q) if (s_c[q] > 0) { tot = first ? s_w[q] : tot + s_w[q]; first = false; }
            first = true;
            for (int q = 0; q < 6; ++q)
                if (s_c[q] > 0) {
                    const double p = s_w[q] / tot, t = p * log2(p);
                    acc = first ? t : acc + t;
                    first = false;
                }
            const double pe = -acc / 2.584962500721156;          // numpy.log2(factorial(3))
            s_wpe = pe;
            if (probes < kMeitdLogCap) {
                MeitdProbe &L = log[probes];
                for (int q = 0; q < 6; ++q) { L.w[q] = s_w[q]; L.c[q] = (int32_t)s_c[q]; }
                L.count = s_k[0]; L.pad = 0; L.wpe = pe;
            }
        }
        __syncthreads();
        wpe = s_wpe;
        count = s_k[0];
        if (s_k[1]) status = kMeitdNaN;
        if (probes >= kMeitdLogCap) status = kMeitdLogFull;
        ++probes;
    };
    // matlab_detect_peaks(v).size + matlab_detect_peaks(-v).size: ITD.py:59 on v and on -v, raw differences, samples 1 .. n-2
    auto count_knots = [&](int r) {
        __syncthreads();
        if (tid < 2) s_k[tid] = 0;
        __syncthreads();
        const double *__restrict__ v = row(r);
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
        if (s_k[1]) status = kMeitdNaN;
        return s_k[0];
    };
    // itd_baseline_extract (MEITD.py:303-338): src -> (rot, base); returns the knot count of the produced baseline if asked for
    auto extract = [&](int src, int base, int rot, bool want_bcount) {
        __syncthreads();
        int res[4];
        nak_small_body<LDS>(row(src), n, 0, e, K, dpg, M, cpg, subg, rhsg, row(base), rot >= 0 ? row(rot) : nullptr, want_bcount ? 1 : 0, s_dyn, res);
        ++extractions;
        if (res[1]) status = kMeitdNaN;
        else if (res[0] < 2) status = kMeitdFewKnots;            // "m > k must hold": what scipy.interpolate.splrep raises
        return res[3];
    };
    auto assign = [&](int dst, int src) {
        __syncthreads();
        const double *__restrict__ a = row(src);
        double *__restrict__ d = row(dst);
        for (int i = tid; i < n; i += NT) d[i] = a[i];
    };
    auto zero = [&](int dst) {
        __syncthreads();
        double *__restrict__ d = row(dst);
        for (int i = tid; i < n; i += NT) d[i] = 0.0;
    };
    auto subtract_into = [&](int a, int b) {                      // a = a - b (MEITD.py:453)
        __syncthreads();
        double *__restrict__ pa = row(a);
        const double *__restrict__ pb = row(b);
        for (int i = tid; i < n; i += NT) pa[i] = pa[i] - pb[i];
    };
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
    if (tid == 0) {
        out->status = status; out->n_high = n_high; out->n_low = n_low; out->x_row = x;
        out->probes = probes; out->extractions = extractions; out->steps = steps; out->pad = 0;
        if (done_flag) { __threadfence_system(); *done_flag = done_seq; }
    }
}

}  // namespace itd
