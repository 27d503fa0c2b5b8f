// itd_fitpack.hpp — the smoothing cubic B-spline the reference's FITPACK flavour of the baseline is built on.
//
// Reference call sites: custom_splrep -> scipy.interpolate.splrep(x, y, k=3)   numba_accelerated_itd.py:69-86 (:84),
// MEITD.py:136-152, siftED2D.ipynb cell 1; evaluation by numba_splev (numba_accelerated_itd.py:89-164).
// The arithmetic lives in a third-party dependency that is NOT in the reference tree: scipy's FITPACK (P. Dierckx, "Curve
// and Surface Fitting with Splines", 1993; routines curfit / fpcurf / fpknot / fpdisc / fpbspl / fpgivs / fprota / fpback /
// fprati; SciPy 1.15.3 in this image, unpinned upstream: environment.yml:1-6).  This file restates that published algorithm
// for the call the reference makes: iopt = 0, k = 3, unit weights, xb = x[0], xe = x[m-1], nest = m + k + 1, tol = 0.001,
// maxit = 20, and s = 0 — splrep's default when no weights are given (s = m - sqrt(2 m) only applies with weights): the
// INTERPOLATING not-a-knot cubic spline, whose coefficients FITPACK obtains by rotating the m x m banded collocation matrix
// into triangular form row by row (Givens) and back substitution.  The s > 0 branches (knot insertion, smoothing parameter)
// are restated as well and held to scipy with explicit s in tests/test_fitpack_host.py.
//   part 1  knots: start from the least-squares cubic polynomial; while f(p = inf) - s > acc add knots at data sites in the
//           intervals of largest residual (fpknot), refit by Givens rotations of the banded observation matrix;
//   part 2  smoothing parameter p with F(p) = s by rational interpolation (fprati) over the augmented least-squares
//           problem [A ; B / p] (B = jumps of the third derivative at the interior knots, fpdisc).
// One instance runs sequentially (it is a serial algorithm: every step depends on the previous one); the GPU runs one
// instance per signal (k_spline_fit in itd_engine.hip: one thread per signal, thousands of signals per launch).
// The functions compile for the host too (ITD_HD): tests/test_fitpack_host.py builds them with g++ and holds t and c to
// scipy's own splrep on this image — a check of the restatement, not a fallback: nothing in pyitd_amd/ calls the host build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ITD_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#define ITD_HD static inline
#endif

namespace itd_fp {

constexpr int K = 3, K1 = 4, K2 = 5;

// workspace of one fit, carved out of a caller buffer (doubles); all arrays are used 1-based like the FORTRAN original
struct Work {
    double *t, *c, *fpint, *z, *a, *b, *g, *q;   // t,c,fpint,z: [nest+1]; a: [4][nest+1]; b,g: [5][nest+1]; q: [4][m+1]
    int32_t *nrdata;                              // [nest+1]
    int nest, m;
};
ITD_HD int64_t work_doubles(int m) { const int nest = m + K1; return (int64_t)(nest + 1) * (4 + 4 + 5 + 5) + (int64_t)(m + 1) * 4 + (nest + 2) / 2 + 1; }
ITD_HD Work work_carve(double *buf, int m)
{
    Work w;
    w.m = m;
    w.nest = m + K1;
    const int L = w.nest + 1;
    w.t = buf; buf += L;
    w.c = buf; buf += L;
    w.fpint = buf; buf += L;
    w.z = buf; buf += L;
    w.a = buf; buf += 4 * L;
    w.b = buf; buf += 5 * L;
    w.g = buf; buf += 5 * L;
    w.q = buf; buf += 4 * (m + 1);
    w.nrdata = reinterpret_cast<int32_t *>(buf);
    return w;
}
#define A_(i, j) w.a[((j) - 1) * (w.nest + 1) + (i)]
#define B_(i, j) w.b[((j) - 1) * (w.nest + 1) + (i)]
#define G_(i, j) w.g[((j) - 1) * (w.nest + 1) + (i)]
#define Q_(i, j) w.q[((j) - 1) * (w.m + 1) + (i)]

// fpbspl: the k+1 non-zero cubic B-splines at x, t(l) <= x < t(l+1)
ITD_HD void fpbspl(const double *t, double x, int l, double *h /* [1..6] */)
{
    double hh[6];
    h[1] = 1.0;
    for (int j = 1; j <= K; ++j) {
        for (int i = 1; i <= j; ++i) hh[i] = h[i];
        h[1] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = l + i, lj = li - j;
            if (t[li] == t[lj]) { h[i + 1] = 0.0; continue; }
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] = h[i] + f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
}

ITD_HD void fpgivs(double piv, double &ww, double &cs, double &sn)
{
    const double store = piv < 0 ? -piv : piv;
    double dd;
    if (store >= ww) { const double r = ww / piv; dd = store * sqrt(1.0 + r * r); }
    else { const double r = piv / ww; dd = ww * sqrt(1.0 + r * r); }
    cs = ww / dd;
    sn = piv / dd;
    ww = dd;
}

ITD_HD void fprota(double cs, double sn, double &a, double &b)
{
    const double stor1 = a, stor2 = b;
    b = cs * stor2 + sn * stor1;
    a = cs * stor1 - sn * stor2;
}

// fpback: back substitution of the banded upper triangular system (bandwidth k), columns of `a` 1-based with leading dim nest+1
ITD_HD void fpback(const double *a, int lda, const double *z, int n, int k, double *c)
{
    const int k1 = k - 1;
    c[n] = z[n] / a[n];                       // a(n,1)
    int i = n - 1;
    for (int j = 2; j <= n; ++j) {
        double store = z[i];
        const int i1 = (j <= k1) ? j - 1 : k1;
        int mm = i;
        for (int l = 1; l <= i1; ++l) {
            ++mm;
            store = store - c[mm] * a[l * lda + i];   // a(i,l+1)
        }
        c[i] = store / a[i];                  // a(i,1)
        --i;
    }
}

ITD_HD double fprati(double &p1, double &f1, double p2, double f2, double &p3, double &f3)
{
    double p;
    if (p3 > 0.0) {
        const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
    } else {
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
    }
    if (f2 < 0.0) { p3 = p2; f3 = f2; }
    else { p1 = p2; f1 = f2; }
    return p;
}

// fpknot: one more knot, at a data site in the interval of largest residual
ITD_HD void fpknot(const double *x, Work &w, int &n, int &nrint)
{
    const int k = (n - nrint - 1) / 2;
    double fpmax = 0.0;
    int jbegin = 1, number = 0, maxpt = 0, maxbeg = 0;
    for (int j = 1; j <= nrint; ++j) {
        const int jpoint = w.nrdata[j];
        if (!(fpmax >= w.fpint[j] || jpoint == 0)) {
            fpmax = w.fpint[j];
            number = j;
            maxpt = jpoint;
            maxbeg = jbegin;
        }
        jbegin = jbegin + jpoint + 1;
    }
    const int ihalf = maxpt / 2 + 1;
    const int nrx = maxbeg + ihalf;
    const int next = number + 1;
    if (next <= nrint) {
        for (int j = next; j <= nrint; ++j) {
            const int jj = next + nrint - j;
            w.fpint[jj + 1] = w.fpint[jj];
            w.nrdata[jj + 1] = w.nrdata[jj];
            const int jk = jj + k;
            w.t[jk + 1] = w.t[jk];
        }
    }
    w.nrdata[number] = ihalf - 1;
    w.nrdata[next] = maxpt - ihalf;
    const double am = maxpt;
    double an = w.nrdata[number];
    w.fpint[number] = fpmax * an / am;
    an = w.nrdata[next];
    w.fpint[next] = fpmax * an / am;
    const int jk = next + k;
    w.t[jk] = x[nrx];
    n = n + 1;
    nrint = nrint + 1;
}

// fpdisc: jumps of the third derivative of the B-splines at the interior knots
ITD_HD void fpdisc(Work &w, int n)
{
    double h[13];
    const int k1 = K2 - 1, k = k1 - 1, nk1 = n - k1, nrint = nk1 - k;
    const double an = nrint;
    const double fac = an / (w.t[nk1 + 1] - w.t[k1]);
    for (int l = K2; l <= nk1; ++l) {
        const int lmk = l - k1;
        for (int j = 1; j <= k1; ++j) {
            const int ik = j + k1, lj = l + j, lk = lj - K2;
            h[j] = w.t[l] - w.t[lk];
            h[ik] = w.t[l] - w.t[lj];
        }
        int lp = lmk;
        for (int j = 1; j <= K2; ++j) {
            int jk = j;
            double prod = h[j];
            for (int i = 1; i <= k; ++i) {
                ++jk;
                prod = prod * h[jk] * fac;
            }
            const int lk = lp + k1;
            B_(lmk, j) = (w.t[lk] - w.t[lp]) / prod;
            ++lp;
        }
    }
}

// curfit / fpcurf for iopt = 0, k = 3, unit weights.  x, y: 1-based [1..m], x strictly increasing.  On return w.t[1..n],
// w.c[1..n-4]; returns ier (0, -1, -2: fine; 1, 2, 3: FITPACK's warnings, the spline is still returned like splrep does).
ITD_HD int curfit(const double *x, const double *y, int m, double s, Work &w, int &n_out, double &fp_out)
{
    const double tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04, half = 0.5;
    const int maxit = 20, nest = w.nest, nmin = 2 * K1, nmax = m + K1;
    const double xb = x[1], xe = x[m];
    const double acc = tol * s;
    int n = nmin, nplus = 0, ier = 0, nrint = 0, nk1 = 0;
    double fpold = 0.0, fp0 = 0.0, fp = 0.0, fpms = 0.0;
    double h[8];
    w.nrdata[1] = m - 2;
    if (!(s > 0.0)) {
        // s = 0: the interpolating spline — splrep's default when no weights are given, i.e. what the reference's
        // custom_splrep(x, y) asks for.  All m + 4 knots at once (not-a-knot: the data sites x(3) .. x(m-2) are the interior
        // knots, fpcurf label 10), one pass of the main loop, ier = -1.
        n = nmax;
        const int mk1 = m - K1;
        int i2 = K2, j2 = K / 2 + 2;
        for (int lq = 1; lq <= mk1; ++lq) { w.t[i2] = x[j2]; ++i2; ++j2; }
    }
    bool smoothing = false;   // leave part 1 for part 2 (label 250)
    bool restart_interp = true;
    // ---- part 1 ---------------------------------------------------------------------------------------------------
    while (restart_interp && !smoothing) {
    restart_interp = false;
    for (int iter = 1; iter <= m; ++iter) {
        if (n == nmin) ier = -2;
        nrint = n - nmin + 1;
        nk1 = n - K1;
        {
            int i = n;
            for (int j = 1; j <= K1; ++j) { w.t[j] = xb; w.t[i] = xe; --i; }
        }
        fp = 0.0;
        for (int i = 1; i <= nk1; ++i) {
            w.z[i] = 0.0;
            for (int j = 1; j <= K1; ++j) A_(i, j) = 0.0;
        }
        int l = K1;
        for (int it = 1; it <= m; ++it) {
            const double xi = x[it];
            double yi = y[it];
            while (!(xi < w.t[l + 1] || l == nk1)) ++l;
            fpbspl(w.t, xi, l, h);
            for (int i = 1; i <= K1; ++i) Q_(it, i) = h[i];
            int j = l - K1;
            for (int i = 1; i <= K1; ++i) {
                ++j;
                const double piv = h[i];
                if (piv == 0.0) continue;
                double cs, sn;
                fpgivs(piv, A_(j, 1), cs, sn);
                fprota(cs, sn, yi, w.z[j]);
                if (i == K1) break;
                int i2 = 1;
                for (int i1 = i + 1; i1 <= K1; ++i1) {
                    ++i2;
                    fprota(cs, sn, h[i1], A_(j, i2));
                }
            }
            fp = fp + yi * yi;
        }
        if (ier == -2) fp0 = fp;
        w.fpint[n] = fp0;
        w.fpint[n - 1] = fpold;
        w.nrdata[n] = nplus;
        fpback(w.a, w.nest + 1, w.z, nk1, K1, w.c);
        fpms = fp - s;
        if ((fpms < 0 ? -fpms : fpms) < acc) { n_out = n; fp_out = fp; return ier; }
        if (fpms < 0.0) { smoothing = true; break; }
        if (n == nmax) { ier = -1; n_out = n; fp_out = fp; return ier; }
        if (n == nest) { ier = 1; n_out = n; fp_out = fp; return ier; }
        if (ier == 0) {
            int npl1 = nplus * 2;
            const double rn = nplus;
            if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
            int mx = npl1 > nplus / 2 ? npl1 : nplus / 2;
            if (mx < 1) mx = 1;
            nplus = nplus * 2 < mx ? nplus * 2 : mx;
        } else {
            nplus = 1;
            ier = 0;
        }
        fpold = fp;
        // residuals per knot interval
        double fpart = 0.0;
        int i = 1, nw = 0;
        l = K2;
        for (int it = 1; it <= m; ++it) {
            if (!(x[it] < w.t[l] || l > nk1)) { nw = 1; ++l; }
            double term = 0.0;
            int l0 = l - K2;
            for (int j = 1; j <= K1; ++j) { ++l0; term = term + w.c[l0] * Q_(it, j); }
            term = (term - y[it]) * (term - y[it]);
            fpart = fpart + term;
            if (nw == 0) continue;
            const double store = term * half;
            w.fpint[i] = fpart - store;
            ++i;
            fpart = store;
            nw = 0;
        }
        w.fpint[nrint] = fpart;
        for (int lq = 1; lq <= nplus; ++lq) {
            fpknot(x, w, n, nrint);
            if (n == nmax) { restart_interp = true; break; }
            if (n == nest) break;
        }
        if (restart_interp) {
            // n = nmax: the knots are placed as for interpolation (fpcurf label 10) and the main loop starts over — its least-
            // squares spline then interpolates (fp = 0 < s), so it goes on to part 2 with all m + 4 knots
            const int mk1 = m - K1;
            if (mk1 != 0) {
                int i2 = K2, j2 = K / 2 + 2;
                for (int lq = 1; lq <= mk1; ++lq) { w.t[i2] = x[j2]; ++i2; ++j2; }   // k odd
            }
            break;
        }
    }
    }
    if (!smoothing) { n_out = n; fp_out = fp; return ier; }   // the trial bound m was exhausted (never in practice)
    // ---- part 2: the smoothing spline, F(p) = s -------------------------------------------------------------------------
    fpdisc(w, n);
    double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 0.0;
    for (int i = 1; i <= nk1; ++i) p = p + A_(i, 1);
    const double rn = nk1;
    p = rn / p;
    int ich1 = 0, ich3 = 0;
    const int n8 = n - nmin;
    for (int iter = 1; iter <= maxit; ++iter) {
        const double pinv = 1.0 / p;
        for (int i = 1; i <= nk1; ++i) {
            w.c[i] = w.z[i];
            G_(i, K2) = 0.0;
            for (int j = 1; j <= K1; ++j) G_(i, j) = A_(i, j);
        }
        for (int it = 1; it <= n8; ++it) {
            for (int i = 1; i <= K2; ++i) h[i] = B_(it, i) * pinv;
            double yi = 0.0;
            for (int j = it; j <= nk1; ++j) {
                const double piv = h[1];
                double cs, sn;
                fpgivs(piv, G_(j, 1), cs, sn);
                fprota(cs, sn, yi, w.c[j]);
                if (j == nk1) break;
                int i2 = K1;
                if (j > n8) i2 = nk1 - j;
                for (int i = 1; i <= i2; ++i) {
                    const int i1 = i + 1;
                    fprota(cs, sn, h[i1], G_(j, i1));
                    h[i] = h[i1];
                }
                h[i2 + 1] = 0.0;
            }
        }
        fpback(w.g, w.nest + 1, w.c, nk1, K2, w.c);
        fp = 0.0;
        int l = K2;
        for (int it = 1; it <= m; ++it) {
            if (!(x[it] < w.t[l] || l > nk1)) ++l;
            int l0 = l - K2;
            double term = 0.0;
            for (int j = 1; j <= K1; ++j) { ++l0; term = term + w.c[l0] * Q_(it, j); }
            fp = fp + (term - y[it]) * (term - y[it]);
        }
        fpms = fp - s;
        if ((fpms < 0 ? -fpms : fpms) < acc) { n_out = n; fp_out = fp; return ier; }
        if (iter == maxit) { n_out = n; fp_out = fp; return 3; }
        const double p2 = p, f2 = fpms;
        if (ich3 == 0) {
            if (!((f2 - f3) > acc)) {
                p3 = p2; f3 = f2;
                p = p * con4;
                if (p <= p1) p = p1 * con9 + p2 * con1;
                continue;
            }
            if (f2 < 0.0) ich3 = 1;
        }
        if (ich1 == 0) {
            if (!((f1 - f2) > acc)) {
                p1 = p2; f1 = f2;
                p = p / con4;
                if (p3 < 0.0) continue;
                if (p >= p3) p = p2 * con1 + p3 * con9;
                continue;
            }
            if (f2 > 0.0) ich1 = 1;
        }
        if (f2 >= f1 || f2 <= f3) { n_out = n; fp_out = fp; return 2; }
        p = fprati(p1, f1, p2, f2, p3, f3);
    }
    n_out = n;
    fp_out = fp;
    return ier;
}

// numba_splev (numba_accelerated_itd.py:89-164) for one argument: t, c 0-based as scipy returns them (c zero padded to n),
// n knots; `equi` = the reference's equi_spaced flag with the data spacing dx (its interval formula is used verbatim,
// although it looks at the DATA spacing while indexing the KNOTS).  `l` is the reference's search state (k1 at the start).
ITD_HD double splev1(const double *t, const double *c, int n, double arg, bool equi, double dx, int &l, int &l1)
{
    const int k = K, k1 = K1, k2 = K2, nk1 = n - k1;
    if (equi) {
        l = (int)((arg - t[0]) / dx) + k;
        l = l < k1 ? k1 : l;
        l = l > nk1 ? nk1 : l;
    } else {
        while (!((arg >= t[l - 1]) || (l1 == k2))) { l1 = l; l = l - 1; }
        while (!((arg < t[l1 - 1]) || (l == nk1))) { l = l1; l1 = l + 1; }
    }
    double h[20], hh[19];
    for (int i = 0; i < 20; ++i) h[i] = 0.0;
    for (int i = 0; i < 19; ++i) hh[i] = 0.0;
    h[0] = 1.0;
    for (int j = 0; j < k; ++j) {
        for (int ll = 0; ll <= j; ++ll) hh[ll] = h[ll];
        h[0] = 0.0;
        for (int ll = 0; ll <= j; ++ll) {
            const int li = l + ll, lj = li - j - 1;
            if (t[li] != t[lj]) {
                const double f = hh[ll] / (t[li] - t[lj]);
                h[ll] += f * (t[li] - arg);
                h[ll + 1] = f * (arg - t[lj]);
            } else {
                h[ll + 1] = 0.0;
                break;
            }
        }
    }
    double sp = 0.0;
    int ll = l - 1 - k1;
    for (int j = 0; j < k1; ++j) { ++ll; sp += c[ll] * h[j]; }
    return sp;
}

#undef A_
#undef B_
#undef G_
#undef Q_

}  // namespace itd_fp

// ---------------------------------------------------------------------------------------------------------------------------
// The s = 0 call specialised for the GPU: one instance per signal, its working arrays interleaved with those of the other
// signals (element i of signal `sig` at [i * stride + sig]: the lanes of a wavefront — neighbouring signals at the same step —
// touch consecutive addresses).  Same operations in the same order as curfit() above with s = 0, so the same bits.
// The data sites are the extended knot list of the ITD baseline, e[0..m-1] (int32 sample indices, e[0] = 0, e[m-1] = n-1);
// FITPACK's knot vector follows from them and is never stored:
//     t(1..4) = e[0],  t(i) = e[i-3] for 5 <= i <= m,  t(m+1..m+4) = e[m-1]          (1-based; n = m + 4 knots, nk1 = m)
// y(it) is produced by the caller's functor (the baseline knot values, numba_accelerated_itd.py:196-206).
// ---------------------------------------------------------------------------------------------------------------------------
namespace itd_fp {

ITD_HD double knot_t(const int32_t *e, int m, int i /* 1-based */)
{
    return (double)(i <= K1 ? e[0] : (i <= m ? e[i - 3] : e[m - 1]));
}

// a: 4 columns of (m+1) entries, z: (m+1) entries, all 1-based and interleaved with `stride`; on return z holds c(1..m).
// The sweep touches only rows l-3 .. l of the triangle for the data point in knot interval l, and l never decreases: those four
// rows (and their right-hand sides) live in a sliding window of locals (registers on the GPU: every index below is a compile-time
// constant) and a row goes to memory once, when the window moves past it.  The back substitution keeps the three coefficients it
// has just solved in locals as well.  Arithmetic and its order are fpcurf's (the first version walked `a` and `z` in memory: on the
// GPU, where one thread sweeps one signal, every rotation was a dependent global-memory round trip).
template <typename YF>
ITD_HD void interp_fit(const int32_t *e, int m, YF y_of, double *a, double *z, int64_t stride, int64_t lda /* = m_max + 1 */)
{
#define SA_(i, j) a[(((int64_t)(j) - 1) * lda + (i)) * stride]
#define SZ_(i) z[(int64_t)(i) * stride]
    const int nk1 = m;
    double W[K1][K1], Z[K1];       // window row r <-> row l - 3 + r of the triangle: W[r][c] = a(row, c + 1), Z[r] = z(row)
#pragma unroll
    for (int r = 0; r < K1; ++r) {
        Z[r] = 0.0;
#pragma unroll
        for (int c = 0; c < K1; ++c) W[r][c] = 0.0;
    }
    int l = K1;
    double h[8], tl[8];
    for (int it = 1; it <= m; ++it) {
        const double xi = (double)e[it - 1];
        double yi = y_of(it - 1);
        while (!(xi < knot_t(e, m, l + 1) || l == nk1)) {
            // the window moves on: its first row (l - 3) is final
            const int row = l - 3;
#pragma unroll
            for (int c = 0; c < K1; ++c) SA_(row, c + 1) = W[0][c];
            SZ_(row) = Z[0];
#pragma unroll
            for (int r = 0; r < K1 - 1; ++r) {
                Z[r] = Z[r + 1];
#pragma unroll
                for (int c = 0; c < K1; ++c) W[r][c] = W[r + 1][c];
            }
            Z[K1 - 1] = 0.0;
#pragma unroll
            for (int c = 0; c < K1; ++c) W[K1 - 1][c] = 0.0;
            ++l;
        }
        // fpbspl with the knots t(l-2) .. t(l+3) in registers: tl[q] = t(l - 2 + q)
        for (int q = 0; q < 6; ++q) tl[q] = knot_t(e, m, l - 2 + q);
        {
            double hh[6];
            h[1] = 1.0;
            for (int j = 1; j <= K; ++j) {
                for (int i = 1; i <= j; ++i) hh[i] = h[i];
                h[1] = 0.0;
                for (int i = 1; i <= j; ++i) {
                    const double tli = tl[i + 2], tlj = tl[i - j + 2];      // t(l+i), t(l+i-j)
                    if (tli == tlj) { h[i + 1] = 0.0; continue; }
                    const double f = hh[i] / (tli - tlj);
                    h[i] = h[i] + f * (tli - xi);
                    h[i + 1] = f * (xi - tlj);
                }
            }
        }
        // rotate the new row (h(1..4) in columns l-3 .. l) into rows l-3 .. l: row l - 4 + i is window row i - 1
#pragma unroll
        for (int i = 1; i <= K1; ++i) {
            const double piv = h[i];
            if (piv == 0.0) continue;
            double cs, sn, ww = W[i - 1][0];
            fpgivs(piv, ww, cs, sn);
            W[i - 1][0] = ww;
            double zj = Z[i - 1];
            fprota(cs, sn, yi, zj);
            Z[i - 1] = zj;
            if (i == K1) break;
#pragma unroll
            for (int i1 = i + 1; i1 <= K1; ++i1) {
                const int i2 = i1 - i + 1;
                double aj = W[i - 1][i2 - 1];
                fprota(cs, sn, h[i1], aj);
                W[i - 1][i2 - 1] = aj;
            }
        }
    }
    // the window's rows l-3 .. l (= nk1-3 .. nk1 once every data point is in)
#pragma unroll
    for (int r = 0; r < K1; ++r) {
        const int row = l - 3 + r;
        if (row >= 1 && row <= nk1) {
#pragma unroll
            for (int c = 0; c < K1; ++c) SA_(row, c + 1) = W[r][c];
            SZ_(row) = Z[r];
        }
    }
    // fpback(a, z, nk1, k1, c): in place in z; z1, z2, z3 = c(i+1), c(i+2), c(i+3)
    double z1 = SZ_(nk1) / SA_(nk1, 1), z2 = 0.0, z3 = 0.0;
    SZ_(nk1) = z1;
    int i = nk1 - 1;
    for (int j = 2; j <= nk1; ++j) {
        double store = SZ_(i);
        const int i1 = (j <= K) ? j - 1 : K;
        if (i1 >= 1) store = store - z1 * SA_(i, 2);
        if (i1 >= 2) store = store - z2 * SA_(i, 3);
        if (i1 >= 3) store = store - z3 * SA_(i, 4);
        const double zi = store / SA_(i, 1);
        SZ_(i) = zi;
        z3 = z2;
        z2 = z1;
        z1 = zi;
        --i;
    }
#undef SA_
#undef SZ_
}

// numba_splev for one argument against the implicit knot vector (0-based: t[j] = knot_t(j+1)) and c (1-based, strided)
ITD_HD double spline_eval(const int32_t *e, int m, const double *c, int64_t stride, double arg, bool equi, double dx)
{
    const int n = m + K1, nk1 = n - K1;
    int l;
    if (equi) {
        l = (int)((arg - (double)e[0]) / dx) + K;
        l = l < K1 ? K1 : l;
        l = l > nk1 ? nk1 : l;
    } else {
        // first l >= k1 with arg < t[l] (0-based), capped at nk1: number of knots <= arg
        // t[0..3] = e[0] <= arg always; interior t[j] = e[j-2], 4 <= j <= m-1; t[m..m+3] = e[m-1]
        int lo = 2, hi = m - 2;                 // e[lo..hi-1] are the interior knots; first index with e[idx] > arg
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((double)e[mid] <= arg) lo = mid + 1; else hi = mid;
        }
        l = lo + 2;                             // 4 + (lo - 2) interior knots <= arg
        if (lo == m - 2 && arg >= (double)e[m - 1]) l = n;
        l = l > nk1 ? nk1 : l;
        l = l < K1 ? K1 : l;
    }
    double h[5] = {1.0, 0.0, 0.0, 0.0, 0.0}, hh[4];
    for (int j = 0; j < K; ++j) {
        for (int ll = 0; ll <= j; ++ll) hh[ll] = h[ll];
        h[0] = 0.0;
        for (int ll = 0; ll <= j; ++ll) {
            const int li = l + ll, lj = li - j - 1;
            const double tli = knot_t(e, m, li + 1), tlj = knot_t(e, m, lj + 1);
            if (tli != tlj) {
                const double f = hh[ll] / (tli - tlj);
                h[ll] += f * (tli - arg);
                h[ll + 1] = f * (arg - tlj);
            } else {
                h[ll + 1] = 0.0;
                break;
            }
        }
    }
    double sp = 0.0;
    int ll = l - 1 - K1;                        // 0-based coefficient index; c is stored 1-based
    for (int j = 0; j < K1; ++j) { ++ll; sp += c[(int64_t)(ll + 1) * stride] * h[j]; }
    return sp;
}

}  // namespace itd_fp
