// itd_spline.hpp — the FITPACK flavour of the baseline, batched:
//     itd_baseline_extract_modified(x)     numba_accelerated_itd.py:182-211   (= itd_baseline_extract, siftED2D.ipynb cell 1)
//     itd_baseline_extract(data)           MEITD.py:303-338                    (the same operator without the "< 10" early-out)
// and what its 2-D consumers need around it (crossways_itd_baseline_extract, siftED2D.ipynb cell 1: transposes, the average).
// Per signal: knots (the tier-1 predicate: matlab_detect_peaks(x) U matlab_detect_peaks(-x) is the same set; the engine's
// k_detect + k_compact give the ordered list e = [0, knots, n-1]), baseline knot values with odd-reflected ends (:196-206),
// the interpolating cubic B-spline through them (custom_splrep -> scipy splrep, s = 0: itd_fitpack.hpp), evaluated at every
// sample (numba_splev, :89-164).
//   k_spline_fit    ONE THREAD PER SIGNAL: the spline's coefficients come from a serial sweep over the knots (Givens rotations
//                   of the banded collocation matrix, then back substitution); the working arrays of neighbouring signals are
//                   interleaved so that a wavefront's lanes (64 signals at the same knot) touch consecutive addresses.
//                   The workload this serves is many short signals (the reference's: 10 240 image rows of 512 samples).
//   k_spline_eval   one thread per sample: knot interval by bisection in the signal's knot list, de Boor.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "itd_fitpack.hpp"

#pragma clang fp contract(off)

namespace itd {

struct SplineMeta {
    int32_t m;        // data sites = knots + 2; 0: the signal is returned unchanged (fewer than min_extrema knots)
    int32_t equi;     // the reference's equi_spaced flag (numba_accelerated_itd.py:78-80)
    double dx;        // x_diff[0]
};

// baseline knot value k of signal x with the extended knot list e[0..m-1] (numba_accelerated_itd.py:196-206)
__device__ __forceinline__ double spline_knot_value(const double *__restrict__ x, int64_t n, const int32_t *__restrict__ e, int m, int k)
{
    if (k == 0) { const double p0 = 2 * x[0] - x[1]; return (p0 + x[0]) / 2.0; }                 // mean(padded[:2]), odd reflection
    if (k == m - 1) { const double pn = 2 * x[n - 1] - x[n - 2]; return (x[n - 1] + pn) / 2.0; } // mean(padded[-2:])
    const int32_t e0 = e[k - 1], e1 = e[k], e2 = e[k + 1];
    const double x0 = x[e0], x1 = x[e1], x2 = x[e2];
    const double frac = (double)(e1 - e0) / (double)(e2 - e0);
    const double tt = frac * (x2 - x0);
    const double u = x0 + tt;
    return 0.5 * u + 0.5 * x1;                                                                    // :171-176
}

// kidx: [batch][kidx_stride] ordered lists [0, knots, n-1]; totals: [batch][2] (knot count, NaN flag) from k_compact
__global__ __launch_bounds__(64) void k_spline_fit(const double *__restrict__ x, int64_t x_stride, int64_t n, int batch,
                                                   const int32_t *__restrict__ kidx, int64_t kidx_stride,
                                                   const int32_t *__restrict__ totals, int min_extrema,
                                                   double *__restrict__ a_ws, double *__restrict__ c_ws, int64_t lda,
                                                   SplineMeta *__restrict__ meta)
{
    const int sig = blockIdx.x * blockDim.x + threadIdx.x;
    if (sig >= batch) return;
    const int knots = totals[2 * sig];
    const int m = knots + 2;
    SplineMeta mt{0, 0, 1.0};
    if (knots >= min_extrema && knots >= 2) {      // splrep needs m > k
        const int32_t *e = kidx + (int64_t)sig * kidx_stride;
        const double *xs = x + (int64_t)sig * x_stride;
        itd_fp::interp_fit(e, m, [&](int k) { return spline_knot_value(xs, n, e, m, k); }, a_ws + sig, c_ws + sig, batch, lda);
        const int32_t d0 = e[1] - e[0];
        bool eq = true;
        for (int k = 2; k < m; ++k) eq = eq && (e[k] - e[k - 1] == d0);
        mt.m = m;
        mt.equi = eq ? 1 : 0;
        mt.dx = (double)d0;
    }
    meta[sig] = mt;
}

__global__ __launch_bounds__(256) void k_spline_eval(const double *__restrict__ x, int64_t x_stride, int64_t n, int batch,
                                                     const int32_t *__restrict__ kidx, int64_t kidx_stride,
                                                     const double *__restrict__ c_ws, const SplineMeta *__restrict__ meta,
                                                     double *__restrict__ base_out, int64_t base_stride,
                                                     double *__restrict__ rot_out, int64_t rot_stride)
{
    const int sig = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SplineMeta mt = meta[sig];
    const double xi = x[(int64_t)sig * x_stride + i];
    double b = xi;                                  // fewer than min_extrema knots: the signal itself (:188-190)
    if (mt.m > 0)
        b = itd_fp::spline_eval(kidx + (int64_t)sig * kidx_stride, mt.m, c_ws + sig, batch, (double)i, mt.equi != 0, mt.dx);
    base_out[(int64_t)sig * base_stride + i] = b;
    if (rot_out) rot_out[(int64_t)sig * rot_stride + i] = xi - b;     // MEITD.py:335
}

// out[c][r] = in[r][c], 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void k_transpose(const double *__restrict__ in, int rows, int cols, double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int64_t plane = (int64_t)blockIdx.z * rows * cols;
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = in[plane + (int64_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (r < rows && c < cols) out[plane + (int64_t)c * rows + r] = tile[tx][j];
    }
}

// out = (a + b) / 2.0   (crossways_itd_baseline_extract's last loop, siftED2D.ipynb cell 1)
__global__ void k_mean2(const double *__restrict__ a, const double *__restrict__ b, int64_t cnt, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) out[i] = (a[i] + b[i]) / 2.0;
}

}  // namespace itd
