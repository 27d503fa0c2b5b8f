// itd_detect_fast.hpp — ordered knot lists for the predicates of the cubic variant and of the instantaneous step
// (ITD_DETECT_CPP: itd.cpp:161-168; ITD_DETECT_ZERO_CROSS: find_extrema's test, itd_fourier_decomposition.py:23-27) in two short
// launches.  The general pair k_detect + k_compact (itd_kernels.hpp: LDS tile, per-tile position lists, records for an extraction
// that these callers never run) took 51 + 35 us at 2^24 samples, a third of the cubic operator's call; these two read the signal
// once from registers, keep a tile's knots as eight 64-bit words and write every position once.
// Same contract as k_detect + k_compact: counts, group sums, the ordered list [lead slot, knots, tail], totals (count, NaN flag),
// the knots in front of every tile.  NaN samples compare false (no knot there) and raise the state's in_nan, as in k_detect.
#pragma once
#include "itd_kernels.hpp"

namespace itd {

// grid = (n_tiles, batch), 64 threads: lane l holds samples s + 64 g + l
template <int MODE>
__global__ __launch_bounds__(kWave) void k_detect_fast(const double *__restrict__ x, int64_t x_stride, int64_t n, int n_tiles,
                                                       int32_t *__restrict__ counts, unsigned long long *__restrict__ fw,
                                                       int32_t *__restrict__ gsum_out, SigState *__restrict__ state)
{
    static_assert(MODE == (int)kCpp || MODE == (int)kZeroCross, "the two predicates without NaN rules of their own");
    constexpr int TW = 512, G = TW / 64;
    const int t = blockIdx.x, sig = blockIdx.y, lane = lane_id();
    const int64_t s = (int64_t)t * TW;
    const double *xs = x + (int64_t)sig * x_stride;
    double xr[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t j = s + g * 64 + lane;
        xr[g] = j < n ? xs[j] : 0.0;
    }
    const double x_before = s >= 1 ? xs[s - 1] : 0.0, x_after = s + TW < n ? xs[s + TW] : 0.0;   // wave-uniform
    bool nan_in = false;
    int total = 0;
    unsigned long long mine = 0ull;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t j = s + g * 64 + lane;
        const double x0 = xr[g];
        // the neighbours: the adjacent lanes' samples; across a 64-sample step the neighbouring step's edge lane (the broadcasts run
        // with every lane active: a shuffle inside the select would read an inactive lane)
        const double up = __shfl_up(x0, 1), dn = __shfl_down(x0, 1);
        const double prev_last = g > 0 ? __shfl(xr[g > 0 ? g - 1 : 0], 63) : x_before;
        const double next_first = g + 1 < G ? __shfl(xr[g + 1 < G ? g + 1 : g], 0) : x_after;
        const double xm = lane > 0 ? up : prev_last;
        const double xp = lane < 63 ? dn : next_first;
        bool f;
        if (MODE == (int)kCpp) f = ((xm < x0) && (x0 >= xp)) || ((xm > x0) && (x0 <= xp));     // itd.cpp:161-168
        else f = ((x0 > 0.0) && (0.0 > xp)) || ((x0 < 0.0) && (0.0 < xp));                       // itd_fourier_decomposition.py:23-27
        f = f && j >= 1 && j <= n - 2;                    // first and last sample never flag
        nan_in = nan_in || (j < n && x0 != x0);
        const unsigned long long mk = __ballot(f);
        total += __popcll(mk);
        if (lane == g) mine = mk;
    }
    const size_t slot = (size_t)sig * n_tiles + t;
    if (lane < G) fw[slot * G + lane] = mine;
    if (lane == 0) {
        counts[slot] = total;
        if (total) atomicAdd(&gsum_out[((size_t)sig * groups_of(n_tiles) + t / kTilesPerGroup) * kGsumPitch], total);
    }
    if (__any(nan_in) && lane == 0) state[sig].in_nan = 1;
}

// The common-baseline form of the cubic operator on complex (I/Q) data, itd.cpp:58-154: a knot is a sample at which BOTH components have
// an extremum under the file's 3-point predicate (:74-80); the scalar series the spline runs on is the components' mean (:96-103: the
// knot values are formed from (I + Q) / 2 at the knots).  One launch: the predicate on both components of the interleaved signal, the
// tile's flag words / count / group sum (k_detect_fast's contract: k_compact_fast follows) and the mean series avg[n].
// grid = (n_tiles, 1), 64 threads: lane l holds samples s + 64 g + l (one 16-byte load per sample: re, im)
__global__ __launch_bounds__(kWave) void k_detect_fast_iq(const double *__restrict__ iq, int64_t n, int n_tiles, int32_t *__restrict__ counts,
                                                          unsigned long long *__restrict__ fw, int32_t *__restrict__ gsum_out,
                                                          SigState *__restrict__ state, double *__restrict__ avg)
{
    constexpr int TW = 512, G = TW / 64;
    const int t = blockIdx.x, lane = lane_id();
    const int64_t s = (int64_t)t * TW;
    const double2 *z = reinterpret_cast<const double2 *>(iq);
    double re[G], im[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t j = s + g * 64 + lane;
        const double2 v = j < n ? z[j] : make_double2(0.0, 0.0);
        re[g] = v.x; im[g] = v.y;
        if (j < n) avg[j] = (v.x + v.y) / 2.0;                       // itd.cpp:101-103
    }
    const double2 zb = s >= 1 ? z[s - 1] : make_double2(0.0, 0.0), za = s + TW < n ? z[s + TW] : make_double2(0.0, 0.0);   // wave-uniform
    bool nan_in = false;
    int total = 0;
    unsigned long long mine = 0ull;
    auto flag = [&](const double (&x)[G], int g, double x_before, double x_after) {
        const double x0 = x[g];
        const double up = __shfl_up(x0, 1), dn = __shfl_down(x0, 1);
        const double prev_last = g > 0 ? __shfl(x[g > 0 ? g - 1 : 0], 63) : x_before;
        const double next_first = g + 1 < G ? __shfl(x[g + 1 < G ? g + 1 : g], 0) : x_after;
        const double xm = lane > 0 ? up : prev_last;
        const double xp = lane < 63 ? dn : next_first;
        return ((xm < x0) && (x0 >= xp)) || ((xm > x0) && (x0 <= xp));                // itd.cpp:77-80
    };
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int64_t j = s + g * 64 + lane;
        const bool fr = flag(re, g, zb.x, za.x), fi = flag(im, g, zb.y, za.y);
        const bool f = fr && fi && j >= 1 && j <= n - 2;
        nan_in = nan_in || (j < n && (re[g] != re[g] || im[g] != im[g]));
        const unsigned long long mk = __ballot(f);
        total += __popcll(mk);
        if (lane == g) mine = mk;
    }
    if (lane < G) fw[(size_t)t * G + lane] = mine;
    if (lane == 0) {
        counts[t] = total;
        if (total) atomicAdd(&gsum_out[(size_t)(t / kTilesPerGroup) * kGsumPitch], total);
    }
    if (__any(nan_in) && lane == 0) state[0].in_nan = 1;
}
// the mean series alone (the knots are the caller's: "retain the extrema and reuse them", itd.cpp:40-44)
__global__ void k_iq_mean(const double *__restrict__ iq, int64_t n, double *__restrict__ avg)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) { const double2 v = reinterpret_cast<const double2 *>(iq)[j]; avg[j] = (v.x + v.y) / 2.0; }
}

// grid = (n_tiles, batch), 64 threads: the tile's knots at their ranks (k_compact's contract)
__global__ __launch_bounds__(kWave) void k_compact_fast(const unsigned long long *__restrict__ fw, const int32_t *__restrict__ counts,
                                                        const int32_t *__restrict__ gsum_in, int n_tiles, int64_t n,
                                                        int32_t *__restrict__ kidx, int64_t kidx_stride, int32_t *__restrict__ total_out,
                                                        const SigState *__restrict__ state, int64_t tail_value,
                                                        int32_t *__restrict__ tile_base_out, int lead)
{
    constexpr int TW = 512, G = TW / 64;
    const int sig = blockIdx.y, t = blockIdx.x, lane = lane_id();
    const int n_groups = groups_of(n_tiles);
    const int32_t *cnt = counts + (size_t)sig * n_tiles;
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
    const size_t slot = (size_t)sig * n_tiles + t;
    unsigned long long w[G];
#pragma unroll
    for (int g = 0; g < G; ++g) w[g] = fw[slot * G + g];
    const int g0 = t / kTilesPerGroup;
    int acc = 0;
    for (int k = lane; k < g0; k += kWave) acc += gs[(size_t)k * kGsumPitch];
    if (lane < t - g0 * kTilesPerGroup) acc += cnt[g0 * kTilesPerGroup + lane];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    const int base = acc;
    if (tile_base_out && lane == 0) tile_base_out[slot] = base;
    int32_t *e = kidx + (size_t)sig * kidx_stride;
    const int64_t s = (int64_t)t * TW;
    int pre = base;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if ((w[g] >> lane) & 1ull) e[lead + pre + __popcll(w[g] & ((1ull << lane) - 1ull))] = (int32_t)(s + g * 64 + lane);
        pre += __popcll(w[g]);
    }
    if (lane == 0 && t == n_tiles - 1) {
        const int m = pre;
        if (lead) {
            e[0] = 0;
            e[m + 1] = (int32_t)(tail_value < 0 ? n - 1 : tail_value);
        }
        total_out[2 * sig] = m;
        total_out[2 * sig + 1] = state[sig].in_nan;      // set by k_detect_fast, which has completed
    }
}

}  // namespace itd
