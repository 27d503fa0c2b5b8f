// weighted_permutation_entropy(x, order=3, delay=1), MEITD.py:79-128, as far as it touches the samples: every window
// (x[i], x[i+1], x[i+2]) falls into one of the six permutation patterns and carries its variance as weight; the entropy is
// drawn on the host from the six sums.  The reference's arithmetic, step by step:
//   sorted_idx = _embed(x).argsort(kind="quicksort")   numpy sorts 3 values by insertion on the index list, NaNs last — ties keep
//                                                      their index order (checked against numpy on rows with ties and NaNs)
//   hashval    = (sorted_idx * [1, 3, 9]).sum(1)       5, 7, 11, 15, 19, 21 — `bins` below in this (numpy.unique's) order
//   weights    = numpy.var(windows, 1)                 mean = ((a0 + a1) + a2) / 3; ((d0*d0 + d1*d1) + d2*d2) / 3
//   counts[h]  = cumsum(weights[hashval == h])[-1]     one by one, in index order
// Signals up to kWpeExactWindows windows are summed in exactly that order (one lane per pattern, one pass over the windows);
// longer ones in segments of kWpeSeg windows, each summed in order, the segments' sums then added in order — deterministic,
// equal to the reference's sum to rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace itd {
constexpr int kWpeSeg = 4096;
constexpr int64_t kWpeExactWindows = 1 << 16;

// numpy's order of doubles: a NaN sorts behind everything (npy_sort's LT for floating types)
__device__ __forceinline__ bool np_lt(double a, double b) { return a < b || (b != b && a == a); }

// a window's pattern (0..5, in the order above) and weight
__device__ __forceinline__ void wpe3_window(double a0, double a1, double a2, unsigned char &bin, double &wt)
{
    // insertion sort of the indices 0, 1, 2 by value
    double v0 = a0, v1 = a1, v2 = a2;
    int i0 = 0, i1 = 1, i2 = 2;
    if (np_lt(v1, v0)) { const double t = v0; v0 = v1; v1 = t; i0 = 1; i1 = 0; }
    if (np_lt(v2, v1)) {
        const double t = v2;
        v2 = v1; i2 = i1;
        if (np_lt(t, v0)) { v1 = v0; i1 = i0; v0 = t; i0 = 2; }
        else { v1 = t; i1 = 2; }
    }
    (void)v2;
    const int h = i0 + 3 * i1 + 9 * i2;
    bin = (unsigned char)(h == 5 ? 0 : h == 7 ? 1 : h == 11 ? 2 : h == 15 ? 3 : h == 19 ? 4 : 5);
    const double mean = ((a0 + a1) + a2) / 3.0;
    const double d0 = a0 - mean, d1 = a1 - mean, d2 = a2 - mean;
    wt = ((d0 * d0 + d1 * d1) + d2 * d2) / 3.0;
}

// One segment of windows per workgroup, through LDS in chunks of kWpeChunk windows.  Per chunk: the samples into LDS; every
// thread takes a run of consecutive windows and counts its patterns; a scan over the threads gives every window its place in its
// pattern's list (index order within a pattern is kept); the weights go there; lane b < 6 then adds pattern b's list to its
// running sum one by one — the chain of additions is the reference's cumsum, only the windows of other patterns are not in the way.
constexpr int kWpeChunk = 4096;
constexpr int kWpeThreads = 256;
// The pass over the windows lo .. hi-1 of x by the NT threads of one workgroup (s_x: kWpeChunk + 2 doubles, s_l: kWpeChunk doubles of
// LDS).  Threads 0..5 come back with their pattern's sum and window count in (s, c); every thread with its share of kn — the number
// of windows whose MIDDLE sample is a knot of x (ITD.py:59 on x and on -x, raw differences: what the count-only detection counts for
// samples 1 .. n-2) — and nanf, whether it saw a NaN.  The order of every pattern's additions does not depend on NT.
template <int NT>
__device__ __forceinline__ void wpe3_pass(const double *__restrict__ x, int64_t lo, int64_t hi, double *__restrict__ s_x, double *__restrict__ s_l,
                                          double &s, long long &c, int &kn, int &nanf, long long *prof = nullptr)
{
    static_assert(kWpeChunk % NT == 0 && kWpeChunk < 65536, "three 16-bit counts per word");
    __shared__ unsigned long long s_wave[2][NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    s = 0.0;                           // (0.0 + w == w: a variance is never -0.0)
    c = 0;
    kn = 0; nanf = 0;
    for (int64_t base = lo; base < hi; base += kWpeChunk) {
        const int m = (int)(hi - base < kWpeChunk ? hi - base : kWpeChunk);
        for (int k = tid; k < m + 2; k += NT) s_x[k] = x[base + k];
        __syncthreads();
        const int run = (m + NT - 1) / NT, k0 = tid * run < m ? tid * run : m, k1 = k0 + run < m ? k0 + run : m;
        constexpr int RUN = kWpeChunk / NT;              // (run <= RUN: a thread's windows' patterns and weights stay in registers)
        unsigned char bins[RUN];
        double wts[RUN];
        // patterns 0..2 count in the 16-bit fields of a, 3..5 in those of b
        unsigned long long a = 0ull, b = 0ull;
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
            const int k = k0 + r;
            bins[r] = 0; wts[r] = 0.0;
            if (k < k1) {
                const double a0 = s_x[k], a1 = s_x[k + 1], a2 = s_x[k + 2];
                wpe3_window(a0, a1, a2, bins[r], wts[r]);
                if (bins[r] < 3) a += 1ull << (16 * bins[r]); else b += 1ull << (16 * (bins[r] - 3));
                const double d0 = a1 - a0, d1 = a2 - a1;
                kn += ((d1 > 0.0 && d0 <= 0.0) || (d1 < 0.0 && d0 >= 0.0)) ? 1 : 0;
                nanf |= (a0 != a0 || a1 != a1 || a2 != a2) ? 1 : 0;
            }
        }
        // exclusive scan over the threads (in thread order = window order)
        unsigned long long ia = a, ib = b;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long ua = __shfl_up(ia, d), ub = __shfl_up(ib, d);
            if (lane >= d) { ia += ua; ib += ub; }
        }
        if (lane == 63) { s_wave[0][wave] = ia; s_wave[1][wave] = ib; }
        __syncthreads();
        unsigned long long ta = 0ull, tb = 0ull, pa = ia - a, pb = ib - b;      // totals; this thread's places
#pragma unroll
        for (int q = 0; q < NT / 64; ++q) {
            if (q < wave) { pa += s_wave[0][q]; pb += s_wave[1][q]; }
            ta += s_wave[0][q]; tb += s_wave[1][q];
        }
        auto field = [](unsigned long long v, int f) { return (int)((v >> (16 * f)) & 0xffffull); };
        int start[6], total[6];
        for (int q = 0, acc = 0; q < 6; ++q) { total[q] = q < 3 ? field(ta, q) : field(tb, q - 3); start[q] = acc; acc += total[q]; }
#pragma unroll
        for (int r = 0; r < RUN; ++r) {
            if (k0 + r < k1) {
                const int bin = bins[r];
                int pos;
                if (bin < 3) { pos = field(pa, bin); pa += 1ull << (16 * bin); } else { pos = field(pb, bin - 3); pb += 1ull << (16 * (bin - 3)); }
                s_l[start[bin == 0 ? 0 : bin == 1 ? 1 : bin == 2 ? 2 : bin == 3 ? 3 : bin == 4 ? 4 : 5] + pos] = wts[r];
            }
        }
        __syncthreads();
        const long long tp0 = prof ? (long long)wall_clock64() : 0;
        if (tid < 6) {
            int k = 0, e = 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) if (q == tid) { k = start[q]; e = start[q] + total[q]; }
            c += e - k;
            // blocks of 16 weights, the next block requested before this one's chain of additions starts (a chain of up to a few
            // thousand dependent additions on one lane: it should wait for nothing but itself)
            if (k + 16 <= e) {
                double w[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) w[q] = s_l[k + q];
                for (; k + 32 <= e; k += 16) {
                    double wn[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) wn[q] = s_l[k + 16 + q];
#pragma unroll
                    for (int q = 0; q < 16; ++q) s += w[q];
#pragma unroll
                    for (int q = 0; q < 16; ++q) w[q] = wn[q];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) s += w[q];
                k += 16;
            }
            for (; k < e; ++k) s += s_l[k];
        }
        __syncthreads();
        if (prof) prof[0] += (long long)wall_clock64() - tp0;
    }
}

// part_k (optional): per segment, the number of windows whose middle sample is a knot of x and whether a sample of the segment's
// windows is a NaN
__global__ __launch_bounds__(kWpeThreads) void k_wpe3(const double *__restrict__ x, int64_t nw, int64_t seg_len, double *__restrict__ part_s,
                                                      long long *__restrict__ part_c, int *__restrict__ part_k,
                                                      unsigned long long *__restrict__ done_words = nullptr, int32_t done_seq = 0
                                                      /* one segment, a polling host: the results as self-validating words (small_put) — words 0..11 the six
                                                         sums (low, high), 12..17 the six window counts, 18 the knots, 19 the NaN flag */)
{
    __shared__ double s_x[kWpeChunk + 2], s_l[kWpeChunk];
    __shared__ int s_kn[2];
    if (threadIdx.x < 2) s_kn[threadIdx.x] = 0;
    int kn, nanf;
    const int64_t lo = (int64_t)blockIdx.x * seg_len, hi = lo + seg_len < nw ? lo + seg_len : nw;
    const int tid = threadIdx.x;
    double s;
    long long c;
    wpe3_pass<kWpeThreads>(x, lo, hi, s_x, s_l, s, c, kn, nanf);
    if (tid < 6) {
        if (done_words) { small_put_f64(done_words, 2 * tid, s, (uint32_t)done_seq); small_put(done_words, 12 + tid, (uint32_t)c, (uint32_t)done_seq); }
        else { part_s[(size_t)blockIdx.x * 6 + tid] = s; part_c[(size_t)blockIdx.x * 6 + tid] = c; }
    }
    if (part_k || done_words) {
        if (kn) atomicAdd(&s_kn[0], kn);
        if (nanf) s_kn[1] = 1;
        __syncthreads();
        if (tid < 2) {
            if (done_words) small_put(done_words, 18 + tid, (uint32_t)s_kn[tid], (uint32_t)done_seq);
            else part_k[(size_t)blockIdx.x * 2 + tid] = s_kn[tid];
        }
    }
}

// the segments' sums, added in segment order
__global__ __launch_bounds__(64) void k_wpe3_combine(const double *__restrict__ part_s, const long long *__restrict__ part_c, int nseg,
                                                     double *__restrict__ out_s, long long *__restrict__ out_c, const int *__restrict__ part_k,
                                                     int *__restrict__ out_k)
{
    const unsigned bin = threadIdx.x;
    if (part_k && (bin == 6 || bin == 7)) {          // knots: a sum; NaN seen: an or
        int v = 0;
        for (int g = 0; g < nseg; ++g) v = bin == 6 ? v + part_k[(size_t)g * 2] : (v | part_k[(size_t)g * 2 + 1]);
        out_k[bin - 6] = v;
    }
    if (bin >= 6) return;
    double s = part_s[bin];
    long long c = part_c[bin];
    for (int g = 1; g < nseg; ++g) { s += part_s[(size_t)g * 6 + bin]; c += part_c[(size_t)g * 6 + bin]; }
    out_s[bin] = s;
    out_c[bin] = c;
}

// ---- any order 2 .. 5 (MEITD.py:79-128 takes `order`; MEITD itself only ever passes 3: the kernels above) ---------------------
// A window's hash is sum(sorted_idx[k] * order^k) < order^order (at most 3125); its weight numpy.var of its `order` values.  One thread
// per (hash value, segment) adds the weights of the segment's windows with that hash one by one in index order; the segments are added
// in order (k_wpe_combine).  Work grows with order^order x windows: a side path, not a fast one.
constexpr int kWpeMaxOrder = 5;
template <int ORDER>
__device__ __forceinline__ void wpe_window(const double *__restrict__ a, unsigned short &hash, double &wt)
{
    int idx[ORDER];
    double v[ORDER];
#pragma unroll
    for (int k = 0; k < ORDER; ++k) { idx[k] = k; v[k] = a[k]; }
#pragma unroll
    for (int i = 1; i < ORDER; ++i) {                // numpy's insertion sort of the index list (npy_sort, n <= 16): stable, NaNs last
        const double vi = v[i];
        const int ii = idx[i];
        int j = i;
#pragma unroll
        for (int t = 0; t < ORDER; ++t) {
            if (j > 0 && np_lt(vi, v[j > 0 ? j - 1 : 0])) { v[j] = v[j - 1]; idx[j] = idx[j - 1]; --j; }
        }
        v[j] = vi;
        idx[j] = ii;
    }
    int h = 0, mul = 1;
#pragma unroll
    for (int k = 0; k < ORDER; ++k) { h += idx[k] * mul; mul *= ORDER; }
    hash = (unsigned short)h;
    double sum = a[0];
#pragma unroll
    for (int k = 1; k < ORDER; ++k) sum += a[k];
    const double mean = sum / (double)ORDER;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < ORDER; ++k) { const double d = a[k] - mean; acc = k == 0 ? d * d : acc + d * d; }
    wt = acc / (double)ORDER;
}

template <int ORDER>
__global__ __launch_bounds__(256) void k_wpe_eval(const double *__restrict__ x, int64_t nw, unsigned short *__restrict__ hashes,
                                                  double *__restrict__ wts)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nw) return;
    double a[ORDER];
#pragma unroll
    for (int k = 0; k < ORDER; ++k) a[k] = x[i + k];
    wpe_window<ORDER>(a, hashes[i], wts[i]);
}

// grid = (ceil(nh / 256), segments): thread = one hash value over one segment of windows
__global__ __launch_bounds__(256) void k_wpe_sum(const unsigned short *__restrict__ hashes, const double *__restrict__ wts, int64_t nw,
                                                 int64_t seg_len, int nh, double *__restrict__ part_s, long long *__restrict__ part_c)
{
    __shared__ unsigned short s_h[1024];
    __shared__ double s_w[1024];
    const int h = blockIdx.x * 256 + threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.y * seg_len, hi = lo + seg_len < nw ? lo + seg_len : nw;
    double s = 0.0;
    long long c = 0;
    for (int64_t base = lo; base < hi; base += 1024) {
        const int m = (int)(hi - base < 1024 ? hi - base : 1024);
        for (int k = threadIdx.x; k < m; k += 256) { s_h[k] = hashes[base + k]; s_w[k] = wts[base + k]; }
        __syncthreads();
        if (h < nh)
            for (int k = 0; k < m; ++k)
                if (s_h[k] == (unsigned short)h) { s += s_w[k]; ++c; }
        __syncthreads();
    }
    if (h < nh) {
        part_s[(size_t)blockIdx.y * nh + h] = s;
        part_c[(size_t)blockIdx.y * nh + h] = c;
    }
}

__global__ __launch_bounds__(256) void k_wpe_combine(const double *__restrict__ part_s, const long long *__restrict__ part_c, int nseg, int nh,
                                                     double *__restrict__ out_s, long long *__restrict__ out_c)
{
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= nh) return;
    double s = part_s[h];
    long long c = part_c[h];
    for (int g = 1; g < nseg; ++g) { s += part_s[(size_t)g * nh + h]; c += part_c[(size_t)g * nh + h]; }
    out_s[h] = s;
    out_c[h] = c;
}

// out = a - b (MEITD.py:453 `x = x - rotation_[:]`)
__global__ __launch_bounds__(256) void k_subtract(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) out[i] = a[i] - b[i];
}
}  // namespace itd
