// itd_kernels.hpp — CDNA4 (gfx950) device code of the ITD engine.
//
// One extraction level of the reference (itd_baseline_extract, ITD.py:79-121) is ONE streaming pass
// over the level's input plus one small compaction launch:
//
//   k_detect   (level 0 only) 3-point knot predicate of detect_peaks(x) U detect_peaks(-x)
//              (ITD.py:44-59, :87-98) on an LDS-staged tile, wavefront __ballot + popcount ordered
//              compaction into a per-tile knot list.
//   k_compact  per-tile counts -> exclusive offsets, per-tile lists -> one ordered knot index array
//              e[0..m+1] with e[0] = 0, e[m+1] = N-1 (ITD.py:95-98); evaluates the stop rule
//              `num_extrema < 2` (ITD.py:404) on the device so the host never synchronises per level.
//   k_extract  knot values B_k (ITD.py:100-110), per-segment slope, the affine baseline map
//              (ITD.py:114-117), rotation = x - baseline (ITD.py:119), AND the knot predicate of the
//              NEXT level on the baseline tile it has just produced (fused k_detect), so every level
//              after the first reads its input exactly once.
//   k_finalize applies the stop rule's row fix-up (ITD.py:404-416) on the device.
//
// Arithmetic: binary64, the reference's association order, no contraction (this file is compiled
// with -ffp-contract=off and carries the pragma below), IEEE division.  Differences (x[i+1]-x[i]) are
// formed exactly like numpy forms dx and then compared with 0, so +-inf inputs behave identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace itd {

constexpr int kThreads = 256;          // 4 wavefronts of 64
constexpr int kWaves = kThreads / 64;
constexpr int kMaxLevels = 24;         // levels 0 .. max_iteration+2 (<= 22) + slack

// Per-signal device state (one per batch element).
struct SigState {
    int32_t m[kMaxLevels];   // m[j] = interior knot count of the input of extraction j+1; -1 = not evaluated
    int32_t stopped;         // 1 once the natural stop rule fired (later kernels return at once)
    int32_t stop_level;      // level j (>= 1) whose input had < 2 knots: the reference stops at counter j-1
    int32_t nan_mask;        // bit j set: the baseline written by extraction j+1 contains a NaN
    int32_t pad;
};

enum DetectMode : int { kKnots = 0, kValleys = 1, kPeaks = 2 };

// ---------------------------------------------------------------------------------------------
// LDS tile: T samples + 1 halo sample each side.  tile[pos + 2] <-> sample s + pos, pos in [-1, T];
// the body starts 16-byte aligned so the staging stores are ds_write_b128.
// ---------------------------------------------------------------------------------------------
template <int T>
struct Tile {
    static constexpr int kSize = T + 4;
    double *p;
    __device__ __forceinline__ double &at(int pos) const { return p[pos + 2]; }
};

template <typename Tin, int T>
__device__ __forceinline__ void load_tile(const Tin *__restrict__ x, int64_t n, int64_t s, Tile<T> tile)
{
    const int tid = threadIdx.x;
    const Tin *src = x + s;
    const bool vec = (s + T <= n) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    if (vec) {
        if constexpr (sizeof(Tin) == 8) {
            using V = __attribute__((ext_vector_type(2))) double;
            const V *v = reinterpret_cast<const V *>(src);
#pragma unroll
            for (int k = tid; k < T / 2; k += kThreads) {
                V q = v[k];
                *reinterpret_cast<V *>(&tile.p[2 + 2 * k]) = q;
            }
        } else {
            using V = __attribute__((ext_vector_type(4))) float;
            using D2 = __attribute__((ext_vector_type(2))) double;
            const V *v = reinterpret_cast<const V *>(src);
#pragma unroll
            for (int k = tid; k < T / 4; k += kThreads) {
                V q = v[k];
                D2 lo = {(double)q.x, (double)q.y};
                D2 hi = {(double)q.z, (double)q.w};
                *reinterpret_cast<D2 *>(&tile.p[2 + 4 * k]) = lo;
                *reinterpret_cast<D2 *>(&tile.p[4 + 4 * k]) = hi;
            }
        }
    } else {
        for (int k = tid; k < T; k += kThreads) {
            const int64_t i = s + k;
            tile.at(k) = (i < n) ? (double)x[i] : 0.0;
        }
    }
    if (tid == 0) tile.at(-1) = (s >= 1) ? (double)x[s - 1] : 0.0;
    if (tid == 64) tile.at(T) = (s + T < n) ? (double)x[s + T] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Knot predicate + ordered compaction of one LDS tile into its per-tile list.
//   vil = x[i+1]-x[i], vix = x[i]-x[i-1];  valley: vil > 0 & vix <= 0  (detect_peaks(x),  ITD.py:59)
//                                          peak:   vil < 0 & vix >= 0  (detect_peaks(-x): the same test
//                                                                       on the exactly negated differences)
// First and last sample are never knots (ITD.py:70-73).  Each wavefront owns a contiguous quarter of
// the tile and walks it in 64-sample groups: __ballot gives the group's flag mask, popcounts give the
// ordered output slot.  Returns the tile's knot count to every thread.
// ---------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ int detect_tile(Tile<T> tile, int64_t s, int64_t n, int mode,
                                           int32_t *__restrict__ list, int *wave_counts /* LDS[kWaves] */)
{
    constexpr int CH = T / kWaves;   // samples per wavefront
    constexpr int G = CH / 64;       // 64-sample groups per wavefront
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    unsigned long long masks[G];
    int cnt = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int pos = w * CH + g * 64 + lane;
        const int64_t i = s + pos;
        const double xm = tile.at(pos - 1), x0 = tile.at(pos), xp = tile.at(pos + 1);
        const double vil = xp - x0;
        const double vix = x0 - xm;
        const bool valley = (vil > 0.0) && (vix <= 0.0);
        const bool peak = (vil < 0.0) && (vix >= 0.0);
        bool f = (mode == kKnots) ? (valley || peak) : (mode == kValleys ? valley : peak);
        f = f && (i >= 1) && (i <= n - 2);
        masks[g] = __ballot(f);
        cnt += __popcll(masks[g]);
    }
    if (lane == 0) wave_counts[w] = cnt;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
        const int c = wave_counts[k];
        if (k < w) base += c;
        total += c;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const unsigned long long mk = masks[g];
        if ((mk >> lane) & 1ull) {
            const int pos = w * CH + g * 64 + lane;
            list[base + __popcll(mk & lt)] = (int32_t)(s + pos);
        }
        base += __popcll(mk);
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// k_detect: level-0 knot scan of the caller's signal (float32 or float64 input).
// grid = (n_tiles, batch).  lists: [batch][n_tiles][T] int32, counts: [batch][n_tiles].
// ---------------------------------------------------------------------------------------------
template <typename Tin, int T>
__global__ __launch_bounds__(kThreads) void k_detect(const Tin *__restrict__ x, int64_t x_stride, int64_t n,
                                                     int n_tiles, int mode, int32_t *__restrict__ lists,
                                                     int32_t *__restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) double s_x[Tile<T>::kSize];
    __shared__ int s_wc[kWaves];
    const int t = blockIdx.x;
    const int sig = blockIdx.y;
    const int64_t s = (int64_t)t * T;
    Tile<T> tile{s_x};
    load_tile<Tin, T>(x + (int64_t)sig * x_stride, n, s, tile);
    __syncthreads();
    const size_t slot = (size_t)sig * n_tiles + t;
    const int total = detect_tile<T>(tile, s, n, mode, lists + slot * T, s_wc);
    if (threadIdx.x == 0) counts[slot] = total;
}

// ---------------------------------------------------------------------------------------------
// k_compact: per-tile lists -> ordered knot array e[0..m+1] and per-tile exclusive offsets.
// grid = (ceil(n_tiles / kTilesPerBlock), batch).  Each block first reduces the counts of all tiles in
// front of its own (a few KB from L2), then scans its own tiles with one wavefront and copies.
// The last block also publishes m, the end knots and the stop rule for level `level`.
// ---------------------------------------------------------------------------------------------
constexpr int kTilesPerBlock = 64;

template <int T>
__global__ __launch_bounds__(kThreads) void k_compact(const int32_t *__restrict__ lists,
                                                      const int32_t *__restrict__ counts, int n_tiles, int64_t n,
                                                      int32_t *__restrict__ kidx, int64_t kidx_stride,
                                                      int32_t *__restrict__ offsets, SigState *__restrict__ state,
                                                      int level, int check_stop)
{
    __shared__ int s_red[kWaves];
    __shared__ int s_off[kTilesPerBlock + 1];
    const int sig = blockIdx.y;
    SigState *st = state + sig;
    if (st->stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t *cnt = counts + (size_t)sig * n_tiles;
    const int t0 = blockIdx.x * kTilesPerBlock;
    // knots in front of this block's first tile
    int acc = 0;
    for (int k = tid; k < t0; k += kThreads) acc += cnt[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) s_red[w] = acc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) base += s_red[k];
    // exclusive scan of this block's (up to 64) tile counts by wavefront 0
    if (w == 0) {
        const int t = t0 + lane;
        const int c = (t < n_tiles) ? cnt[t] : 0;
        int inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(inc, d);
            if (lane >= d) inc += v;
        }
        s_off[lane] = base + inc - c;
        if (lane == 63) s_off[64] = base + inc;
    }
    __syncthreads();
    int32_t *e = kidx + (size_t)sig * kidx_stride;
    int32_t *off = offsets + (size_t)sig * (n_tiles + 1);
    const int nt = min(kTilesPerBlock, n_tiles - t0);
    if (tid < nt) off[t0 + tid] = s_off[tid];
    for (int k = 0; k < nt; ++k) {
        const int o = s_off[k], c = s_off[k + 1] - o;
        const int32_t *src = lists + ((size_t)sig * n_tiles + t0 + k) * T;
        for (int j = tid; j < c; j += kThreads) e[1 + o + j] = src[j];
    }
    if (t0 + nt == n_tiles && tid == 0) {
        const int m = s_off[nt];
        off[n_tiles] = m;
        e[0] = 0;                   // ITD.py:96
        e[m + 1] = (int32_t)(n - 1); // ITD.py:98
        st->m[level] = m;
        if (check_stop && m < 2) {  // ITD.py:404: the pending baseline is not decomposable
            st->stop_level = level;
            st->stopped = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_extract: one extraction on one tile.
//   xin        level input (float32/float64 caller signal at level 0, float64 baseline afterwards)
//   kidx/off   this level's ordered knots and per-tile offsets (from k_compact)
//   rot_out    rotation row   (FINAL: rotation + baseline, the "Out of time!" row, ITD.py:420)
//   base_out   baseline row   (FINAL: not written)
//   lists/counts  per-tile knot lists of the NEXT level (knots of the baseline just produced)
// grid = (n_tiles, batch).
// ---------------------------------------------------------------------------------------------
template <typename Tin, int T, bool FINAL>
__global__ __launch_bounds__(kThreads) void k_extract(const Tin *__restrict__ xin, int64_t x_stride, int64_t n,
                                                      int n_tiles, const int32_t *__restrict__ kidx,
                                                      int64_t kidx_stride, const int32_t *__restrict__ offsets,
                                                      double *__restrict__ rot_out, int64_t rot_stride,
                                                      double *__restrict__ base_out, int64_t base_stride,
                                                      int32_t *__restrict__ lists, int32_t *__restrict__ counts,
                                                      SigState *__restrict__ state, int level)
{
    constexpr int CH = T / kWaves;
    constexpr int G = CH / 64;
    constexpr int W = T / 64;  // 64-bit flag words per tile
    __shared__ __attribute__((aligned(16))) double s_x[Tile<T>::kSize];
    __shared__ __attribute__((aligned(16))) double s_b[Tile<T>::kSize];
    __shared__ double s_B[T];       // knot value  B_k   at the knot's position in the tile
    __shared__ double s_S[T];       // slope of the segment that STARTS at that knot
    __shared__ int32_t s_kk[T + 8]; // knot indices of extended ranks a-1 .. b+3
    __shared__ unsigned long long s_bm[W];
    __shared__ double s_hx[5], s_hB[5], s_hS[5]; // out-of-tile knots: [0]=a-1 [1]=a [2]=b+1 [3]=b+2 [4]=b+3
    __shared__ double s_ends[4];                 // x[0], x[1], x[n-2], x[n-1]
    __shared__ int s_wc[kWaves];
    __shared__ int s_lastpos[kWaves];

    const int t = blockIdx.x;
    const int sig = blockIdx.y;
    SigState *st = state + sig;
    if (st->stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t s = (int64_t)t * T;
    const Tin *x = xin + (int64_t)sig * x_stride;
    Tile<T> xt{s_x}, bt{s_b};

    load_tile<Tin, T>(x, n, s, xt);

    const int32_t *off = offsets + (size_t)sig * (n_tiles + 1);
    const int a = off[t];         // interior knots in front of the tile = extended rank of the carry-in knot
    const int b = off[t + 1];
    const int m = off[n_tiles];
    const int c = b - a;          // knots inside the tile: extended ranks a+1 .. b
    const int32_t *e = kidx + (size_t)sig * kidx_stride;
    for (int j = tid; j < c + 5; j += kThreads) {
        int r = a - 1 + j;
        r = r < 0 ? 0 : (r > m + 1 ? m + 1 : r);
        s_kk[j] = e[r];
    }
    if (tid < W) s_bm[tid] = 0ull;
    if (tid >= 64 && tid < 68) {
        const int q = tid - 64;
        const int64_t i = (q == 0) ? 0 : (q == 1 ? 1 : (q == 2 ? n - 2 : n - 1));
        s_ends[q] = (double)x[i];
    }
    __syncthreads();
    // values at the (up to) five knots outside the tile; flag bits of the knots inside it
    if (tid < 5) {
        const int j = tid < 2 ? tid : c + tid;
        s_hx[tid] = (double)x[s_kk[j]];
    }
    for (int j = 2 + tid; j <= c + 1; j += kThreads) {
        const int pos = s_kk[j] - (int32_t)s;
        atomicOr(&s_bm[pos >> 6], 1ull << (pos & 63));
    }
    __syncthreads();

    // ---- knot values, ITD.py:100-110.  j indexes s_kk; extended rank r = a-1+j ----------------
    auto in_tile = [&](int j) { return j >= 2 && j <= c + 1; };
    auto hslot = [&](int j) { return j < 2 ? j : j - c; };
    auto xval = [&](int j) -> double { return in_tile(j) ? xt.at(s_kk[j] - (int32_t)s) : s_hx[hslot(j)]; };
    for (int j = 1 + tid; j <= c + 3; j += kThreads) {
        const int r = a - 1 + j;
        double Bv;
        if (r <= 0) {
            Bv = (s_ends[0] + s_ends[1]) / 2.0;   // numpy.mean(x[:2]),  ITD.py:101
        } else if (r >= m + 1) {
            Bv = (s_ends[2] + s_ends[3]) / 2.0;   // numpy.mean(x[-2:]), ITD.py:102
        } else {
            const int32_t e0 = s_kk[j - 1], e1 = s_kk[j], e2 = s_kk[j + 1];
            const double x0 = xval(j - 1), x1 = xval(j), x2 = xval(j + 1);
            const double frac = (double)(e1 - e0) / (double)(e2 - e0);
            const double tt = frac * (x2 - x0);
            const double u = x0 + tt;
            Bv = 0.5 * u + 0.5 * x1;              // ITD.py:107-110
        }
        if (in_tile(j)) s_B[s_kk[j] - (int32_t)s] = Bv;
        else s_hB[hslot(j)] = Bv;
    }
    __syncthreads();
    // ---- per-segment slope (B_{k+1}-B_k)/(x[e_{k+1}]-x[e_k]), ITD.py:115-116 ---------------------
    auto Bof = [&](int j) -> double { return in_tile(j) ? s_B[s_kk[j] - (int32_t)s] : s_hB[hslot(j)]; };
    for (int j = 1 + tid; j <= c + 2; j += kThreads) {
        const int r = a - 1 + j;
        if (r > m) continue; // rank m+1 (= sample n-1) starts no segment
        const double sl = (Bof(j + 1) - Bof(j)) / (xval(j + 1) - xval(j));
        if (in_tile(j)) s_S[s_kk[j] - (int32_t)s] = sl;
        else s_hS[hslot(j)] = sl;
    }
    __syncthreads();

    // ---- baseline map + rotation, ITD.py:114-119 -------------------------------------------------
    // Segment of sample i = the last knot at or before i: inside the tile that is a max-scan of the flag
    // words; before the tile's first knot it is the carry-in knot (rank a).
    double *rot = rot_out + (int64_t)sig * rot_stride;
    double *bas = FINAL ? nullptr : base_out + (int64_t)sig * base_stride;
    int carry = -1;
    {
        // last knot position in the words in front of this wavefront's chunk
        const int nw = w * (CH / 64);
        int v = -1;
        if (lane < nw) {
            const unsigned long long q = s_bm[lane];
            if (q) v = lane * 64 + 63 - __clzll(q);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
        carry = v;
    }
    const double cB = s_hB[1], cS = s_hS[1], cX = s_hx[1];
    bool has_nan = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int gbase = w * CH + g * 64;
        const int pos = gbase + lane;
        const int64_t i = s + pos;
        const unsigned long long mk = s_bm[gbase >> 6];
        const unsigned long long le = mk & ((2ull << lane) - 1ull);
        const int P = le ? (gbase + 63 - __clzll(le)) : carry;
        const double xi = xt.at(pos);
        double Bk, Sk, Xk;
        if (P >= 0) { Bk = s_B[P]; Sk = s_S[P]; Xk = xt.at(P); }
        else        { Bk = cB;     Sk = cS;     Xk = cX; }
        double bi = Bk + Sk * (xi - Xk);
        if (i >= n - 1) bi = 0.0;             // baseline[n-1] is never written, ITD.py:112-117
        const double ri = xi - bi;
        bt.at(pos) = bi;
        if (i < n) {
            if constexpr (FINAL) {
                rot[i] = ri + bi;             // ITD.py:420
            } else {
                rot[i] = ri;
                bas[i] = bi;
            }
            has_nan = has_nan || (bi != bi);
        }
        if (mk) carry = gbase + 63 - __clzll(mk);
    }
    if (lane == 0) s_lastpos[w] = carry;
    // halo sample s-1 lives in the carry-in segment (rank a)
    if (tid == 0) bt.at(-1) = (s >= 1) ? (cB + cS * (xt.at(-1) - cX)) : 0.0;
    __syncthreads();
    if (tid == 0) {
        // halo sample s+T: either a knot itself (rank b+1) or in the tile's last segment
        const int64_t i = s + T;
        double v = 0.0;
        if (i < n - 1) {
            const double xi = xt.at(T);
            if (b + 1 <= m && s_kk[c + 2] == (int32_t)i) {
                v = s_hB[2] + s_hS[2] * (xi - s_hx[2]);
            } else {
                const int P = s_lastpos[kWaves - 1];
                if (P >= 0) v = s_B[P] + s_S[P] * (xi - xt.at(P));
                else v = cB + cS * (xi - cX);
            }
        }
        bt.at(T) = v;
    }
    if (__any(has_nan) && lane == 0) atomicOr(&st->nan_mask, 1 << level);
    __syncthreads();

    // ---- knots of the baseline just produced = the next level's input ---------------------------
    const size_t slot = (size_t)sig * n_tiles + t;
    const int total = detect_tile<T>(bt, s, n, kKnots, lists + slot * T, s_wc);
    if (tid == 0) counts[slot] = total;
}

// ---------------------------------------------------------------------------------------------
// k_finalize: the stop rule's row fix-up on the device (ITD.py:404-416).
// Natural stop detected at level j (input of extraction j+1 has < 2 knots) means the reference's
// counter is c = j-1: row c of the result is baselines[c-1] (= the input of extraction j, B_c), or the
// untouched all-zero last row when c = 0.  grid = (blocks, batch).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_finalize(double *__restrict__ rows, int64_t rows_stride, int64_t n,
                                                       const double *__restrict__ bases, int64_t bases_stride,
                                                       int64_t bases_row_pitch, int bases_pingpong,
                                                       const SigState *__restrict__ state)
{
    const int sig = blockIdx.y;
    const SigState *st = state + sig;
    if (!st->stopped) return;
    const int c = st->stop_level - 1;
    double *dst = rows + (int64_t)sig * rows_stride + (int64_t)c * n;
    // B_c was written by extraction c (level c-1) into baselines row c-1 (ping-pong: slot (c-1)&1)
    const double *src = nullptr;
    if (c >= 1) {
        const int row = bases_pingpong ? ((c - 1) & 1) : (c - 1);
        src = bases + (int64_t)sig * bases_stride + (int64_t)row * bases_row_pitch;
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        dst[i] = src ? src[i] : 0.0;
}

// gather x[e_k] for the host-facing single-level operator (knot values come back through the host)
template <typename Tin>
__global__ void k_knot_values(const Tin *__restrict__ x, int64_t n, const int32_t *__restrict__ e, int m,
                              double *__restrict__ bk)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > m + 1) return;
    double v;
    if (k == 0) v = ((double)x[0] + (double)x[1]) / 2.0;
    else if (k == m + 1) v = ((double)x[n - 2] + (double)x[n - 1]) / 2.0;
    else {
        const int32_t e0 = e[k - 1], e1 = e[k], e2 = e[k + 1];
        const double x0 = (double)x[e0], x1 = (double)x[e1], x2 = (double)x[e2];
        const double frac = (double)(e1 - e0) / (double)(e2 - e0);
        const double tt = frac * (x2 - x0);
        const double u = x0 + tt;
        v = 0.5 * u + 0.5 * x1;
    }
    bk[k] = v;
}

}  // namespace itd
