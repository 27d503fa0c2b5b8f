// itd_kernels.hpp — CDNA4 (gfx950) device code of the ITD engine.
//
// One extraction level of the reference (itd_baseline_extract, ITD.py:79-121) is ONE kernel launch that
// streams the level's input once:
//
//   k_detect   (level 0 only) 3-point knot predicate of detect_peaks(x) U detect_peaks(-x)
//              (ITD.py:44-59, :87-98) on an LDS-staged tile; wavefront __ballot + popcount ordered
//              compaction into a per-tile knot list, plus a 64-byte per-tile record (count, first three
//              and last two knots with their values).
//   k_extract  finds the two knots in front of and the three knots behind its tile from the neighbouring
//              tiles' records (no global knot array, no prefix scan over the signal), computes the knot
//              values B_k (ITD.py:100-110), the per-segment slope, the affine baseline map
//              (ITD.py:114-117), rotation = x - baseline (ITD.py:119), AND the knot list/record of the
//              NEXT level from the baseline tile it has just produced.  Block 0 also totals the level's
//              knot count and evaluates the stop rule `num_extrema < 2` (ITD.py:404) on the device, so
//              the host never synchronises between levels.
//   k_compact  (API helpers only) per-tile lists -> one ordered index array e[0..m+1] (ITD.py:95-98).
//   k_last_count / k_finalize   stop test of the last pending baseline and the row fix-up (ITD.py:404-416).
//
// Arithmetic: binary64, the reference's association order, no contraction (compiled with
// -ffp-contract=off, and the pragma below), IEEE division.  Differences (x[i+1]-x[i]) are formed exactly
// like numpy forms dx and then compared with 0, so +-inf inputs behave identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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
    double ends[2][4];       // [level & 1]: x[0], x[1], x[n-2], x[n-1] of that level's input (ITD.py:101-102)
};

// What the producer of a tile's knot list leaves for the tile's neighbours (64 bytes, one per tile).
struct TileRec {
    int32_t count;     // knots in the tile
    int32_t hidx[3];   // its first min(count,3) knots ...
    int32_t tidx[2];   // ... and its last two: tidx[1] = last, tidx[0] = second last (count >= 2)
    double hval[3];    // values of the level's input at those knots
    double tval[2];
};
static_assert(sizeof(TileRec) == 64, "TileRec layout");

enum DetectMode : int { kKnots = 0, kValleys = 1, kPeaks = 2 };

// Tiles are grouped in runs of 64; every producer of a tile list adds its count to the group's sum (one
// atomic per non-empty tile).  Each group sum sits on a 128-byte line of its own: device-scope atomics on
// one line serialise (~3 ns each, measured).  Three buffers rotate by level: level j reads j%3, adds into
// (j+1)%3 and clears (j+2)%3.
constexpr int kTilesPerGroup = 64;
constexpr int kGsumPitch = 32;  // int32 elements between consecutive group sums
__host__ __device__ constexpr int groups_of(int n_tiles) { return (n_tiles + kTilesPerGroup - 1) / kTilesPerGroup; }

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

// Tile fetch split in two (issue early / commit late) so the HBM latency overlaps the halo search.
template <typename Tin, int T>
struct TileRegs {
    using V = typename std::conditional<sizeof(Tin) == 8, __attribute__((ext_vector_type(2))) double,
                                        __attribute__((ext_vector_type(4))) float>::type;
    static constexpr int kPer = sizeof(Tin) == 8 ? 2 : 4;
    static constexpr int kN = T / (kPer * kThreads);
    V q[kN];
    double lo, hi;
    bool vec;
};

template <typename Tin, int T>
__device__ __forceinline__ void tile_fetch(TileRegs<Tin, T> &r, const Tin *__restrict__ x, int64_t n, int64_t s)
{
    using R = TileRegs<Tin, T>;
    const int tid = threadIdx.x;
    const Tin *src = x + s;
    r.vec = (s + T <= n) && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);
    if (r.vec) {
        const typename R::V *v = reinterpret_cast<const typename R::V *>(src);
#pragma unroll
        for (int k = 0; k < R::kN; ++k) r.q[k] = v[tid + k * kThreads];
    }
    r.lo = 0.0;
    r.hi = 0.0;
    if (tid == 0 && s >= 1) r.lo = (double)x[s - 1];
    if (tid == 64 && s + T < n) r.hi = (double)x[s + T];
}

template <typename Tin, int T>
__device__ __forceinline__ void tile_commit(const TileRegs<Tin, T> &r, const Tin *__restrict__ x, int64_t n,
                                            int64_t s, Tile<T> tile)
{
    using R = TileRegs<Tin, T>;
    using D2 = __attribute__((ext_vector_type(2))) double;
    const int tid = threadIdx.x;
    if (r.vec) {
#pragma unroll
        for (int k = 0; k < R::kN; ++k) {
            const int e = (tid + k * kThreads) * R::kPer;
            if constexpr (sizeof(Tin) == 8) {
                *reinterpret_cast<D2 *>(&tile.p[2 + e]) = r.q[k];
            } else {
                D2 a = {(double)r.q[k].x, (double)r.q[k].y};
                D2 b = {(double)r.q[k].z, (double)r.q[k].w};
                *reinterpret_cast<D2 *>(&tile.p[2 + e]) = a;
                *reinterpret_cast<D2 *>(&tile.p[4 + e]) = b;
            }
        }
    } else {
        for (int k = tid; k < T; k += kThreads) {
            const int64_t i = s + k;
            tile.at(k) = (i < n) ? (double)x[i] : 0.0;
        }
    }
    if (tid == 0) tile.at(-1) = r.lo;
    if (tid == 64) tile.at(T) = r.hi;
}

// ---------------------------------------------------------------------------------------------
// Knot predicate + ordered compaction of one LDS tile into its per-tile list and record.
//   vil = x[i+1]-x[i], vix = x[i]-x[i-1];  valley: vil > 0 & vix <= 0  (detect_peaks(x),  ITD.py:59)
//                                          peak:   vil < 0 & vix >= 0  (detect_peaks(-x): the same test
//                                                                       on the exactly negated differences)
// First and last sample are never knots (ITD.py:70-73).  Each wavefront owns a contiguous quarter of
// the tile and walks it in 64-sample groups: __ballot gives the group's flag mask, popcounts give the
// ordered output slot.  Returns the tile's knot count to every thread.
// ---------------------------------------------------------------------------------------------
template <int T>
__device__ __forceinline__ int detect_tile(Tile<T> tile, int64_t s, int64_t n, int mode,
                                           int32_t *__restrict__ list, TileRec *__restrict__ rec,
                                           int *wave_counts /* LDS[kWaves] */)
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
            const int k = base + __popcll(mk & lt);
            const int32_t idx = (int32_t)(s + pos);
            list[k] = idx;
            if (k < 3) {
                rec->hidx[k] = idx;
                rec->hval[k] = tile.at(pos);
            }
            if (k >= total - 2) {
                const int q = 1 - (total - 1 - k);
                rec->tidx[q] = idx;
                rec->tval[q] = tile.at(pos);
            }
        }
        base += __popcll(mk);
    }
    if (threadIdx.x == 0) rec->count = total;
    return total;
}

// the four end samples of the level's input for the NEXT level (ITD.py:101-102), taken from an LDS tile
template <int T>
__device__ __forceinline__ void publish_ends(Tile<T> tile, int64_t s, int64_t n, double *ends)
{
    const int q = threadIdx.x;
    if (q < 4) {
        const int64_t i = (q == 0) ? 0 : (q == 1 ? 1 : (q == 2 ? n - 2 : n - 1));
        if (i >= s && i < s + T) ends[q] = tile.at((int)(i - s));
    }
}

// ---------------------------------------------------------------------------------------------
// k_detect: level-0 knot scan of the caller's signal (float32 or float64 input).
// grid = (n_tiles, batch).  lists: [batch][n_tiles][T] int32; counts: [batch][n_tiles]; recs likewise.
// ---------------------------------------------------------------------------------------------
template <typename Tin, int T>
__global__ __launch_bounds__(kThreads) void k_detect(const Tin *__restrict__ x, int64_t x_stride, int64_t n,
                                                     int n_tiles, int mode, int32_t *__restrict__ lists,
                                                     int32_t *__restrict__ counts, TileRec *__restrict__ recs,
                                                     int32_t *__restrict__ gsum_out, SigState *__restrict__ state)
{
    __shared__ __attribute__((aligned(16))) double s_x[Tile<T>::kSize];
    __shared__ int s_wc[kWaves];
    const int t = blockIdx.x;
    const int sig = blockIdx.y;
    const int64_t s = (int64_t)t * T;
    const Tin *xs = x + (int64_t)sig * x_stride;
    Tile<T> tile{s_x};
    TileRegs<Tin, T> regs;
    tile_fetch<Tin, T>(regs, xs, n, s);
    tile_commit<Tin, T>(regs, xs, n, s, tile);
    __syncthreads();
    publish_ends<T>(tile, s, n, state[sig].ends[0]);
    const size_t slot = (size_t)sig * n_tiles + t;
    const int total = detect_tile<T>(tile, s, n, mode, lists + slot * T, recs + slot, s_wc);
    if (threadIdx.x == 0) {
        counts[slot] = total;
        if (total) atomicAdd(&gsum_out[((size_t)sig * groups_of(n_tiles) + t / kTilesPerGroup) * kGsumPitch], total);
    }
}

// ---------------------------------------------------------------------------------------------
// k_compact (API helpers: detect_peaks / knot lists handed back to the caller):
// per-tile lists -> ordered knot array e[0..m+1], one block per tile.  The block sums the group totals in
// front of its group and the tile counts in front of it inside the group, then copies its list.
// ---------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(kThreads) void k_compact(const int32_t *__restrict__ lists,
                                                      const int32_t *__restrict__ counts,
                                                      const int32_t *__restrict__ gsum_in, int n_tiles, int64_t n,
                                                      int32_t *__restrict__ kidx, int64_t kidx_stride,
                                                      int32_t *__restrict__ total_out)
{
    __shared__ int s_red[kWaves];
    const int sig = blockIdx.y;
    const int t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n_groups = groups_of(n_tiles);
    const int32_t *cnt = counts + (size_t)sig * n_tiles;
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
    const int g = t / kTilesPerGroup;
    int acc = 0;
    for (int k = tid; k < g; k += kThreads) acc += gs[(size_t)k * kGsumPitch];
    if (tid < t - g * kTilesPerGroup) acc += cnt[g * kTilesPerGroup + tid];
    const int c = cnt[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) s_red[w] = acc;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) base += s_red[k];
    int32_t *e = kidx + (size_t)sig * kidx_stride;
    const int32_t *src = lists + ((size_t)sig * n_tiles + t) * T;
    for (int j = tid; j < c; j += kThreads) e[1 + base + j] = src[j];
    if (tid == 0 && t == n_tiles - 1) {
        const int m = base + c;
        e[0] = 0;                    // ITD.py:96
        e[m + 1] = (int32_t)(n - 1); // ITD.py:98
        total_out[sig] = m;
    }
}

// ---------------------------------------------------------------------------------------------
// Nearest non-empty tile at or beyond `start` in direction DIR (+1 / -1), wave-uniform; -1 if none.
// Only reached when the 64-tile window next to the tile is empty: walks the rest of start's group tile
// by tile, then skips empty groups through the group sums.
// ---------------------------------------------------------------------------------------------
template <int DIR>
__device__ int far_nonempty(const int32_t *__restrict__ cnts, const int32_t *__restrict__ gs, int n_tiles,
                            int start, int *count_out)
{
    const int lane = threadIdx.x & 63;
    const int n_groups = groups_of(n_tiles);
    while (start >= 0 && start < n_tiles) {
        const int g = start / kTilesPerGroup;
        const int tt = start + DIR * lane;
        const bool in = (tt >= g * kTilesPerGroup) && (tt < (g + 1) * kTilesPerGroup) && (tt < n_tiles);
        const int c = in ? cnts[tt] : 0;
        const unsigned long long mk = __ballot(c != 0);
        if (mk) {
            const int l = __ffsll((long long)mk) - 1;
            *count_out = __shfl(c, l);
            return start + DIR * l;
        }
        int gg = g + DIR;
        int hit = -1;
        while (gg >= 0 && gg < n_groups) {
            const int q = gg + DIR * lane;
            const int v = (q >= 0 && q < n_groups) ? gs[(size_t)q * kGsumPitch] : 0;
            const unsigned long long gm = __ballot(v != 0);
            if (gm) {
                hit = gg + DIR * (__ffsll((long long)gm) - 1);
                break;
            }
            gg += DIR * 64;
        }
        if (hit < 0) return -1;
        start = (DIR > 0) ? hit * kTilesPerGroup : min(hit * kTilesPerGroup + kTilesPerGroup - 1, n_tiles - 1);
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// k_extract: one extraction on one tile.
//   xin           level input (float32/float64 caller signal at level 0, float64 baseline afterwards)
//   lists/counts/recs/gsum_in   this level's per-tile knot lists, counts, records, group sums
//   rot_out       rotation row   (FINAL: rotation + baseline, the "Out of time!" row, ITD.py:420)
//   base_out      baseline row   (FINAL: not written)
//   the same lists/counts/recs arrays receive the NEXT level's data for this tile (a block reads only its
//   own list, and neighbours' records are double buffered by level parity); gsum_out / gsum_clear rotate.
// grid = (n_tiles, batch).
// ---------------------------------------------------------------------------------------------
template <typename Tin, int T, bool FINAL>
__global__ __launch_bounds__(kThreads) void k_extract(const Tin *__restrict__ xin, int64_t x_stride, int64_t n,
                                                      int n_tiles, int32_t *__restrict__ lists,
                                                      const int32_t *__restrict__ counts_in,
                                                      int32_t *__restrict__ counts_out,
                                                      const TileRec *__restrict__ recs_in,
                                                      TileRec *__restrict__ recs_out,
                                                      const int32_t *__restrict__ gsum_in,
                                                      int32_t *__restrict__ gsum_out, int32_t *__restrict__ gsum_clear,
                                                      double *__restrict__ rot_out, int64_t rot_stride,
                                                      double *__restrict__ base_out, int64_t base_stride,
                                                      SigState *__restrict__ state, int level)
{
    constexpr int CH = T / kWaves;
    constexpr int G = CH / 64;
    constexpr int W = T / 64;   // 64-bit flag words per tile
    constexpr int LPT = T / kThreads;  // list entries per thread
    __shared__ __attribute__((aligned(16))) double s_x[Tile<T>::kSize];
    __shared__ __attribute__((aligned(16))) double s_b[Tile<T>::kSize];
    __shared__ double s_B[T];       // knot value  B_k   at the knot's position in the tile
    __shared__ double s_S[T];       // slope of the segment that STARTS at that knot
    __shared__ int32_t s_kk[T + 8]; // knot indices: [0],[1] two knots in front, [2..c+1] the tile's, [c+2..c+4] behind
    __shared__ unsigned long long s_bm[W];
    __shared__ double s_hx[5], s_hB[5], s_hS[5]; // out-of-tile knots: [0],[1] in front; [2],[3],[4] behind
    __shared__ int32_t s_hidx[5];
    __shared__ int s_nb, s_nf;                   // real knots found in front (0..2) / behind (0..3)
    __shared__ int s_wc[kWaves];
    __shared__ int s_lastpos[kWaves];

    const int t = blockIdx.x;
    const int sig = blockIdx.y;
    SigState *st = state + sig;
    if (st->stopped) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t s = (int64_t)t * T;
    const Tin *x = xin + (int64_t)sig * x_stride;
    const size_t slot0 = (size_t)sig * n_tiles;
    const int32_t *cnts = counts_in + slot0;
    const TileRec *recs = recs_in + slot0;
    const int n_groups = groups_of(n_tiles);
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
    Tile<T> xt{s_x}, bt{s_b};

    // ---- issue every independent load first: the tile, this tile's knot list, the neighbours' counts -------
    TileRegs<Tin, T> regs;
    tile_fetch<Tin, T>(regs, x, n, s);
    const int c = cnts[t];   // knots inside the tile
    int32_t lr[LPT];
    {
        const int32_t *mylist = lists + (slot0 + t) * T;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int j = tid + k * kThreads;
            lr[k] = (j < c) ? mylist[j] : 0;
        }
    }
    const double e0 = st->ends[level & 1][0], e1 = st->ends[level & 1][1];
    const double e2 = st->ends[level & 1][2], e3 = st->ends[level & 1][3];

    if (w == 0) {
        // ---- halo search: two knots in front of the tile, three behind, from the neighbours' records ----
        const int tb = t - 1 - lane, tf = t + 1 + lane;
        const int cb = (tb >= 0) ? cnts[tb] : 0;
        const int cf = (tf < n_tiles) ? cnts[tf] : 0;
        {   // in front (nearest first): slot 1 = the knot that starts the segment entering the tile, slot 0 before it
            unsigned long long mk = __ballot(cb != 0);
            int found = 0, far = t - 65;
            while (found < 2) {
                int u, cu;
                if (mk) {
                    const int l = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    u = t - 1 - l;
                    cu = __shfl(cb, l);
                } else {
                    u = far_nonempty<-1>(cnts, gs, n_tiles, far, &cu);
                    if (u < 0) break;
                    far = u - 1;
                }
                const TileRec *r = recs + u;
                if (lane == 0) { s_hidx[1 - found] = r->tidx[1]; s_hx[1 - found] = r->tval[1]; }
                ++found;
                if (found < 2 && cu >= 2) {
                    if (lane == 0) { s_hidx[1 - found] = r->tidx[0]; s_hx[1 - found] = r->tval[0]; }
                    ++found;
                }
            }
            if (lane == 0) {
                s_nb = found;
                for (int q = found; q < 2; ++q) { s_hidx[1 - q] = 0; s_hx[1 - q] = e0; }  // e[0] = 0, ITD.py:96
            }
        }
        {   // behind (nearest first): slots 2, 3, 4
            unsigned long long mk = __ballot(cf != 0);
            int found = 0, far = t + 65;
            while (found < 3) {
                int u, cu;
                if (mk) {
                    const int l = __ffsll((long long)mk) - 1;
                    mk &= mk - 1;
                    u = t + 1 + l;
                    cu = __shfl(cf, l);
                } else {
                    u = far_nonempty<1>(cnts, gs, n_tiles, far, &cu);
                    if (u < 0) break;
                    far = u + 1;
                }
                const TileRec *r = recs + u;
                const int take = min(cu, 3 - found);
                if (lane == 0)
                    for (int q = 0; q < take; ++q) { s_hidx[2 + found + q] = r->hidx[q]; s_hx[2 + found + q] = r->hval[q]; }
                found += take;
            }
            if (lane == 0) {
                s_nf = found;
                for (int q = found; q < 3; ++q) { s_hidx[2 + q] = (int32_t)(n - 1); s_hx[2 + q] = e3; }  // e[m+1] = n-1, ITD.py:98
            }
        }
    } else if (w == 1 && t == 0) {
        // ---- block 0: total knot count of this level and the stop rule (ITD.py:400-404) ----------------------
        int acc = 0;
        for (int k = lane; k < n_groups; k += 64) acc += gs[(size_t)k * kGsumPitch];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) {
            st->m[level] = acc;
            if (level >= 1 && acc < 2) {   // the pending baseline is not decomposable: later launches return at once
                st->stop_level = level;
                st->stopped = 1;
            }
        }
    }
    if (t == 0 && w >= 2) {
        int32_t *gc = gsum_clear + (size_t)sig * n_groups * kGsumPitch;
        for (int k = tid - 128; k < n_groups; k += kThreads - 128) gc[(size_t)k * kGsumPitch] = 0;
    }

    tile_commit<Tin, T>(regs, x, n, s, xt);
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < c) s_kk[2 + j] = lr[k];
    }
    if (tid < W) s_bm[tid] = 0ull;
    __syncthreads();
    if (tid < 5) s_kk[tid < 2 ? tid : c + tid] = s_hidx[tid];
    // flag bits of the knots inside the tile
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int j = tid + k * kThreads;
        if (j < c) {
            const int pos = lr[k] - (int32_t)s;
            atomicOr(&s_bm[pos >> 6], 1ull << (pos & 63));
        }
    }
    __syncthreads();
    const int nb = s_nb, nf = s_nf;

    // ---- knot values, ITD.py:100-110.  j indexes s_kk ------------------------------------------------------
    auto in_tile = [&](int j) { return j >= 2 && j <= c + 1; };
    auto hslot = [&](int j) { return j < 2 ? j : j - c; };
    auto xval = [&](int j) -> double { return in_tile(j) ? xt.at(s_kk[j] - (int32_t)s) : s_hx[hslot(j)]; };
    auto is_end0 = [&](int j) { return (j == 1 && nb == 0) || (j == 0 && nb < 2); };   // e[0] = sample 0
    auto is_endn = [&](int j) { return j >= c + 2 && (j - (c + 2)) >= nf; };           // e[m+1] = sample n-1
    for (int j = 1 + tid; j <= c + 3; j += kThreads) {
        double Bv;
        if (is_end0(j)) {
            Bv = (e0 + e1) / 2.0;                 // numpy.mean(x[:2]),  ITD.py:101
        } else if (is_endn(j)) {
            Bv = (e2 + e3) / 2.0;                 // numpy.mean(x[-2:]), ITD.py:102
        } else {
            const int32_t k0 = s_kk[j - 1], k1 = s_kk[j], k2 = s_kk[j + 1];
            const double x0 = xval(j - 1), x1 = xval(j), x2 = xval(j + 1);
            const double frac = (double)(k1 - k0) / (double)(k2 - k0);
            const double tt = frac * (x2 - x0);
            const double u = x0 + tt;
            Bv = 0.5 * u + 0.5 * x1;              // ITD.py:107-110
        }
        if (in_tile(j)) s_B[s_kk[j] - (int32_t)s] = Bv;
        else s_hB[hslot(j)] = Bv;
    }
    __syncthreads();
    // ---- per-segment slope (B_{k+1}-B_k)/(x[e_{k+1}]-x[e_k]), ITD.py:115-116 -----------------------------
    auto Bof = [&](int j) -> double { return in_tile(j) ? s_B[s_kk[j] - (int32_t)s] : s_hB[hslot(j)]; };
    for (int j = 1 + tid; j <= c + 2; j += kThreads) {
        if (is_endn(j)) continue; // sample n-1 starts no segment
        const double sl = (Bof(j + 1) - Bof(j)) / (xval(j + 1) - xval(j));
        if (in_tile(j)) s_S[s_kk[j] - (int32_t)s] = sl;
        else s_hS[hslot(j)] = sl;
    }
    __syncthreads();

    // ---- baseline map + rotation, ITD.py:114-119 -----------------------------------------------------------
    // Segment of sample i = the last knot at or before i: inside the tile that is a max-scan of the flag
    // words; before the tile's first knot it is the carry-in knot (slot 1).
    double *rot = rot_out + (int64_t)sig * rot_stride;
    double *bas = FINAL ? nullptr : base_out + (int64_t)sig * base_stride;
    int carry = -1;
    {
        const int nw = w * (CH / 64);   // flag words in front of this wavefront's chunk
        int v = -1;
        if (lane < nw) {
            const unsigned long long q = s_bm[lane];
            if (q) v = lane * 64 + 63 - __clzll((long long)q);
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
        const int P = le ? (gbase + 63 - __clzll((long long)le)) : carry;
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
        if (mk) carry = gbase + 63 - __clzll((long long)mk);
    }
    if (lane == 0) s_lastpos[w] = carry;
    // halo sample s-1 lives in the carry-in segment
    if (tid == 0) bt.at(-1) = (s >= 1) ? (cB + cS * (xt.at(-1) - cX)) : 0.0;
    __syncthreads();
    if (tid == 0) {
        // halo sample s+T: either a knot itself (slot 2) or in the tile's last segment
        const int64_t i = s + T;
        double v = 0.0;
        if (i < n - 1) {
            const double xi = xt.at(T);
            if (nf >= 1 && s_hidx[2] == (int32_t)i) {
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

    // ---- knots of the baseline just produced = the next level's input ------------------------------------
    if constexpr (!FINAL) publish_ends<T>(bt, s, n, st->ends[(level + 1) & 1]);
    const size_t slot = slot0 + t;
    const int total = detect_tile<T>(bt, s, n, kKnots, lists + slot * T, recs_out + slot, s_wc);
    if (tid == 0) {
        counts_out[slot] = total;
        if (total) atomicAdd(&gsum_out[((size_t)sig * n_groups + t / kTilesPerGroup) * kGsumPitch], total);
    }
}

// ---------------------------------------------------------------------------------------------
// k_last_count: knot count of the last pending baseline and its stop test (ITD.py:400-404), which takes
// priority over the "Out of time!" branch.  grid = (batch), one wavefront each.
// ---------------------------------------------------------------------------------------------
__global__ void k_last_count(const int32_t *__restrict__ gsum_in, int n_tiles, SigState *__restrict__ state, int level)
{
    const int sig = blockIdx.x;
    SigState *st = state + sig;
    if (st->stopped) return;
    const int lane = threadIdx.x;
    const int n_groups = groups_of(n_tiles);
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
    int acc = 0;
    for (int k = lane; k < n_groups; k += 64) acc += gs[(size_t)k * kGsumPitch];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
        st->m[level] = acc;
        if (acc < 2) {
            st->stop_level = level;
            st->stopped = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_finalize: the stop rule's row fix-up on the device (ITD.py:404-416).
// Natural stop detected at level j (input of extraction j+1 has < 2 knots) means the reference's
// counter is c = j-1: row c of the result is baselines[c-1] (= the input of extraction j, B_c), or the
// untouched all-zero last row when c = 0.  grid = (blocks, batch).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_finalize(double *__restrict__ rows, int64_t rows_stride, int64_t n,
                                                       const double *__restrict__ bases, int64_t bases_stride,
                                                       int64_t bases_row_pitch, int bases_rotate,
                                                       const SigState *__restrict__ state)
{
    const int sig = blockIdx.y;
    const SigState *st = state + sig;
    if (!st->stopped) return;
    const int c = st->stop_level - 1;
    double *dst = rows + (int64_t)sig * rows_stride + (int64_t)c * n;
    // B_c was written by extraction c (level c-1) into baselines row c-1 (rotating slots: (c-1) % 3)
    const double *src = nullptr;
    if (c >= 1) {
        const int row = bases_rotate ? ((c - 1) % bases_rotate) : (c - 1);
        src = bases + (int64_t)sig * bases_stride + (int64_t)row * bases_row_pitch;
    }
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
        dst[i] = src ? src[i] : 0.0;
}

// knot values of an explicit knot list (host-facing single-level operator, ITD.py:100-110)
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
