// itd_kernels.hpp — CDNA4 (gfx950) device code of the ITD engine.
//
// One extraction level of the reference (itd_baseline_extract, ITD.py:79-121) is ONE kernel launch that
// streams the level's input once.  The unit of work is a TILE of TW consecutive samples owned by ONE
// 64-lane wavefront: the wavefront stages the tile in LDS, and everything after that (halo search, knot
// values, baseline map, rotation, next level's knot scan) is wave-synchronous — no workgroup barrier, no
// inter-workgroup communication inside a launch.
//
//   k_detect   (level 0 only) 3-point knot predicate of detect_peaks(x) U detect_peaks(-x)
//              (ITD.py:44-59, :87-98) on the LDS tile: one __ballot word per 64-sample group.  Leaves a
//              per-tile count, a group sum, and a 128-byte per-tile record: count, first three and last two
//              knots with their values (for the neighbours) and the flag words (for the tile itself).
//   k_extract  takes its own knots from the flag words of its record, finds the two knots in front of and the
//              three knots behind its tile in the neighbouring tiles' records (no global knot array, no knot
//              list, no prefix scan over the signal), computes the knot values B_k (ITD.py:100-110), the
//              per-segment slope, the affine baseline map (ITD.py:114-117), rotation = x - baseline
//              (ITD.py:119), AND the record of the NEXT level from the baseline tile it has just produced.
//              The wavefront of tile 0 also totals the level's knot count and evaluates the stop rule
//              `num_extrema < 2` (ITD.py:404) on the device, so the host never synchronises between levels.
//   k_compact  (API helpers only) per-tile lists -> one ordered index array e[0..m+1] (ITD.py:95-98).
//   k_finalize  stop test of the last pending baseline and the row fix-up (ITD.py:400-416).
//
// Arithmetic: binary64, the reference's association order, no contraction (compiled with
// -ffp-contract=off, and the pragma below), IEEE division.  Differences (x[i+1]-x[i]) are formed exactly
// like numpy forms dx and then compared with 0, so +-inf inputs behave identically.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#pragma clang fp contract(off)

// cache policy of k_extract's streams (bit 0: rotation stores, bit 1: baseline stores, bit 2: tile loads are
// nontemporal).  Rotation rows are never read again by the engine and a level's input is read exactly once, so both
// stream past the caches; the baseline is the next launch's input and stays cacheable (Infinity Cache hit).
#ifndef ITD_NT
#define ITD_NT 5
#endif
// timing-only ablations of k_extract (results are wrong by construction): 1 no count windows, 2 no speculative
// records, 8 no knots inside the tile, 16 no group-sum atomic, 32 no count store, 64 no record store, 512 (unused), 1024 no far
// search, 2048 no knot phases (one constant segment), 4096 no knot scan, 8192 no candidate selection / halo assembly,
// 16384 no writes to the signal's state, 32768 no side loads, 65536 no knot scan / outputs at all, 131072 / 262144 only the
// three streams (before / after the scalar state loads)
#ifndef ITD_ABL_R
#define ITD_ABL_R 0
#endif
#ifndef ITD_PRIO
#define ITD_PRIO 0
#endif
// 1: k_extract reads the signal's stop level before it issues its loads (stopped signals cost no traffic); 0: after
#ifndef ITD_EARLY_STOP
#define ITD_EARLY_STOP 1
#endif
#ifndef ITD_BS_SWEEP
#define ITD_BS_SWEEP 1   // knot values and slopes in one sweep over the slots (0: round 4's two loops, for A/B builds)
#endif
namespace itd {

constexpr int kWave = 64;              // one wavefront per tile, one tile per workgroup
// knots one pass of k_extract holds by rank (>= 128: a 128-sample group always fits); denser tiles take more passes
#ifndef ITD_RANK_CAP
#define ITD_RANK_CAP 136
#endif
constexpr int kRankCap = ITD_RANK_CAP;
// the level-0 launch reads the caller's signal, usually the densest knots of the decomposition: it gets more slots
#ifndef ITD_RANK_CAP0
#define ITD_RANK_CAP0 264
#endif
constexpr int kRankCap0 = ITD_RANK_CAP0;
// tiles per wavefront of k_extract (all their loads are issued up front).  2 hides the second tile's load latency and the
// first tile's store acknowledgements, but needs 109 VGPRs (4 wavefronts per SIMD) and measured 74 vs 70 us per level: 1 ships.
#ifndef ITD_KT
#define ITD_KT 1
#endif
constexpr int kTilesPerWave = ITD_KT;
#ifndef ITD_KT0
#define ITD_KT0 1
#endif
constexpr int kScanTilesPerWave = ITD_KT0;
#ifndef ITD_KT_FUSE0
#define ITD_KT_FUSE0 1
#endif
constexpr int kFuse0TilesPerWave = ITD_KT_FUSE0;   // tiles per wavefront of the fused level-0 launch (the second tile's loads fly during the first's work)   // tiles per wavefront of the level-0 scan (4 measured 30 us vs 26 us at 1: the scan is instruction bound)
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
constexpr int kMaxLevels = 24;         // levels 0 .. max_iteration+2 (<= 22) + slack

// Per-signal device state (one per batch element).
struct SigState {
    int32_t m[kMaxLevels];   // m[j] = interior knot count of the input of extraction j+1 (under the reference's counting
                             // rules, NaN rules included); -1 = not evaluated
    int32_t c_delta[kMaxLevels];   // [j]: what the reference's NaN rules add to the plain knot count of the baseline written by
                             // launch j (valleys of detect_peaks' NaN branch minus plain valleys of the mutated array,
                             // ITD.py:46-51,64-68); applies only if that baseline holds a NaN (nan_mask bit j)
    int32_t stop_level;      // -1, or the level j (>= 1) whose input had < 2 knots: the reference stops at counter j-1.
                             // Written by workgroup 0 of launch j; launches test `stop_level >= 0 && stop_level < level`,
                             // so the stop takes effect from launch j+1 on and launch j itself runs to completion in
                             // every workgroup (its results are discarded through n_rows, but they are deterministic)
    int32_t nan_mask;        // bit j set: the baseline written by launch j (extraction j+1) contained a NaN before the
                             // reference's in-place NaN -> +inf mutation (ITD.py:50)
    int32_t in_nan;          // the caller's signal contains a NaN (rejected: ITD_ERR_NONFINITE)
    int32_t l0_fail;         // the fused level-0 launch met a tile whose halo knots lie beyond its reach (kReach windows):
                             // level 0 has to be repeated through k_scan0 + the record-driven launch
    int32_t fin_stopped;     // written by k_finalize only: the verdict after the last pending baseline's stop test
    int32_t fin_stop_level;  // (kept apart from stop_level, which k_finalize's own workgroups still read)
    int32_t kf_fail;         // written by the fused sparse levels only (itd_knotfirst.hpp): != 0 = their result for this signal is not
                             // the reference's (a knot the knot-side recursion missed, capacity, non-finite knot data): the call is
                             // repeated level by level
    int32_t res_fail;        // written by k_resident only (itd_resident.hpp): the one-workgroup form met a non-finite sample (it
                             // handles finite data only): the call is repeated through the level-by-level engine
    int32_t skip;            // the device-side repair (itd_set_device_repair): 1 = this signal's result was final before the repair's
                             // launches, which return at once for it; the state is a copy of the first run's.  < 0: the repair re-ran
                             // this signal, -(1: the fused levels had refused | 2: fused level 0 out of reach | 4: resident form)
    double ends[2][4];       // [level & 1]: x[0], x[1], x[n-2], x[n-1] of that level's input (ITD.py:101-102)
};

#ifndef ITD_TILE
#define ITD_TILE 512
#endif
// Flag words of a tile (interleaved format): word 2g = knot flags of the EVEN positions of 128-sample group g (bit l <->
// position 128 g + 2 l), word 2g+1 = the odd ones — the layout in which k_extract holds a tile in registers.
__device__ __forceinline__ int flag_pos(int word, int bit) { return 128 * (word >> 1) + 2 * bit + (word & 1); }
constexpr int kMaxGroups = (ITD_TILE / 64 <= 8) ? 8 : 16;   // flag words per record (tiles of up to 512 / 1024 samples)

// What the producer of a tile's knots leaves for the next launch (128 bytes = one line per tile).  The first 64 bytes
// are for the tile's neighbours: they fetch them speculatively together with their own tile (k_extract), so they hold
// everything a neighbour can need — including the tile's first and last sample, which are the neighbour's samples s+TW
// and s-1: re-reading those two from the signal would touch lines the owning tile streams past the caches.
struct TileRec {
    unsigned long long packed;   // bits 0..10 knot count; 10-bit tile positions of the first three knots at bits 11, 21, 31,
                                 // of the second-last knot at bit 41 and of the last knot at bit 51
    double hval[3];              // values of the level's input at the first min(count,3) knots ...
    double tval[2];              // ... and at the last two: tval[1] = last, tval[0] = second last (count >= 2)
    double edge[2];              // the tile's first and last sample (positions 0 and TW-1) of the level's input
    unsigned long long flags[kMaxGroups];   // the tile's knot flags (flag_pos): the consumer of the tile (next launch, same
                                            // tile) reads them instead of re-running the predicate
};
static_assert(sizeof(TileRec) == 64 + 8 * kMaxGroups && sizeof(TileRec) % 16 == 0, "TileRec layout");
static_assert(ITD_TILE <= 1024, "10-bit knot positions");
constexpr int kRecLanes = sizeof(TileRec) / 16;   // 16-byte lanes that move one record
// dword offsets inside the first 64 bytes (what the staged copies in LDS are indexed with)
constexpr int kRecDwHval = 2, kRecDwTval = 8, kRecDwEdge = 12;
__host__ __device__ constexpr unsigned long long rec_pack(int count, int h0, int h1, int h2, int t0, int t1)
{
    return (unsigned long long)count | ((unsigned long long)h0 << 11) | ((unsigned long long)h1 << 21) |
           ((unsigned long long)h2 << 31) | ((unsigned long long)t0 << 41) | ((unsigned long long)t1 << 51);
}
__host__ __device__ constexpr int rec_count(unsigned long long p) { return (int)(p & 0x7ffull); }
__host__ __device__ constexpr int rec_hpos(unsigned long long p, int k) { return (int)((p >> (11 + 10 * k)) & 0x3ffull); }   // k = 0..2
__host__ __device__ constexpr int rec_tpos(unsigned long long p, int k) { return (int)((p >> (41 + 10 * k)) & 0x3ffull); }   // k = 0: second last, 1: last

// kCpp: the knot predicate of itd.cpp:161-168 (strict on the left, non-strict on the right); kZeroCross: the sign changes
// s[i] -> s[i+1] of find_extrema, itd_fourier_decomposition.py:23-27 (both feed the cubic variant, itd_cubic.hpp)
enum DetectMode : int { kKnots = 0, kValleys = 1, kPeaks = 2, kCpp = 3, kZeroCross = 4 };

// Tiles are grouped in runs of 64; every producer of a tile list adds its count to the group's sum (one
// atomic per non-empty tile).  Each group sum sits on a 128-byte line of its own: device-scope atomics on
// one line serialise (~3 ns each, measured).  Three buffers rotate by level: level j reads j%3, adds into
// (j+1)%3 and clears (j+2)%3.
constexpr int kTilesPerGroup = 64;
constexpr int kGsumPitch = 32;  // int32 elements between consecutive group sums
__host__ __device__ constexpr int groups_of(int n_tiles) { return (n_tiles + kTilesPerGroup - 1) / kTilesPerGroup; }

// compiler-level ordering point between wave-synchronous LDS phases (LDS executes one wave's accesses in
// program order; this only stops the compiler from moving accesses across the phase boundary)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// LDS tile: TW samples + 1 halo sample each side.  tile[pos + 2] <-> sample s + pos, pos in [-1, TW];
// the body starts 16-byte aligned so the staging stores are ds_write_b128.
// ---------------------------------------------------------------------------------------------
template <int TW>
struct Tile {
    // pos -1 .. TW are samples; TW+1 .. TW+5 are five extra slots (k_extract keeps the values of the knots
    // around the tile there, so a knot's value is one indexed read whether the knot is inside or outside)
    static constexpr int kSize = TW + 8;
    double *p;
    __device__ __forceinline__ double &at(int pos) const { return p[pos + 2]; }
};

// Tile fetch split in two (issue early / commit late) so the HBM latency overlaps the halo search.
template <typename Tin, int TW>
struct TileRegs {
    // per-lane vector of kPer consecutive samples: 16 bytes (double2 / float4) when the tile allows, else 8 bytes
    static constexpr int kPer = sizeof(Tin) == 8 ? 2 : (TW % 256 == 0 ? 4 : 2);
    static_assert(TW % (kPer * kWave) == 0, "whole vector loads per lane");
    using V = Tin __attribute__((ext_vector_type(kPer)));
    static constexpr int kN = TW / (kPer * kWave);
    static constexpr uintptr_t kAlignMask = sizeof(Tin) * kPer - 1;
    V q[kN];
    double lo, hi;
    bool vec;
};

template <bool NT>
__device__ __forceinline__ void stream_store(double *p, double v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <typename Tin, int TW, bool NT = false>
__device__ __forceinline__ void tile_fetch(TileRegs<Tin, TW> &r, const Tin *__restrict__ x, int64_t n, int64_t s)
{
    using R = TileRegs<Tin, TW>;
    const int lane = lane_id();
    const Tin *src = x + s;
    r.vec = (s + TW <= n) && ((reinterpret_cast<uintptr_t>(src) & R::kAlignMask) == 0);
    if (r.vec) {
        const typename R::V *v = reinterpret_cast<const typename R::V *>(src);
#pragma unroll
        for (int k = 0; k < R::kN; ++k) r.q[k] = NT ? __builtin_nontemporal_load(&v[lane + k * kWave]) : v[lane + k * kWave];
    }
    r.lo = 0.0;
    r.hi = 0.0;
    if (lane == 0 && s >= 1) r.lo = (double)x[s - 1];
    if (lane == 1 && s + TW < n) r.hi = (double)x[s + TW];
}

template <typename Tin, int TW>
__device__ __forceinline__ void tile_commit(const TileRegs<Tin, TW> &r, const Tin *__restrict__ x, int64_t n,
                                            int64_t s, Tile<TW> tile)
{
    using R = TileRegs<Tin, TW>;
    using D2 = __attribute__((ext_vector_type(2))) double;
    const int lane = lane_id();
    if (r.vec) {
#pragma unroll
        for (int k = 0; k < R::kN; ++k) {
            const int e = (lane + k * kWave) * R::kPer;
            if constexpr (sizeof(Tin) == 8) {
                *reinterpret_cast<D2 *>(&tile.p[2 + e]) = r.q[k];
            } else if constexpr (R::kPer == 4) {
                D2 a = {(double)r.q[k].x, (double)r.q[k].y};
                D2 b = {(double)r.q[k].z, (double)r.q[k].w};
                *reinterpret_cast<D2 *>(&tile.p[2 + e]) = a;
                *reinterpret_cast<D2 *>(&tile.p[4 + e]) = b;
            } else {
                D2 a = {(double)r.q[k].x, (double)r.q[k].y};
                *reinterpret_cast<D2 *>(&tile.p[2 + e]) = a;
            }
        }
    } else {
        for (int k = lane; k < TW; k += kWave) {
            const int64_t i = s + k;
            tile.at(k) = (i < n) ? (double)x[i] : 0.0;
        }
    }
    if (lane == 0) tile.at(-1) = r.lo;
    if (lane == 1) tile.at(TW) = r.hi;
}

// ---------------------------------------------------------------------------------------------
// Knot predicate + ordered compaction of one LDS tile into its per-tile list and record.
//   vil = x[i+1]-x[i], vix = x[i]-x[i-1];  valley: vil > 0 & vix <= 0  (detect_peaks(x),  ITD.py:59)
//                                          peak:   vil < 0 & vix >= 0  (detect_peaks(-x): the same test
//                                                                       on the exactly negated differences)
// First and last sample are never knots (ITD.py:70-73).  The wavefront walks the tile in 64-sample groups:
// __ballot gives the group's flag mask, popcounts give the ordered output slot.  Returns the knot count.
// ---------------------------------------------------------------------------------------------
// popcount of the bits of m below this lane, plus init (v_mbcnt_lo/hi: two VALU instructions)
__device__ __forceinline__ int mbcnt64(unsigned long long m, int init)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)init));
}

// bits lo..hi (inclusive) of a 64-bit word, empty when hi < lo
__device__ __forceinline__ unsigned long long bit_range(int lo, int hi)
{
    if (hi < lo || hi < 0 || lo > 63) return 0ull;
    lo = lo < 0 ? 0 : lo;
    const unsigned long long upto = hi >= 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull);
    return upto & ~((1ull << lo) - 1ull);
}

// The tile's flag words live in two VGPRs: lane g holds word g (one 64-bit word per 64-sample group).  Loops over
// the groups stay rolled (v_readlane with a scalar lane index), so code size and scalar-register
// pressure do not grow with the tile width.
struct WaveMasks {
    unsigned lo = 0, hi = 0;
    __device__ __forceinline__ unsigned long long get(int g) const
    {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, g) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)lo, g);
    }
    __device__ __forceinline__ void set(int g, unsigned long long m)
    {
        const bool mine = lane_id() == g;
        lo = mine ? (unsigned)m : lo;
        hi = mine ? (unsigned)(m >> 32) : hi;
    }
};

template <int TW>
__device__ __forceinline__ int scan_flags(Tile<TW> tile, int64_t s, int64_t n, int mode, WaveMasks &wm)
{
    constexpr int G = TW / 64;
    const int lane = lane_id();
    const bool edge = (s == 0) || (s + TW >= n - 1);
    int total = 0;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int pos = flag_pos(g, lane);
        const double xm = tile.at(pos - 1), x0 = tile.at(pos), xp = tile.at(pos + 1);
        const double vil = xp - x0;
        const double vix = x0 - xm;
        const bool valley = (vil > 0.0) && (vix <= 0.0);
        const bool peak = (vil < 0.0) && (vix >= 0.0);
        bool f = (mode == kKnots) ? (valley || peak) : (mode == kValleys ? valley : peak);
        if (mode == kCpp) f = ((xm < x0) && (x0 >= xp)) || ((xm > x0) && (x0 <= xp));
        if (mode == kZeroCross) f = ((x0 > 0.0) && (0.0 > xp)) || ((x0 < 0.0) && (0.0 < xp));
        // first and last sample are never knots (ITD.py:70-73), nothing beyond sample n-2: edge tiles only
        if (edge) f = f && (s + pos >= 1) && (s + pos <= n - 2);
        const unsigned long long mk = __ballot(f);
        wm.set(g, mk);
        total += __popcll(mk);
    }
    return total;
}

template <int TW>
__device__ __forceinline__ int record_from_masks(Tile<TW> tile, int64_t s, const WaveMasks &wm, int total,
                                                 int32_t *__restrict__ list, TileRec *__restrict__ rec, int32_t *rec_lds);

// producer side (LDS tile: level-0 scan, NaN-faithful path, API helpers): the tile's record for the next launch and
// (optionally) its ordered knot list.  The first three / last two knots are found with scalar bit scans of the
// wave-uniform flag words, merging the even and odd word of each 128-sample group in sample order.
template <int TW>
__device__ __forceinline__ int detect_tile(Tile<TW> tile, int64_t s, int64_t n, int mode,
                                           int32_t *__restrict__ list, TileRec *__restrict__ rec,
                                           int32_t *rec_lds /* sizeof(TileRec) bytes of wave-private LDS, 16-byte aligned */)
{
    WaveMasks wm;
    const int total = scan_flags<TW>(tile, s, n, mode, wm);
    return record_from_masks<TW>(tile, s, wm, total, list, rec, rec_lds);
}

// the record (and optional ordered list) of a tile whose knot flag words are known (lane g holds word g)
template <int TW>
__device__ __forceinline__ int record_from_masks(Tile<TW> tile, int64_t s, const WaveMasks &wm, int total,
                                                 int32_t *__restrict__ list, TileRec *__restrict__ rec, int32_t *rec_lds)
{
    constexpr int G = TW / 64;
    static_assert(G <= kMaxGroups, "tile too wide for the record's flag words");
    const int lane = lane_id();
    int h0 = 0, h1 = 0, h2 = 0, t0 = 0, t1 = 0;
    if (total > 0) {
        int k = 0;
#pragma unroll 1
        for (int g = 0; g < G / 2 && k < 3; ++g) {
            unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
            while ((E | O) && k < 3) {
                const int pe = E ? __ffsll((long long)E) - 1 : 64, po = O ? __ffsll((long long)O) - 1 : 64;
                int pos;
                if (pe <= po) { pos = 128 * g + 2 * pe; E &= E - 1; }
                else { pos = 128 * g + 2 * po + 1; O &= O - 1; }
                if (k == 0) h0 = pos; else if (k == 1) h1 = pos; else h2 = pos;
                ++k;
            }
        }
        k = 0;
#pragma unroll 1
        for (int g = G / 2 - 1; g >= 0 && k < 2; --g) {
            unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
            while ((E | O) && k < 2) {
                const int pe = E ? 63 - __clzll((long long)E) : -1, po = O ? 63 - __clzll((long long)O) : -1;
                int pos;
                if (po >= pe) { pos = 128 * g + 2 * po + 1; O &= ~(1ull << po); }
                else { pos = 128 * g + 2 * pe; E &= ~(1ull << pe); }
                if (k == 0) t1 = pos; else t0 = pos;
                ++k;
            }
        }
    }
    // assemble the record in LDS, then hand it to HBM with one store of 16-byte lanes (an empty tile: the neighbours' half only)
    TileRec *lrec = reinterpret_cast<TileRec *>(rec_lds);
    if (lane < 5) {
        const int pos = lane == 0 ? h0 : lane == 1 ? h1 : lane == 2 ? h2 : lane == 3 ? t0 : t1;
        const double v = tile.at(pos);
        if (lane < 3) lrec->hval[lane] = v;
        else          lrec->tval[lane - 3] = v;
    }
    if (lane == 5) lrec->packed = rec_pack(total, h0, h1, h2, t0, t1);
    if (lane == 6) lrec->edge[0] = tile.at(0);
    if (lane == 7) lrec->edge[1] = tile.at(TW - 1);
    if (lane < G) lrec->flags[lane] = ((unsigned long long)wm.hi << 32) | wm.lo;   // lane g holds word g
    wave_sync();
    if (lane < (total > 0 ? kRecLanes : 4)) {
        using I4 = __attribute__((ext_vector_type(4))) int;
        reinterpret_cast<I4 *>(rec)[lane] = reinterpret_cast<const I4 *>(rec_lds)[lane];
    }
    if (list) {
        int base = 0;
#pragma unroll 1
        for (int g = 0; g < G / 2; ++g) {
            const unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
            const int before = mbcnt64(O, mbcnt64(E, base));   // knots in front of the lane's even position
            const int bE = (int)((E >> lane) & 1ull);
            if (bE) list[before] = (int32_t)(s + 128 * g + 2 * lane);
            if ((O >> lane) & 1ull) list[before + bE] = (int32_t)(s + 128 * g + 2 * lane + 1);
            base += __popcll(E) + __popcll(O);
        }
    }
    return total;
}

// the four end samples of the level's input for the NEXT level (ITD.py:101-102), taken from an LDS tile
template <int TW>
__device__ __forceinline__ void publish_ends(Tile<TW> tile, int64_t s, int64_t n, double *ends)
{
    const int q = lane_id();
    if (q < 4) {
        const int64_t i = (q == 0) ? 0 : (q == 1 ? 1 : (q == 2 ? n - 2 : n - 1));
        if (i >= s && i < s + TW) ends[q] = tile.at((int)(i - s));
    }
}

// ---------------------------------------------------------------------------------------------
// k_detect: level-0 knot scan of the caller's signal (float32 or float64 input).
// grid = (n_tiles, batch), 64 threads.  lists: [batch][n_tiles][TW] int32; counts/recs: [batch][n_tiles].
// ---------------------------------------------------------------------------------------------
template <typename Tin, int TW>
__global__ __launch_bounds__(kWave) void k_detect(const Tin *__restrict__ x, int64_t x_stride, int64_t n,
                                                  int n_tiles, int mode, int32_t *__restrict__ lists,
                                                  int32_t *__restrict__ counts, TileRec *__restrict__ recs,
                                                  int32_t *__restrict__ gsum_out, SigState *__restrict__ state)
{
    __shared__ __attribute__((aligned(16))) double s_x[Tile<TW>::kSize];
    __shared__ __attribute__((aligned(16))) int32_t s_rec[sizeof(TileRec) / 4];
    const int t = blockIdx.x;
    const int sig = blockIdx.y;
    const int64_t s = (int64_t)t * TW;
    const Tin *xs = x + (int64_t)sig * x_stride;
    Tile<TW> tile{s_x};
    TileRegs<Tin, TW> regs;
    tile_fetch<Tin, TW>(regs, xs, n, s);
    tile_commit<Tin, TW>(regs, xs, n, s, tile);
    wave_sync();
    if (s == 0 || s + TW >= n - 2) publish_ends<TW>(tile, s, n, state[sig].ends[0]);
    {
        bool nan_in = false;
#pragma unroll
        for (int g = 0; g < TW / 64; ++g) {
            const double v = tile.at(g * 64 + lane_id());
            nan_in = nan_in || (v != v);
        }
        if (__any(nan_in) && lane_id() == 0) state[sig].in_nan = 1;
    }
    const size_t slot = (size_t)sig * n_tiles + t;
    const int total = detect_tile<TW>(tile, s, n, mode, lists ? lists + slot * TW : nullptr, recs + slot, s_rec);
    if (lane_id() == 0) {
        counts[slot] = total;
        if (total) atomicAdd(&gsum_out[((size_t)sig * groups_of(n_tiles) + t / kTilesPerGroup) * kGsumPitch], total);
    }
}

// ---------------------------------------------------------------------------------------------
// k_nan_level0: level 0 of a decomposition whose INPUT holds a NaN, the way the reference runs it (ITD.py:87-88 on top of
// :44-68): idx_max = detect_peaks(x) takes the NaN branch — differences that are NaN count as +inf, NaN samples and their two
// neighbours cannot be peaks, and x is overwritten in place, NaN -> +inf (:46-51, :64-68); idx_min = detect_peaks(-x) then sees a
// NaN-free array (plain rules on the mutated values), and so does everything after it.  This launch writes the mutated signal
// (float64) to `xm`, and the level-0 counts / records / group sums / end samples of the knot set
//     valleys under the NaN rules on the original values  U  peaks under the plain rules on the mutated values;
// the record-driven level-0 extraction then runs on `xm`.  The rules apply per signal, only if that signal holds a NaN
// (SigState::in_nan, kept from the run that found it): a signal without one gets the plain knot set.  Rare path: LDS tile,
// one wavefront per tile, like k_detect.
// The single-level helpers use it too, with `lists` for the ordered compaction and one of
//   kKnots    the extraction's set above (itd_baseline_extract, detect_knots);
//   kValleys  detect_peaks(x) alone: the valleys under the NaN rules (ITD.py:33-76);
//   kPeaks    matlab_detect_peaks(x) (numba_accelerated_itd.py:17-59): the same branch on the NEGATED differences
//             (dx = -dx at :29, NaN -> +inf after it) — not detect_peaks(-x), which would see -NaN samples.
// xm may be null when the mutated signal is not needed.
// ---------------------------------------------------------------------------------------------
template <typename Tin, int TW>
__global__ __launch_bounds__(kWave) void k_nan_level0(const Tin *__restrict__ x, int64_t x_stride, int64_t n, int n_tiles,
                                                      double *__restrict__ xm, int64_t xm_stride, int32_t *__restrict__ counts,
                                                      TileRec *__restrict__ recs, int32_t *__restrict__ gsum_out,
                                                      SigState *__restrict__ state, int mode = (int)kKnots,
                                                      int32_t *__restrict__ lists = nullptr)
{
    constexpr int G = TW / 64;
    __shared__ __attribute__((aligned(16))) double s_x[Tile<TW>::kSize];
    __shared__ __attribute__((aligned(16))) int32_t s_rec[sizeof(TileRec) / 4];
    const int t = blockIdx.x, sig = blockIdx.y, lane = lane_id();
    const int64_t s = (int64_t)t * TW;
    const Tin *xs = x + (int64_t)sig * x_stride;
    Tile<TW> tile{s_x};
    TileRegs<Tin, TW> regs;
    tile_fetch<Tin, TW>(regs, xs, n, s);
    tile_commit<Tin, TW>(regs, xs, n, s, tile);
    wave_sync();
    const bool sig_nan = state[sig].in_nan != 0;
    const double inf = __builtin_huge_val();
    const bool edge = (s == 0) || (s + TW >= n - 1);
    // valleys of the original values (detect_peaks(x), ITD.py:44-73)
    WaveMasks vm;
    int total = 0;
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int pos = flag_pos(g, lane);
        const double xa = tile.at(pos - 1), x0 = tile.at(pos), xp = tile.at(pos + 1);
        double vil = xp - x0, vix = x0 - xa;
        if (mode == (int)kPeaks) { vil = -vil; vix = -vix; }      // numba_accelerated_itd.py:29
        bool ok = true;
        if (sig_nan) {
            vil = vil != vil ? inf : vil;                         // :51
            vix = vix != vix ? inf : vix;
            ok = !(xa != xa || x0 != x0 || xp != xp);             // :64-68 (samples beyond the signal were staged as 0: never NaN)
        }
        bool f = (vil > 0.0) && (vix <= 0.0) && ok;               // :59
        if (edge) f = f && (s + pos >= 1) && (s + pos <= n - 2);  // :70-73
        vm.set(g, __ballot(f));
    }
    wave_sync();
    // the mutation (:50), in the staged tile (its two halo samples included) and in the float64 copy the extraction reads
    for (int k = lane - 1; k <= TW; k += kWave) {
        double v = tile.at(k);
        if (v != v) { v = inf; tile.at(k) = v; }
        if (xm && k >= 0 && k < TW && s + k < n) xm[(int64_t)sig * xm_stride + s + k] = v;
    }
    wave_sync();
    if (s == 0 || s + TW >= n - 2) publish_ends<TW>(tile, s, n, state[sig].ends[0]);
    // peaks of the mutated values (detect_peaks(-x): plain rules), merged with the valleys
    WaveMasks wm;
    if (mode == (int)kKnots) (void)scan_flags<TW>(tile, s, n, (int)kPeaks, wm);
    wm.lo |= vm.lo;
    wm.hi |= vm.hi;
#pragma unroll 1
    for (int g = 0; g < G; ++g) total += __popcll(wm.get(g));
    const size_t slot = (size_t)sig * n_tiles + t;
    (void)record_from_masks<TW>(tile, s, wm, total, lists ? lists + slot * TW : nullptr, recs + slot, s_rec);
    if (lane == 0) {
        counts[slot] = total;
        if (total) atomicAdd(&gsum_out[((size_t)sig * groups_of(n_tiles) + t / kTilesPerGroup) * kGsumPitch], total);
    }
}

// ---------------------------------------------------------------------------------------------
// k_compact (API helpers: detect_peaks / knot lists handed back to the caller):
// per-tile lists -> ordered knot array e[0..m+1], one wavefront per tile.  It sums the group totals in
// front of its group and the tile counts in front of it inside the group, then copies its list.
// ---------------------------------------------------------------------------------------------
template <int TW>
__global__ __launch_bounds__(kWave) void k_compact(const int32_t *__restrict__ lists,
                                                   const int32_t *__restrict__ counts,
                                                   const int32_t *__restrict__ gsum_in, int n_tiles, int64_t n,
                                                   int32_t *__restrict__ kidx, int64_t kidx_stride,
                                                   int32_t *__restrict__ total_out, const SigState *__restrict__ state,
                                                   int64_t tail_value /* e[m+1]; < 0: n-1 (ITD.py:98) */,
                                                   int32_t *__restrict__ tile_base_out = nullptr /* optional [n_tiles]: knots in front of each tile */,
                                                   int lead = 1 /* 1: e = [0, knots, tail] (ITD.py:95-98); 0: the knots alone */)
{
    const int sig = blockIdx.y;
    const int t = blockIdx.x;
    const int lane = lane_id();
    const int n_groups = groups_of(n_tiles);
    const int32_t *cnt = counts + (size_t)sig * n_tiles;
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
    const int g = t / kTilesPerGroup;
    int acc = 0;
    for (int k = lane; k < g; k += kWave) acc += gs[(size_t)k * kGsumPitch];
    if (lane < t - g * kTilesPerGroup) acc += cnt[g * kTilesPerGroup + lane];
    const int c = cnt[t];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    const int base = acc;
    if (tile_base_out && lane == 0) tile_base_out[(size_t)sig * n_tiles + t] = base;
    int32_t *e = kidx + (size_t)sig * kidx_stride;
    const int32_t *src = lists + ((size_t)sig * n_tiles + t) * TW;
    for (int j = lane; j < c; j += kWave) e[lead + base + j] = src[j];
    if (lane == 0 && t == n_tiles - 1) {
        const int m = base + c;
        if (lead) {
            e[0] = 0;                    // ITD.py:96
            e[m + 1] = (int32_t)(tail_value < 0 ? n - 1 : tail_value); // ITD.py:98
        }
        total_out[2 * sig] = m;
        total_out[2 * sig + 1] = state[sig].in_nan;   // set by k_detect, which has completed
    }
}

// knot totals of a batch without the ordered lists (count-only detection): one wavefront per signal sums its group sums;
// totals[2b] = count, totals[2b+1] = the signal holds a NaN (like k_compact).  grid = ceil(batch / 4), 256 threads.
__global__ __launch_bounds__(256) void k_batch_totals(const int32_t *__restrict__ gsum, int n_groups, int batch,
                                                      const SigState *__restrict__ state, int32_t *__restrict__ totals)
{
    const int sig = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (sig >= batch) return;
    const int32_t *gs = gsum + (size_t)sig * n_groups * kGsumPitch;
    int acc = 0;
    for (int k = lane; k < n_groups; k += kWave) acc += gs[(size_t)k * kGsumPitch];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) { totals[2 * sig] = acc; totals[2 * sig + 1] = state[sig].in_nan; }
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
    const int lane = lane_id();
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

// ---- scalars for a polling host: one 32-bit value and its call's number in ONE 8-byte store to host-mapped memory -------------------
// A few operators hand a handful of scalars back to a host that polls for them (itd_engine.hip: small_wait) instead of synchronising
// the stream.  Data words followed by a fence and a flag word are NOT enough on this path: stores of one wavefront to host memory can
// arrive out of order (a 5000-call fuzz of MEITD's host-driven loop saw the flag in front of the data about once per 100 000
// launches — a stale NaN flag or knot count, i.e. a wrong branch).  So every word validates itself: value in the low half, the call's
// number in the high half, written whole or not at all; the host takes a result only when all of its words carry the number it waits for.
__device__ __forceinline__ void small_put(unsigned long long *words, int i, uint32_t value, uint32_t tag)
{
    __hip_atomic_store(words + i, ((unsigned long long)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void small_put_f64(unsigned long long *words, int i, double v, uint32_t tag)     // two words: low, high
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    small_put(words, i, (uint32_t)b, tag);
    small_put(words, i + 1, (uint32_t)(b >> 32), tag);
}

// ---------------------------------------------------------------------------------------------
// k_extract (further down, after the helpers it shares with k_scan0): one extraction on one tile, one wavefront, the tile
// held in REGISTERS; grid = (ceil(n_tiles / KT), batch), 64 threads, KT = kTilesPerWave consecutive tiles per wavefront.
//   xin           level input (float32/float64 caller signal at level 0, float64 baseline afterwards)
//   counts/recs   per-tile knot counts and records, double buffered by level parity (neighbours read them)
//   gsum_in/out/clear   group sums of the counts, rotating by level % 3
//   rot_out       rotation row   (FINAL: rotation + baseline, the "Out of time!" row, ITD.py:420)
//   base_out      baseline row   (FINAL: not written)
// Lane l owns two consecutive samples of every 128-sample group g: tile positions 128 g + 2 l (even) and + 1 (odd) —
// exactly what one coalesced 16-byte load/store per lane moves.  Flag word 2g holds the even samples' knot flags of
// group g, word 2g+1 the odd ones (flag_pos).  Consequences:
//   * no LDS tile: LDS holds only the by-rank knot arrays and small staging buffers (4.3 KB per wavefront);
//   * a sample's left/right neighbours are its own lane's other sample or the adjacent lane's (DPP wave shifts):
//     the next level's knot scan runs on registers, sharing every difference between the two samples that use it;
//   * half as many group iterations, each moving 16 bytes per lane.
// Knot ranks are tile-relative (ke = knots at or before the lane's even sample), computed once; a pass (a run of groups
// whose knots fit the by-rank arrays) covers ranks [rb, rb+m): by-rank slot L = rank - rb + 2, slots 0,1 = the two knots
// in front of the run, m+2..m+4 = the three behind it (own knots of other groups, written by their owner lanes, or the
// tile's halo knots).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dbits(double v) { return __builtin_bit_cast(unsigned long long, v); }
__device__ __forceinline__ double bits_d(unsigned lo, unsigned hi)
{
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// RN(a / b) for integers 0 <= a, 0 < b < 2^31 — the knot spacings' ratio of ITD.py:107 (int64 differences, true-divided in float64).
// This IS the division the compiler emits (reciprocal estimate, two Newton steps, quotient, residual, correction: correctly rounded)
// without its range scaling and special-case fix-up (v_div_scale x 2, v_div_fmas' scale, v_div_fixup), which do nothing for operands
// that are exact small integers: four vector instructions less per ratio.  Held equal to the full division, bit for bit, over every
// pair with b <= 2048 and random larger ones by tests/test_gpu_parity.py (itd_debug_int_ratio_check).
__device__ __forceinline__ double int_ratio(int32_t a, int32_t b)
{
    const double x = (double)a, y = (double)b;
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    const double res = __builtin_fma(-y, q, x);
    return __builtin_fma(res, r, q);
}
// DPP wavefront shifts (gfx9): CTRL 0x138 wave_shr:1 (lane l <- l-1), 0x130 wave_shl:1 (l <- l+1),
// 0x13C wave_ror:1 (shr with lane 0 <- lane 63), 0x134 wave_rol:1 (shl with lane 63 <- lane 0).
// Lanes without a source lane keep `old`.
template <int CTRL>
__device__ __forceinline__ double wave_dpp(double old, double v)
{
    const unsigned long long o = dbits(old), u = dbits(v);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return bits_d((unsigned)lo, (unsigned)hi);
}
template <int CTRL>
__device__ __forceinline__ float wave_dpp(float old, float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// lane L of `old` replaced by a wave-uniform value (v_writelane_b32; the value comes from the scalar unit: no hazard to cover)
template <int L>
__device__ __forceinline__ unsigned write_lane(unsigned old, unsigned value)
{
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(value), "n"(L));
    return old;
}
// this lane's bit of a wave-uniform 64-bit mask: one v_cndmask with the mask as the select operand
__device__ __forceinline__ int lane_bit(unsigned long long mask)
{
    int r;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(mask));
    return r;
}

// ---------------------------------------------------------------------------------------------
// knot_predicate: the knot rule (ITD.py:59 on x and on -x, raw differences) over NG consecutive 128-sample groups as mask
// algebra.  Per group d0 = even - left, d1 = odd - even; the difference behind an odd sample is the next lane's d0 (the next
// group's lane 0 for lane 63; `dlast` behind the last position, only lane 63's value is used).  A position with the difference
// dp in front of it and dn behind it is a knot <=> (dn > 0 & dp <= 0) | (dn < 0 & dp >= 0).
// Every compare leaves the vector unit as a 64-bit lane mask; "the next lane's" masks are those masks shifted by one bit on the
// scalar unit, and while no difference is NaN (no NaN sample, no inf - inf: the common case, tested once per call)
// d > 0 <=> !(d <= 0) and d < 0 <=> !(d >= 0), so four ordered compares and one unordered compare per group do instead of
// eight ordered ones plus a DPP shift of the differences.  No exec-masked short circuits, no bool -> mask round trips.
// ---------------------------------------------------------------------------------------------
// zero_inner (optional): receives a non-zero word if any difference of the INNER groups 1 .. NG-2 is exactly zero (an exact tie of
// two neighbouring samples; NaN differences count too): the fused level-0 launch flags such tiles for the fused sparse levels
// (itd_knotfirst.hpp).  The masks exist anyway: two scalar operations per group.
template <typename T, int NG>
__device__ __forceinline__ void knot_predicate(const T (&d0)[NG], const T (&d1)[NG], T dlast,
                                               unsigned long long (&E)[NG], unsigned long long (&O)[NG],
                                               unsigned long long *zero_inner = nullptr)
{
    using ull = unsigned long long;
    ull zacc = 0;
    constexpr ull kTop = 1ull << 63;
    ull unord = __ballot(dlast != dlast) & kTop;
#pragma unroll
    for (int g = 0; g < NG; ++g) unord |= __ballot(__builtin_isunordered(d0[g], d1[g]));
    const ull plast = __ballot(dlast > (T)0) & kTop, nlast = __ballot(dlast < (T)0) & kTop;
    // group by group, so that only two groups' masks are alive at a time (scalar registers are scarce in the callers)
    if (__builtin_expect(unord == 0, 1)) {
        ull le = __ballot(d0[0] <= (T)0), ge = __ballot(d0[0] >= (T)0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gn = g + 1 < NG ? g + 1 : g;
            const ull le_n = __ballot(d0[gn] <= (T)0), ge_n = __ballot(d0[gn] >= (T)0);   // (unused for the last group)
            const ull p1 = __ballot(d1[g] > (T)0), n1 = __ballot(d1[g] < (T)0);
            const ull p2 = (~le >> 1) | (g + 1 < NG ? ~le_n << 63 : plast);
            const ull n2 = (~ge >> 1) | (g + 1 < NG ? ~ge_n << 63 : nlast);
            E[g] = (p1 & le) | (n1 & ge);
            O[g] = (p2 & ~p1) | (n2 & ~n1);
            if (g >= 1 && g + 2 <= NG) zacc |= (le & ge) | ~(p1 | n1);   // d0 == 0 / d1 == 0
            le = le_n;
            ge = ge_n;
        }
    } else {   // some difference is NaN: all eight ordered compares
        zacc = ~0ull;
        ull p0 = __ballot(d0[0] > (T)0), n0 = __ballot(d0[0] < (T)0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int gn = g + 1 < NG ? g + 1 : g;
            const ull p0_n = __ballot(d0[gn] > (T)0), n0_n = __ballot(d0[gn] < (T)0);
            const ull le0 = __ballot(d0[g] <= (T)0), ge0 = __ballot(d0[g] >= (T)0);
            const ull p1 = __ballot(d1[g] > (T)0), n1 = __ballot(d1[g] < (T)0);
            const ull le1 = __ballot(d1[g] <= (T)0), ge1 = __ballot(d1[g] >= (T)0);
            const ull p2 = (p0 >> 1) | (g + 1 < NG ? p0_n << 63 : plast);
            const ull n2 = (n0 >> 1) | (g + 1 < NG ? n0_n << 63 : nlast);
            E[g] = (p1 & le0) | (n1 & ge0);
            O[g] = (p2 & le1) | (n2 & ge1);
            p0 = p0_n;
            n0 = n0_n;
        }
    }
    if (zero_inner) *zero_inner = zacc;
}

// ---------------------------------------------------------------------------------------------
// scan_publish: knot scan of a tile held in registers (xr[g][0/1] = positions 128 g + 2 lane / + 1) and everything the
// next launch needs from it: count, group sum, the 128-byte record.  x_lo / x_hi are samples s-1 and s+TW (wave-uniform).
// Differences: d0 = even - left, d1 = odd - even, d2 = right - odd (= the next lane's d0): every difference is computed
// once and shared by the two samples whose predicate uses it (ITD.py:59 on raw differences).
// ---------------------------------------------------------------------------------------------
// Two neighbouring values that rounding may make (or has made) equal within the levels to come: the fused sparse levels
// (itd_knotfirst.hpp) keep both samples of such a pair as candidates.  |a - b| <= 2^-20 max(|a|, |b|): differences shrink by a few
// bits per level, a pair 2^-27 apart at level 3 was an exact tie at level 8 (signal 51 of the bench's batch recipe).
constexpr double kNearTie = 0x1p-20;
__device__ __forceinline__ bool near_tie(double a, double b)
{
    return __builtin_fabs(a - b) <= kNearTie * __builtin_fmax(__builtin_fabs(a), __builtin_fabs(b));
}

template <int TW, bool COUNT_ONLY = false, bool NEARF = false>
__device__ __forceinline__ int scan_publish(const double (&xr)[TW / 128][2], double x_lo, double x_hi, int64_t s, int nrem,
                                            size_t slot, size_t gsum_index, int32_t *__restrict__ counts_out,
                                            TileRec *__restrict__ recs_out, int32_t *__restrict__ gsum_out,
                                            int32_t *s_rec /* sizeof(TileRec) bytes, 16-byte aligned */, int32_t *s_pos /* 8 ints */,
                                            int count_hi = 0 /* added to the int behind the group sum by the same (64-bit) atomic:
                                                                the fused level-0 launch's own knot count, for m[0] */,
                                            unsigned long long *near_words = nullptr /* NEARF: the tile's 2 TW / 128 words (flag_pos format) that
                                                                mark both samples of every near tie of the scanned row (itd_knotfirst.hpp) */)
{
    constexpr int G2 = TW / 128;
    const int lane = lane_id();
    unsigned long long E[G2], O[G2];   // flag words of the even / odd positions of group g (record words 2g, 2g+1)
    int total = 0;
    int gcount[G2];
    if (ITD_ABL_R & 4096) {   // timing skeleton: no scan, a constant non-zero count so that no level stops
        total = 3;
#pragma unroll
        for (int g = 0; g < G2; ++g) { gcount[g] = 0; E[g] = O[g] = 0; }
    } else {
        double d0[G2], d1[G2];
        unsigned long long nt0[G2], nt1[G2];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const double fill = (g == 0) ? x_lo : wave_dpp<0x13C>(0.0, xr[g > 0 ? g - 1 : 0][1]);   // lane 0 <- lane 63 of the previous group
            const double left = wave_dpp<0x138>(fill, xr[g][1]);
            nt0[g] = nt1[g] = 0ull;
            d0[g] = xr[g][0] - left;
            d1[g] = xr[g][1] - xr[g][0];
            if constexpr (NEARF) {   // the launch in front of the fused sparse levels: near ties of the row just written
                nt0[g] = __ballot(__builtin_fabs(d0[g]) <= kNearTie * __builtin_fmax(__builtin_fabs(left), __builtin_fabs(xr[g][0])));      // (left neighbour, even sample)
                nt1[g] = __ballot(__builtin_fabs(d1[g]) <= kNearTie * __builtin_fmax(__builtin_fabs(xr[g][0]), __builtin_fabs(xr[g][1])));  // (even sample, odd sample)
            }
        }
        knot_predicate<double, G2>(d0, d1, x_hi - xr[G2 - 1][1], E, O);
        if constexpr (NEARF) {
            // both samples of a near pair: an even sample with either neighbour, an odd sample with its even one or with the next
            // lane's even one (the next group's first / the next tile's first sample for lane 63); positions 1 .. n-2 only
            const unsigned long long nlast = __ballot(near_tie(xr[G2 - 1][1], x_hi)) & (1ull << 63);
            unsigned long long mine = 0ull;
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                unsigned long long Es = nt0[g] | nt1[g];
                unsigned long long Os = nt1[g] | (nt0[g] >> 1) | (g + 1 < G2 ? nt0[g + 1 < G2 ? g + 1 : g] << 63 : nlast);
                const int lo = s > 0 ? 0 : 1, hi = nrem - 2;
                Es &= bit_range((lo - 128 * g + 1) >> 1, (hi - 128 * g) >> 1);
                Os &= bit_range((lo - 128 * g) >> 1, (hi - 128 * g - 1) >> 1);
                if (lane == 2 * g) mine = Es;
                if (lane == 2 * g + 1) mine = Os;
            }
            if (lane < 2 * G2 && near_words) near_words[lane] = mine;
        }
        if ((s == 0) || (nrem <= TW + 1)) {   // the tile holds sample 0, or sample n-2 or later: the first and the last sample are
            const int lo = s > 0 ? 0 : 1, hi = nrem - 2;   // never knots (ITD.py:70-73), nothing beyond sample n-2
#pragma unroll
            for (int g = 0; g < G2; ++g) {   // even position 128 g + 2 l, odd position 128 g + 2 l + 1 within [lo, hi]
                E[g] &= bit_range((lo - 128 * g + 1) >> 1, (hi - 128 * g) >> 1);
                O[g] &= bit_range((lo - 128 * g) >> 1, (hi - 128 * g - 1) >> 1);
            }
        }
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            gcount[g] = __popcll(E[g]) + __popcll(O[g]);
            total += gcount[g];
        }
    }
    // count and group sum first (a wavefront cannot retire before its last store is acknowledged)
    if (lane == 0) {
        if (!(ITD_ABL_R & 32)) counts_out[slot] = total;
        if (count_hi)   // the group sums are 128 bytes apart: element +1 is free, and 8-byte aligned with element 0
            atomicAdd(reinterpret_cast<unsigned long long *>(&gsum_out[gsum_index]), ((unsigned long long)(unsigned)count_hi << 32) | (unsigned)total);
        else if (total && !(ITD_ABL_R & 16)) atomicAdd(&gsum_out[gsum_index], total);
    }
    if constexpr (COUNT_ONLY) return total;   // the "Out of time!" launch: only the stop test reads this level's knots
    // the record: first three / last two knots, written by their owner lanes (rank = knots before the sample); the tile's
    // first and last sample for the neighbours
    TileRec *lrec = reinterpret_cast<TileRec *>(s_rec);
    if (lane < 5) s_pos[lane] = 0;
    if (lane == 0) lrec->edge[0] = xr[0][0];
    if (lane == 63) lrec->edge[1] = xr[G2 - 1][1];
    if (total > 0) {
        int gbase = 0;
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            if (gcount[g] > 0 && (gbase < 3 || gbase + gcount[g] > total - 2)) {
                const int bE = lane_bit(E[g]), bO = lane_bit(O[g]);
                const int re = mbcnt64(O[g], mbcnt64(E[g], gbase));   // knots before the even sample
                const int ro_ = re + bE;                        // knots before the odd sample
                const int pe = 128 * g + 2 * lane;
                if (bE) {
                    if (re < 3) { s_pos[re] = pe; lrec->hval[re] = xr[g][0]; }
                    if (re >= total - 2) { s_pos[3 + re - (total - 2)] = pe; lrec->tval[re - (total - 2)] = xr[g][0]; }
                }
                if (bO) {
                    if (ro_ < 3) { s_pos[ro_] = pe + 1; lrec->hval[ro_] = xr[g][1]; }
                    if (ro_ >= total - 2) { s_pos[3 + ro_ - (total - 2)] = pe + 1; lrec->tval[ro_ - (total - 2)] = xr[g][1]; }
                }
            }
            gbase += gcount[g];
        }
        unsigned flo = 0, fhi = 0;   // lane j < 2 * G2 gets word j (v_writelane from the scalar registers)
        auto put = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g < G2) {
                flo = write_lane<2 * g>(flo, (unsigned)E[g]);
                fhi = write_lane<2 * g>(fhi, (unsigned)(E[g] >> 32));
                flo = write_lane<2 * g + 1>(flo, (unsigned)O[g]);
                fhi = write_lane<2 * g + 1>(fhi, (unsigned)(O[g] >> 32));
            }
        };
        put(std::integral_constant<int, 0>{}); put(std::integral_constant<int, 1>{}); put(std::integral_constant<int, 2>{});
        put(std::integral_constant<int, 3>{}); put(std::integral_constant<int, 4>{}); put(std::integral_constant<int, 5>{});
        put(std::integral_constant<int, 6>{}); put(std::integral_constant<int, 7>{});
        static_assert(G2 <= 8, "flag words per record");
        if (lane < 2 * G2) lrec->flags[lane] = ((unsigned long long)fhi << 32) | flo;
    }
    wave_sync();
    if (lane == 0) lrec->packed = rec_pack(total, s_pos[0], s_pos[1], s_pos[2], s_pos[3], s_pos[4]);
    wave_sync();
    if (lane < (total > 0 ? kRecLanes : 4) && !(ITD_ABL_R & 64)) {   // 16-byte lanes; an empty tile: the neighbours' half only
        using I4 = __attribute__((ext_vector_type(4))) int;
        reinterpret_cast<I4 *>(recs_out + slot)[lane] = reinterpret_cast<const I4 *>(s_rec)[lane];
    }
    return total;
}

// ---------------------------------------------------------------------------------------------
// knot_masks: the knot predicate (ITD.py:59 on x and on -x) over NG consecutive 128-sample groups held in registers
// (v[g][0/1] = positions 128 g + 2 lane / + 1 of the window), evaluated in the samples' own precision T.  For float input
// this is exact: the float32 difference of two float32 values is zero, positive, negative or NaN exactly when the float64
// difference of the widened values is (subtraction is correctly rounded and float32 denormals are not flushed), so the
// flags equal those of the reference's float64 differences.
// The window's first and last position have an unknown neighbour: their flags are cleared (callers overlap windows).
// `lo`, `hi`: window-relative positions allowed to flag (the signal's first/last-sample rule ITD.py:70-73, clipping).
// ---------------------------------------------------------------------------------------------
template <typename T, int NG>
__device__ __forceinline__ void knot_masks(const T (&v)[NG][2], int lo, int hi, unsigned long long (&E)[NG], unsigned long long (&O)[NG],
                                           unsigned long long *zero_inner = nullptr)
{
    T d0[NG], d1[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const T fill = (g == 0) ? v[0][0] : wave_dpp<0x13C>((T)0, v[g > 0 ? g - 1 : 0][1]);   // lane 0 <- lane 63 of the previous group
        const T left = wave_dpp<0x138>(fill, v[g][1]);
        d0[g] = v[g][0] - left;
        d1[g] = v[g][1] - v[g][0];
    }
    knot_predicate<T, NG>(d0, d1, (T)0, E, O, zero_inner);   // last window position: right neighbour unknown (a zero difference never flags)
    E[0] &= ~1ull;                                // window position 0: left neighbour unknown
    if (lo > 1 || hi < 128 * NG - 2) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {   // wave-uniform lane ranges: even position 128 g + 2 l in [lo, hi], odd position 128 g + 2 l + 1 in [lo, hi]
            E[g] &= bit_range((lo - 128 * g + 1) >> 1, (hi - 128 * g) >> 1);
            O[g] &= bit_range((lo - 128 * g) >> 1, (hi - 128 * g - 1) >> 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// nan_rules: the reference's stop test on a baseline that holds non-finite values (rare).  detect_peaks(b) takes its NaN
// branch when b holds a NaN anywhere (ITD.py:46-51, 64-68): differences that are NaN count as +inf, NaN samples and their two
// neighbours cannot be peaks, and b is mutated in place, NaN -> +inf (ITD.py:50; there is no copy at ITD.py:41).
// detect_peaks(-b) (ITD.py:401) and the next extraction (ITD.py:87-88) then see a NaN-free array: plain rules on the mutated
// values.  For a tile in registers (xr: the baseline just produced, x_lo / x_hi: its samples s-1 and s+TW) this returns
//     (valleys under the NaN-branch rules on the original values) - (valleys under the plain rules on the mutated values)
// = what must be added to scan_publish's plain knot count of the mutated tile to get the reference's num_extrema, and mutates
// xr, x_lo, x_hi.  Only tiles that hold a NaN or an infinity can differ (finite - finite is never NaN), so callers skip all
// others; the level's sum is used only if some tile of the level saw a NaN (SigState::nan_mask) — the reference's
// `if indl.size != 0`.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool nonfinite(double v) { return __builtin_amdgcn_class(v, 0x207); }   // sNaN | qNaN | -inf | +inf
template <int TW>
__device__ __forceinline__ int nan_rules(double (&xr)[TW / 128][2], double &x_lo, double &x_hi, int64_t s, int nrem)
{
    constexpr int G2 = TW / 128;
    const int lane = lane_id();
    const double inf = __builtin_huge_val();
    const bool edge = (s == 0) || (nrem <= TW + 1);
    auto valleys = [&](bool nan_branch) {
        double d0[G2];
        unsigned long long NE[G2], NO[G2];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const double fill = (g == 0) ? x_lo : wave_dpp<0x13C>(0.0, xr[g > 0 ? g - 1 : 0][1]);
            const double left = wave_dpp<0x138>(fill, xr[g][1]);
            d0[g] = xr[g][0] - left;
            NE[g] = __ballot(xr[g][0] != xr[g][0]);
            NO[g] = __ballot(xr[g][1] != xr[g][1]);
        }
        int cnt = 0;
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            double a0 = d0[g];
            double a1 = xr[g][1] - xr[g][0];
            const double fill = (g == G2 - 1) ? (x_hi - xr[g][1]) : wave_dpp<0x134>(0.0, d0[g < G2 - 1 ? g + 1 : g]);
            double a2 = wave_dpp<0x130>(fill, d0[g]);
            if (nan_branch) {   // ITD.py:51
                a0 = a0 != a0 ? inf : a0;
                a1 = a1 != a1 ? inf : a1;
                a2 = a2 != a2 ? inf : a2;
            }
            bool fe = (a1 > 0.0) && (a0 <= 0.0);   // ITD.py:59
            bool fo = (a2 > 0.0) && (a1 <= 0.0);
            if (edge) {   // ITD.py:70-73
                const int p = 128 * g + 2 * lane;
                fe = fe && (s > 0 || p >= 1) && (p <= nrem - 2);
                fo = fo && (p + 1 <= nrem - 2);
            }
            unsigned long long VE = __ballot(fe), VO = __ballot(fo);
            if (nan_branch) {   // ITD.py:64-68: NaN samples and their neighbours cannot be peaks
                const unsigned long long cin = (g == 0) ? (__ballot(x_lo != x_lo) ? 1ull : 0ull) : (NO[g > 0 ? g - 1 : 0] >> 63);
                const unsigned long long cout = (g == G2 - 1) ? (__ballot(x_hi != x_hi) ? 1ull : 0ull) : (NE[g < G2 - 1 ? g + 1 : g] & 1ull);
                VE &= ~(NE[g] | NO[g] | (NO[g] << 1) | cin);
                VO &= ~(NE[g] | NO[g] | (NE[g] >> 1) | (cout << 63));
            }
            cnt += __popcll(VE) + __popcll(VO);
        }
        return cnt;
    };
    const int v_nan = valleys(true);
#pragma unroll
    for (int g = 0; g < G2; ++g) {   // ITD.py:50
        xr[g][0] = xr[g][0] != xr[g][0] ? inf : xr[g][0];
        xr[g][1] = xr[g][1] != xr[g][1] ? inf : xr[g][1];
    }
    x_lo = x_lo != x_lo ? inf : x_lo;
    x_hi = x_hi != x_hi ? inf : x_hi;
    return v_nan - valleys(false);
}

// ---------------------------------------------------------------------------------------------
// Workgroup -> work item (a wavefront's run of tiles), XCD aware: consecutive workgroup ids are dealt round-robin to the 8
// XCDs, each with an L2 of its own.  With the identity mapping an XCD sees every 8th 4-KB tile — a 32-KB address stride that
// uses only part of its L2 channels — and reads its neighbours' records and counts through the fabric.  Here every span of
// 8 x ITD_XCD_CHUNK workgroups hands each XCD ITD_XCD_CHUNK CONSECUTIVE items (a shorter last span: an eighth of what is
// left each); the same mapping in every launch keeps a tile, its records and its neighbours on one XCD.  Measured (2^24
// samples, levels >= 1): identity 66.8 us, chunks of 8 / 64 / 256 / 1024 items: 64.2 / 63.0 / 61.9 / 62.4 us.
// ---------------------------------------------------------------------------------------------
#ifndef ITD_XCD_CHUNK
#define ITD_XCD_CHUNK 256   // 0: identity
#endif
__device__ __forceinline__ int xcd_item(int b, int n_items)
{
    constexpr int kXcd = 8, kChunk = ITD_XCD_CHUNK, kSpan = kXcd * (kChunk > 0 ? kChunk : 1);
    if (kChunk <= 0) return b;
    const int base = (b / kSpan) * kSpan;                 // start of b's span
    const int c = min(kSpan, n_items - base) / kXcd;      // items per XCD in this span (kChunk in a full span)
    const int r = b - base;
    if (r >= kXcd * c) return b;                          // the last < 8 items
    return base + (r % kXcd) * c + r / kXcd;
}

// ITD_SERPENTINE=1: odd levels walk the signal backwards, so a launch starts with the tiles the previous one wrote last
// (still the hottest lines of the Infinity Cache): step 0.626 -> 0.615 ms, level 0 79 -> 73 us (same-box A/B)
#ifndef ITD_SERPENTINE
#define ITD_SERPENTINE 1
#endif
__device__ __forceinline__ int launch_item(int b, int n_items, int level)
{
    const int it = xcd_item(b, n_items);
    return (ITD_SERPENTINE && (level & 1)) ? n_items - 1 - it : it;
}

// ---------------------------------------------------------------------------------------------
// Bounds-checked streaming accesses of one tile through a raw buffer descriptor (base = the tile's first element, extent =
// what is left of the row): elements beyond the row read as 0 and are not written, any 4-byte alignment is legal.  One code
// path for full, ragged and unaligned tiles, and every access is unconditional — so the compiler can count them and wait for
// exactly the load it needs (vmcnt) instead of for all of them.
// ---------------------------------------------------------------------------------------------
#ifndef ITD_BUF_AUX
#define ITD_BUF_AUX 2            // A/B builds: the aux bits of the streaming accesses (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
constexpr int kBufNT = ITD_BUF_AUX;   // aux bits of the raw buffer builtins on gfx94x/gfx950: 2 = nt
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void *base, int64_t bytes_left)
{
    const int64_t lim = bytes_left < 0 ? 0 : (bytes_left > 0x7fffffffll ? 0x7fffffffll : bytes_left);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)lim, 0x00020000);
}
// the same with 32-bit arithmetic: `elems` elements of `elem_bytes` bytes (<= 0 elements: an empty descriptor, loads read 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc32(const void *base, int elems, unsigned elem_bytes)
{
    const unsigned e = elems < 0 ? 0u : (unsigned)elems;
    const unsigned lim = e > 0x7fffffffu / elem_bytes ? 0x7fffffffu : e * elem_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)lim, 0x00020000);
}
template <bool NT>
__device__ __forceinline__ void tile_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff, double &a, double &b)
{
    using U4 = unsigned __attribute__((ext_vector_type(4)));
    using D2v = double __attribute__((ext_vector_type(2)));
    const U4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NT ? kBufNT : 0);
    const D2v d = __builtin_bit_cast(D2v, v);
    a = d.x;
    b = d.y;
}
template <bool NT>
__device__ __forceinline__ void tile_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff, float &a, float &b)
{
    using U2 = unsigned __attribute__((ext_vector_type(2)));
    using F2v = float __attribute__((ext_vector_type(2)));
    const U2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, NT ? kBufNT : 0);
    const F2v d = __builtin_bit_cast(F2v, v);
    a = d.x;
    b = d.y;
}
template <bool NT>
__device__ __forceinline__ void tile_store2(__amdgpu_buffer_rsrc_t r, int voff, double a, double b)
{
    using U4 = unsigned __attribute__((ext_vector_type(4)));
    using D2v = double __attribute__((ext_vector_type(2)));
    const D2v d = {a, b};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(U4, d), r, voff, 0, NT ? kBufNT : 0);
}

// ---------------------------------------------------------------------------------------------
// k_scan0: level-0 knot scan of the caller's signal for a decomposition (the register form of k_detect: knots only, no
// lists).  grid = (ceil(n_tiles / KT), batch), 64 threads: a wavefront takes KT consecutive tiles and requests all of them
// up front.  The sample next to a tile
// is the neighbouring tile's edge sample, already in a register, except at the two ends of the wavefront's range.  Also
// publishes the signal's four end samples and the NaN-input flag.  Loads are cacheable: level 0 reads the signal again.
// ---------------------------------------------------------------------------------------------
template <typename Tin, int TW, int KT>
__global__ __launch_bounds__(kWave) void k_scan0(const Tin *__restrict__ xin, int64_t x_stride, int64_t n, int n_tiles,
                                                 int32_t *__restrict__ counts, TileRec *__restrict__ recs,
                                                 int32_t *__restrict__ gsum_out, SigState *__restrict__ state, int level)
{
    // level > 0: the scan of a stored baseline as the input of level `level` (the fused sparse levels capped below the call's last level:
    // the level launches go on from what the sample pass left — itd_engine.hip); the end samples go to that level's slot, and a NaN is
    // not the caller's (the fused levels refuse non-finite data: the call is repeated)
    constexpr int G2 = TW / 128;
    __shared__ __attribute__((aligned(16))) int32_t s_rec[sizeof(TileRec) / 4];
    __shared__ int32_t s_pos[8];
    const int lane = lane_id();
    const int t0 = xcd_item(blockIdx.x, gridDim.x) * KT, sig = blockIdx.y;
    const int es = level & 1;
    const Tin *x = xin + (int64_t)sig * x_stride;
    SigState *st = state + sig;
    if (st->skip > 0) return;       // (the device-side repair: this signal's first result stands)
    Tin q[KT][G2][2];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t s = (int64_t)(t0 + k) * TW;
        const __amdgpu_buffer_rsrc_t rx = tile_rsrc(x + s, (n - s) * (int64_t)sizeof(Tin));   // empty beyond the signal: reads 0
#pragma unroll
        for (int g = 0; g < G2; ++g) tile_load2<false>(rx, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), q[k][g][0], q[k][g][1]);
    }
    const int64_t s_first = (int64_t)t0 * TW, s_end = (int64_t)(t0 + KT) * TW;
    double x_before = 0.0, x_after = 0.0;   // the samples next to the wavefront's range (wave-uniform addresses)
    if (s_first >= 1) x_before = (double)x[s_first - 1];
    if (s_end < n) x_after = (double)x[s_end];
    bool nan_in = false;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int t = t0 + k;
        if (t >= n_tiles) break;
        const int64_t s = (int64_t)t * TW;
        double xr[G2][2];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            xr[g][0] = (double)q[k][g][0];
            xr[g][1] = (double)q[k][g][1];
        }
        // samples s-1 and s+TW: the neighbouring tile's last / first sample (lane 63 / lane 0 of its registers)
        double xlo = x_before, xhi = x_after;
        if (k > 0) {
            const double v = (double)q[k > 0 ? k - 1 : 0][G2 - 1][1];
            xlo = bits_d((unsigned)__builtin_amdgcn_readlane((int)(unsigned)dbits(v), 63), (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(dbits(v) >> 32), 63));
        }
        if (k + 1 < KT) {
            const double v = (double)q[k + 1 < KT ? k + 1 : k][0][0];
            xhi = bits_d((unsigned)__builtin_amdgcn_readlane((int)(unsigned)dbits(v), 0), (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(dbits(v) >> 32), 0));
        }
        const int nrem = (int)min((int64_t)(n - s), (int64_t)(TW + 2));
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const int p = 128 * g + 2 * lane;
            nan_in = nan_in || (p < nrem && xr[g][0] != xr[g][0]) || (p + 1 < nrem && xr[g][1] != xr[g][1]);
        }
        if (s == 0 || nrem <= TW + 1) {   // the signal's end samples x[0], x[1], x[n-2], x[n-1] (ITD.py:101-102)
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                const int p = 128 * g + 2 * lane;
                if (s == 0 && p == 0) { st->ends[es][0] = xr[g][0]; st->ends[es][1] = xr[g][1]; }
                if (p == nrem - 2) { st->ends[es][2] = xr[g][0]; st->ends[es][3] = xr[g][1]; }
                if (p + 1 == nrem - 2) st->ends[es][2] = xr[g][1];
                if (p == nrem - 1) st->ends[es][3] = xr[g][0];
            }
        }
        scan_publish<TW>(xr, xlo, xhi, s, nrem, (size_t)sig * n_tiles + t,
                         ((size_t)sig * groups_of(n_tiles) + t / kTilesPerGroup) * kGsumPitch, counts, recs, gsum_out, s_rec, s_pos);
        wave_sync();   // the next tile reuses the record staging
    }
    if (level == 0 && __any(nan_in) && lane == 0) st->in_nan = 1;
}

// the group sums of `levels` consecutive rotating buffers cleared for the signals of a launch sequence (in front of a k_scan0 that starts
// the level launches in the middle of a decomposition: the buffers hold earlier levels' sums)
__global__ void k_clear_gsum(int32_t *a, int32_t *b, int64_t elems)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < elems; i += (int64_t)gridDim.x * blockDim.x) { a[i] = 0; b[i] = 0; }
}


// FUSE0 (level 0 only): the launch reads no records at all.  The wavefront loads one 128-sample group in front of and one
// behind its tile with the tile, evaluates the knot predicate of the caller's signal itself (knot_masks, in the signal's
// own precision) and takes the two knots in front / three behind from those halo groups; only if a halo group holds too
// few does it walk on through the signal in overlapping 512-sample windows (at most kReach per side), and if even that
// finds too few away from the signal's ends it raises SigState::l0_fail — the engine then repeats the level-0 step with
// k_scan0 + the record-driven launch (itd_engine.hip).  The signal is read once: the separate scan pass disappears.
constexpr int kReach = 8;   // extension windows per side (510 new samples each)
// ITD_PROF=1 (diagnostic build, tools/level0_prof.py): wall-clock of a wavefront's phases in the fused level-0 launch, s_memtime
// at the phase boundaries accumulated in scalar registers, one 128-byte row per wavefront at the end (no atomics)
#ifndef ITD_PROF
#define ITD_PROF 0
#endif
#if ITD_PROF
__device__ unsigned long long *g_prof_buf;   // [wavefronts][16]
#define PROF_MARK(i)                                                   \
    if constexpr (FUSE0) {                                             \
        const unsigned long long now__ = __builtin_readcyclecounter(); \
        pt##i += now__ - pt_last;                                      \
        pt_last = now__;                                               \
    }
#else
#define PROF_MARK(i)
#endif
template <typename Tin, int TW, bool FINAL, int CAP, int KT, bool FUSE0 = false, bool TIES = false>
__global__ __launch_bounds__(kWave) void k_extract(const Tin *__restrict__ xin, int64_t x_stride, int64_t n,
                                                     int n_tiles, int batch,
                                                     const int32_t *__restrict__ counts_in,
                                                     int32_t *__restrict__ counts_out,
                                                     const TileRec *__restrict__ recs_in,
                                                     TileRec *__restrict__ recs_out,
                                                     const int32_t *__restrict__ gsum_in,
                                                     int32_t *__restrict__ gsum_out, int32_t *__restrict__ gsum_clear,
                                                     double *__restrict__ rot_out, int64_t rot_stride,
                                                     double *__restrict__ base_out, int64_t base_stride,
                                                     SigState *__restrict__ state, int level,
                                                     int keep_nan = 0 /* single-level helper: store the baseline as computed (the NaN -> +inf
                                                                         write, ITD.py:50, belongs to the driver's stop test) */,
                                                     unsigned long long *__restrict__ near_out = nullptr /* TIES, [batch][n_tiles][8]: flag
                                                                         words of both samples of every near tie of the baseline this launch
                                                                         writes (the launch in front of the fused sparse levels) */)
{
    constexpr int G2 = TW / 128;   // 128-sample groups; flag words 2g (even samples) and 2g+1 (odd samples)
    static_assert(TW % 128 == 0 && 2 * G2 <= kMaxGroups, "tile geometry");
    static_assert(CAP >= 128, "a pass must be able to take one 128-sample group");
    static_assert(!FUSE0 || !FINAL, "the fused level-0 launch is never the last level");
    constexpr int RK = CAP + 8;
    __shared__ double s_X[RK];      // value of the level's input at knot slot L
    __shared__ double s_B[RK];      // knot value B_L
    __shared__ double s_S[RK];      // slope of the segment that starts at slot L; before the slopes exist: the knots' indices
    __shared__ __attribute__((aligned(16))) int32_t s_rec[sizeof(TileRec) / 4];
    __shared__ double s_bl[2];
    __shared__ int32_t s_pos[8];      // tile positions of the new record's five knots
    __shared__ __attribute__((aligned(16))) int32_t s_rb[9][16];   // first 64 bytes of up to nine neighbour records (speculative four + late five)
    __shared__ double s_hX[8];      // the five knots around the tile (value) ...
    __shared__ int32_t s_hI[8];     // ... and their sample indices
    int32_t *s_gi = reinterpret_cast<int32_t *>(s_S);

    const int sig = blockIdx.y;
#if ITD_PROF
    unsigned long long pt_last = __builtin_readcyclecounter(), pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, pt4 = 0, pt5 = 0, pt6 = 0, pt7 = 0, pt8 = 0;
    const unsigned long long pt_begin = pt_last;
    const unsigned long long pt_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    SigState *st = state + sig;
    const Tin *x = xin + (int64_t)sig * x_stride;
    const size_t slot0 = (size_t)sig * n_tiles;
    const int32_t *cnts = counts_in + slot0;
    const TileRec *recs = recs_in + slot0;
    const int n_groups = groups_of(n_tiles);
    const int32_t *gs = gsum_in + (size_t)sig * n_groups * kGsumPitch;
#if ITD_EARLY_STOP
    // a signal that stopped at an earlier level streams nothing: the test comes before any load is issued (one scalar
    // load in front of the wavefront's requests; A/B in profiles/r02_ablation.txt).  Not in the fused level-0 launch: level 0
    // never stops and is never part of a device-side repair — the round trip in front of its loads bought nothing
    if constexpr (!FUSE0) {
        const int sl = st->stop_level;
        if ((sl >= 0 && sl < level) || st->skip > 0) return;
    }
#endif

    // ---- everything the wavefront's KT tiles need from HBM is requested up front (vector loads only: they retire in order,
    //      so the wait for one tile's data leaves the later tiles' requests — and the earlier tiles' stores — in flight):
    //      the second tile's load latency hides behind the first tile's work, the first tile's store acknowledgements behind
    //      the second tile's ---------------------------------------------------------------------------------------------
    struct TileLoads {
        Tin q[G2][2];               // the tile: two consecutive samples per lane and 128-sample group
        unsigned long long own;     // lane j < 2*G2: the tile's flag word j; the other lanes: its packed knot word
        int cb, cf;                 // knot counts of tiles t-1-lane / t+1+lane
        int specw;                  // lanes 16q..16q+15: first 64 bytes of the record of tile t-1, t+1, t-2, t+2 (speculative)
        Tin hq[2][2];               // FUSE0: the 128 samples in front of the tile and the 128 behind it
    };
    TileLoads pre[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int t = launch_item(blockIdx.x, gridDim.x, level) * KT + k;
        const int lane = lane_id();
        if constexpr (FUSE0) {
            if (t < n_tiles) {
                const int si = t * TW, rem = (int)n - si;   // 32-bit: see the note at `si` below
                // halo groups through descriptors of their own: an empty one (extent 0) reads zeros, so tile 0 / the last
                // tiles need no branch
                const __amdgpu_buffer_rsrc_t rl = tile_rsrc32(x + (si >= 128 ? si - 128 : 0), si >= 128 ? 128 : 0, sizeof(Tin));
                const __amdgpu_buffer_rsrc_t rh = tile_rsrc32(x + (rem > TW ? si + TW : 0), rem > TW ? rem - TW : 0, sizeof(Tin));
                const __amdgpu_buffer_rsrc_t rx = tile_rsrc32(x + si, rem, sizeof(Tin));
                tile_load2<false>(rl, lane * 2 * (int)sizeof(Tin), 0, pre[k].hq[0][0], pre[k].hq[0][1]);
#pragma unroll
                for (int g = 0; g < G2; ++g)
                    tile_load2<(ITD_NT & 4) != 0>(rx, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), pre[k].q[g][0], pre[k].q[g][1]);
                tile_load2<false>(rh, lane * 2 * (int)sizeof(Tin), 0, pre[k].hq[1][0], pre[k].hq[1][1]);
            }
        } else
        if (t < n_tiles) {
            // the small loads first: loads retire in order, and the halo logic that needs them should not wait for the tile
            pre[k].own = (ITD_ABL_R & 32768) ? 0ull : reinterpret_cast<const unsigned long long *>(recs + t)[lane < 2 * G2 ? 8 + lane : 0];
            const int tb = t - 1 - lane, tf = t + 1 + lane;
            pre[k].cb = (tb >= 0 && !(ITD_ABL_R & (1 | 32768))) ? cnts[tb] : 0;
            pre[k].cf = (tf < n_tiles && !(ITD_ABL_R & (1 | 32768))) ? cnts[tf] : 0;
            const int q4 = lane >> 4, w16 = lane & 15;
            const int uspec = q4 == 0 ? t - 1 : q4 == 1 ? t + 1 : q4 == 2 ? t - 2 : t + 2;
            pre[k].specw = 0;
            if (uspec >= 0 && uspec < n_tiles && !(ITD_ABL_R & (2 | 32768))) pre[k].specw = reinterpret_cast<const int32_t *>(recs + uspec)[w16];
            const int si = t * TW;
            const __amdgpu_buffer_rsrc_t rx = tile_rsrc32(x + si, (int)n - si, sizeof(Tin));
#pragma unroll
            for (int g = 0; g < G2; ++g)
                tile_load2<(ITD_NT & 4) != 0>(rx, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), pre[k].q[g][0], pre[k].q[g][1]);
        }
    }

    auto bare = [&]() {   // timing only (ITD_ABL_R 131072 / 262144): the three streams and nothing else, like tools/membench2
        const int lane = lane_id();
        const int t = launch_item(blockIdx.x, gridDim.x, level) * KT;
        const int64_t s = (int64_t)t * TW;
        double *rot_t = rot_out + (int64_t)sig * rot_stride + s;
        double *bas_t = FINAL ? rot_t : base_out + (int64_t)sig * base_stride + s;
        const __amdgpu_buffer_rsrc_t r_rot = tile_rsrc(rot_t, (n - s) * 8), r_bas = tile_rsrc(bas_t, FINAL ? 0 : (n - s) * 8);
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const double a = (double)pre[0].q[g][0] * 0.5 + (double)(pre[0].cb + pre[0].cf + pre[0].specw) + (double)(int)pre[0].own, b = (double)pre[0].q[g][1] * 0.5;
            tile_store2<(ITD_NT & 1) != 0>(r_rot, (128 * g + 2 * lane) * 8, a, b);
            if constexpr (!FINAL) tile_store2<(ITD_NT & 2) != 0>(r_bas, (128 * g + 2 * lane) * 8, b, a);
        }
    };
    if (ITD_ABL_R & 131072) { bare(); return; }
#if !ITD_EARLY_STOP
    const int sl_late = st->stop_level;
#endif
    // the four end samples of the level's input (ITD.py:101-102): published by the launch that wrote it; FUSE0: the signal's own
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
    if constexpr (FUSE0) {
        // only a tile within the halo search's reach of an end of the signal (128 + kReach x 510 samples: 9 tiles) can ever use them:
        // every other wavefront goes on without waiting for four scalar loads from the far ends of the signal
        const int t_first = launch_item(blockIdx.x, gridDim.x, level) * KT;
        if (t_first < 12 || t_first + KT + 12 > n_tiles) { e0 = (double)x[0]; e1 = (double)x[1]; e2 = (double)x[n - 2]; e3 = (double)x[n - 1]; }
    } else {
        e0 = st->ends[level & 1][0]; e1 = st->ends[level & 1][1]; e2 = st->ends[level & 1][2]; e3 = st->ends[level & 1][3];
    }
#if !ITD_EARLY_STOP
    if (sl_late >= 0 && sl_late < level) return;
#endif
    if (ITD_ABL_R & 262144) { if (e0 + e3 != 1.2345e300) bare(); return; }
    const double m0 = (e0 + e1) / 2.0;     // numpy.mean(x[:2]),  ITD.py:101
    const double mn = (e2 + e3) / 2.0;     // numpy.mean(x[-2:]), ITD.py:102
    double *ends_next = st->ends[(level + 1) & 1];
    const double inf = __builtin_huge_val();
    PROF_MARK(0)   // requests issued, the signal's end samples here

#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int t = launch_item(blockIdx.x, gridDim.x, level) * KT + kt;
    if (t >= n_tiles) break;
    __builtin_amdgcn_sched_barrier(0);   // no code motion between the tiles: the later tile only keeps its loaded registers alive
    int lane_v = lane_id();
    asm volatile("" : "+v"(lane_v));   // opaque per tile: nothing derived from the lane index is shared between the tiles' code
    const int lane = lane_v;
    const int q4 = lane >> 4, w16 = lane & 15;
    const int64_t s = (int64_t)t * TW;
    // Sample arithmetic in 32 bits (n < 2^31 - 1: itd_engine.hip check_args), relative to the tile where it could overflow:
    // gfx9 has no scalar 64-bit ordered compare, so an int64 `<` on these wave-uniform values becomes a vector compare, its
    // result a lane mask, and everything that depends on it exec-masked vector code
    const int ni = (int)n, si = t * TW;
    const int rem = ni - si;            // samples of the signal from the tile's first on (>= 1)
    const bool full = rem >= TW;
    int nb = 0, nf = 0, own_c = 0;   // real knots found in front (0..2) / behind (0..3); the tile's own knots
    // the tile's own knots, decoded once: kinfo[g] = ke | bitE << 16 | bitO << 17 with ke = knots of the tile at or before the
    // lane's even sample of group g; gcnt = the groups' knot counts, 8 bits each
    int kinfo[G2];
    unsigned gcnt = 0;
    if constexpr (!FUSE0) {
        const unsigned long long own_word = pre[kt].own;
        const unsigned long long own_packed = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(own_word >> 32), 63) << 32) |
                                              (unsigned)__builtin_amdgcn_readlane((int)(unsigned)own_word, 63);
        own_c = (ITD_ABL_R & (8 | 32768)) ? 0 : rec_count(own_packed);
        const int cb = pre[kt].cb, cf = pre[kt].cf, specw = pre[kt].specw;

        // ---- candidate tiles: nearest non-empty ones in the +-64-tile count windows ---------------------------------------------
        int ub0 = -1, ub1 = -1, uf0 = -1, uf1 = -1, uf2 = -1, cb0 = 0, cf0 = 0, cf1 = 0;
        if (ITD_ABL_R & 8192) {   // timing skeleton: keep the loads alive, skip the selection
            if (cb + cf + specw == 0x7fffffff) ub0 = 0;
        } else {
            unsigned long long mb = __ballot(cb != 0), mf = __ballot(cf != 0);
            if (mb) {
                const int l = __ffsll((long long)mb) - 1; mb &= mb - 1;
                ub0 = t - 1 - l; cb0 = __builtin_amdgcn_readlane(cb, l);
                if (cb0 < 2 && mb) ub1 = t - 1 - (__ffsll((long long)mb) - 1);
            }
            if (mf) {
                const int l = __ffsll((long long)mf) - 1; mf &= mf - 1;
                uf0 = t + 1 + l; cf0 = __builtin_amdgcn_readlane(cf, l);
                if (cf0 < 3 && mf) {
                    const int l1 = __ffsll((long long)mf) - 1; mf &= mf - 1;
                    uf1 = t + 1 + l1; cf1 = __builtin_amdgcn_readlane(cf, l1);
                    if (cf0 + cf1 < 3 && mf) uf2 = t + 1 + (__ffsll((long long)mf) - 1);
                }
            }
        }
        // staging slots: 0..3 = the speculative records, 4+k = candidate k fetched after all (rare)
        auto slot_of = [&](int u, int k) { return u == t - 1 ? 0 : u == t + 1 ? 1 : u == t - 2 ? 2 : u == t + 2 ? 3 : 4 + k; };
        const int sb0 = slot_of(ub0, 0), sb1 = slot_of(ub1, 1), sf0 = slot_of(uf0, 2), sf1 = slot_of(uf1, 3), sf2 = slot_of(uf2, 4);
        s_rb[q4][w16] = specw;
        if (!(ITD_ABL_R & 8192) && ((ub0 >= 0 && sb0 >= 4) || (ub1 >= 0 && sb1 >= 4) || (uf0 >= 0 && sf0 >= 4) || (uf1 >= 0 && sf1 >= 4))) {
            const int uk = q4 == 0 ? ub0 : q4 == 1 ? ub1 : q4 == 2 ? uf0 : uf1;
            const int sk = q4 == 0 ? sb0 : q4 == 1 ? sb1 : q4 == 2 ? sf0 : sf1;
            if (uk >= 0 && sk >= 4) s_rb[sk][w16] = reinterpret_cast<const int32_t *>(recs + uk)[w16];
        }
        if (uf2 >= 0 && sf2 >= 4 && lane < 16) s_rb[8][lane] = reinterpret_cast<const int32_t *>(recs + uf2)[lane];

        // staged record slot -> its tile (for the knots' absolute sample indices)
        auto tile_of = [&](int sl) { return sl == 0 ? t - 1 : sl == 1 ? t + 1 : sl == 2 ? t - 2 : sl == 3 ? t + 2 : sl == 4 ? ub0 : sl == 5 ? ub1 : sl == 6 ? uf0 : sl == 7 ? uf1 : uf2; };
        // ---- halo knots: two in front of the tile (slot 1 starts the segment that enters the tile), three behind; lane k < 5
        //      picks knot k out of the staged records.  Missing ones are the end knots e[0] = 0 / e[m+1] = n-1 (ITD.py:96,98).
        if (ub0 >= 0) nb = (cb0 >= 2 || ub1 >= 0) ? 2 : 1;
        if (uf0 >= 0) nf = min(3, cf0 + cf1 + (uf2 >= 0 ? 1 : 0));
        wave_sync();
        if (lane < 5 && !(ITD_ABL_R & 8192)) {
            int sl, sh, vw;   // staged slot, bit offset of the knot's position in the packed word, dword of its value
            bool real;
            if (lane < 2) {
                real = (lane == 1) ? (nb >= 1) : (nb >= 2);
                const bool from0 = (lane == 1) || (cb0 >= 2);          // slot 0: ub0's second last knot, or ub1's last
                const int which = (lane == 1 || !from0) ? 1 : 0;       // tpos(1) / tval[1] = the record's last knot
                sl = from0 ? sb0 : sb1;
                sh = 41 + 10 * which;
                vw = kRecDwTval + 2 * which;
            } else {
                int j = lane - 2;
                real = j < nf;
                const int c0 = min(cf0, 3), c1 = min(cf1, 3);
                if (j < c0) sl = sf0;
                else if (j - c0 < c1) { sl = sf1; j -= c0; }
                else { sl = sf2; j -= c0 + c1; }
                sh = 11 + 10 * j;
                vw = kRecDwHval + 2 * j;
            }
            int32_t idx = lane < 2 ? 0 : ni - 1;
            double val = lane < 2 ? e0 : e3;
            if (real) {
                const unsigned long long pk = ((unsigned long long)(unsigned)s_rb[sl][1] << 32) | (unsigned)s_rb[sl][0];
                idx = tile_of(sl) * TW + (int)((pk >> sh) & 0x3ffull);
                val = bits_d((unsigned)s_rb[sl][vw], (unsigned)s_rb[sl][vw + 1]);
            }
            s_hX[lane] = val;
            s_hI[lane] = idx;
        }
        if (lane < 2) s_bl[lane] = 0.0;
        wave_sync();
        // only when a 64-tile window ran dry does the search walk on through the group sums (dependent loads, rare)
        if (nb < 2 && t - 65 >= 0 && !(ITD_ABL_R & 1024)) {
            int far = t - 65;
            while (nb < 2) {
                int cu;
                const int u = far_nonempty<-1>(cnts, gs, n_tiles, far, &cu);
                if (u < 0) break;
                far = u - 1;
                const int uu = __builtin_amdgcn_readfirstlane(u);
                const TileRec *r = recs + uu;
                const unsigned long long pk = r->packed;
                const int32_t i1 = uu * TW + rec_tpos(pk, 1), i0 = uu * TW + rec_tpos(pk, 0);
                const double v1 = r->tval[1], v0 = r->tval[0];
                if (lane == 0) { s_hI[nb == 0 ? 1 : 0] = i1; s_hX[nb == 0 ? 1 : 0] = v1; }
                ++nb;
                if (nb < 2 && cu >= 2) {
                    if (lane == 0) { s_hI[0] = i0; s_hX[0] = v0; }
                    ++nb;
                }
            }
            wave_sync();
        }
        if (nf < 3 && t + 65 < n_tiles && !(ITD_ABL_R & 1024)) {
            int far = t + 65;
            while (nf < 3) {
                int cu;
                const int u = far_nonempty<1>(cnts, gs, n_tiles, far, &cu);
                if (u < 0) break;
                far = u + 1;
                const int uu = __builtin_amdgcn_readfirstlane(u);
                const TileRec *r = recs + uu;
                const int take = min(cu, 3 - nf);
                if (lane < take) { s_hI[2 + nf + lane] = uu * TW + rec_hpos(r->packed, lane); s_hX[2 + nf + lane] = r->hval[lane]; }
                nf += take;
            }
            wave_sync();
        }

        if (own_c > 0 && !(ITD_ABL_R & 2048)) {
            WaveMasks wm;   // lane j < 2*G2 holds flag word j
            if (lane < 2 * G2) { wm.lo = (unsigned)own_word; wm.hi = (unsigned)(own_word >> 32); }
            int gbase = 0;
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                const unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
                const int bE = lane_bit(E), bO = lane_bit(O);
                const int ke = mbcnt64(O, mbcnt64(E, gbase)) + bE;
                kinfo[g] = ke | (bE << 16) | (bO << 17);
                const int cg = __popcll(E) + __popcll(O);
                gcnt |= (unsigned)cg << (8 * g);
                gbase += cg;
            }
        } else {
#pragma unroll
            for (int g = 0; g < G2; ++g) kinfo[g] = 0;
        }
    } else {
        // ---- FUSE0 front end: knots of the caller's signal from the registers, halo knots from the two halo groups -----
        Tin v[G2 + 2][2];
        v[0][0] = pre[kt].hq[0][0]; v[0][1] = pre[kt].hq[0][1];
#pragma unroll
        for (int g = 0; g < G2; ++g) { v[g + 1][0] = pre[kt].q[g][0]; v[g + 1][1] = pre[kt].q[g][1]; }
        v[G2 + 1][0] = pre[kt].hq[1][0]; v[G2 + 1][1] = pre[kt].hq[1][1];
        {   // NaN input is rejected (ITD_ERR_NONFINITE): the reference would write +inf into the caller's array, ITD.py:46-51
            bool nan_in = false;   // samples beyond the row were read as zeros: one unordered compare per sample pair
#pragma unroll
            for (int g = 0; g < G2; ++g) nan_in = nan_in || __builtin_isunordered(pre[kt].q[g][0], pre[kt].q[g][1]);
            if (__any(nan_in) && lane == 0) st->in_nan = 1;
        }
        // window = [s-128, s+TW+128): positions allowed to flag are the signal's 1 .. n-2 (ITD.py:70-73)
        unsigned long long E[G2 + 2], O[G2 + 2];
        knot_masks<Tin, G2 + 2>(v, max(1, 129 - si), min(TW + 255, rem + 126), E, O);
        PROF_MARK(1)   // the tile has arrived; knot predicate on tile + halo groups
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const int bE = lane_bit(E[g + 1]), bO = lane_bit(O[g + 1]);
            const int ke = mbcnt64(O[g + 1], mbcnt64(E[g + 1], own_c)) + bE;
            kinfo[g] = ke | (bE << 16) | (bO << 17);
            const int cg = __popcll(E[g + 1]) + __popcll(O[g + 1]);
            gcnt |= (unsigned)cg << (8 * g);
            own_c += cg;
        }
        const int cL = __popcll(E[0]) + __popcll(O[0]), cR = __popcll(E[G2 + 1]) + __popcll(O[G2 + 1]);
        nb = min(cL, 2);
        nf = min(cR, 3);
        if (lane < 5) {   // defaults: the end knots e[0] = 0, e[m+1] = n-1 (ITD.py:96,98)
            s_hX[lane] = lane < 2 ? e0 : e3;
            s_hI[lane] = lane < 2 ? 0 : ni - 1;
        }
        if (lane < 2) s_bl[lane] = 0.0;
        // the samples next to the tile, where the record-driven launch keeps them (s_rb[0] / s_rb[1] edge fields)
        if (lane == 63) *reinterpret_cast<double *>(&s_rb[0][kRecDwEdge + 2]) = (double)v[0][1];
        if (lane == 0) *reinterpret_cast<double *>(&s_rb[1][kRecDwEdge]) = (double)v[G2 + 1][0];
        wave_sync();
        {   // owner lanes: the last two knots of the front halo group, the first three of the rear one
            const int bE = lane_bit(E[0]), bO = lane_bit(O[0]);
            const int re = mbcnt64(O[0], mbcnt64(E[0], 0));   // knots of the group before the even sample
            const int se = re - cL + 2, so = re + bE - cL + 2;
            const int32_t pe = si - 128 + 2 * lane;
            if (bE && se >= 0) { s_hX[se] = (double)v[0][0]; s_hI[se] = pe; }
            if (bO && so >= 0) { s_hX[so] = (double)v[0][1]; s_hI[so] = pe + 1; }
        }
        {
            const int bE = lane_bit(E[G2 + 1]), bO = lane_bit(O[G2 + 1]);
            const int re = mbcnt64(O[G2 + 1], mbcnt64(E[G2 + 1], 0));
            const int ro_ = re + bE;
            const int32_t pe = si + TW + 2 * lane;
            if (bE && re < 3) { s_hX[2 + re] = (double)v[G2 + 1][0]; s_hI[2 + re] = pe; }
            if (bO && ro_ < 3) { s_hX[2 + ro_] = (double)v[G2 + 1][1]; s_hI[2 + ro_] = pe + 1; }
        }
        wave_sync();
        // ---- a halo group with too few knots (smooth stretches): walk on through the signal.  Windows of 512 samples that
        //      overlap by two, so every position is an inner position of some window; wave-uniform, rare for noisy input ----
        int ql = si - 128;                     // front: first position not yet evaluated (everything above it is)
        unsigned qr = (unsigned)si + TW + 127; // rear: last position not yet evaluated (unsigned: may pass 2^31 at the far end)
        const unsigned n1 = (unsigned)ni - 1u;
        bool fail = false;
        if ((nb < 2 && ql > 0) || (nf < 3 && qr < n1)) {
            Tin w[4][2];
            unsigned long long WE[4], WO[4];
            for (int step = 0; nb < 2 && ql > 0 && step < kReach; ++step) {
                const int p0 = ql > 510 ? ql - 510 : 0;                     // window [p0, p0+512): inner positions p0+1 .. p0+510
                const __amdgpu_buffer_rsrc_t rw = tile_rsrc32(x + p0, ni - p0, sizeof(Tin));
#pragma unroll
                for (int g = 0; g < 4; ++g) tile_load2<false>(rw, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), w[g][0], w[g][1]);
                knot_masks<Tin, 4>(w, max(1, 1 - p0), ql - p0, WE, WO);
                for (int g = 3; g >= 0 && nb < 2; --g) {                     // nearest first: from the top
                    unsigned long long Eg = WE[g], Og = WO[g];
                    while ((Eg | Og) && nb < 2) {
                        const int be = Eg ? 63 - __clzll((long long)Eg) : -1, bo = Og ? 63 - __clzll((long long)Og) : -1;
                        const bool odd = bo >= be;
                        const int l = odd ? bo : be;
                        if (odd) Og &= ~(1ull << l); else Eg &= ~(1ull << l);
                        if (lane == l) {
                            const Tin val = odd ? (g == 0 ? w[0][1] : g == 1 ? w[1][1] : g == 2 ? w[2][1] : w[3][1]) : (g == 0 ? w[0][0] : g == 1 ? w[1][0] : g == 2 ? w[2][0] : w[3][0]);
                            s_hX[1 - nb] = (double)val;
                            s_hI[1 - nb] = (int32_t)(p0 + 128 * g + 2 * l + (odd ? 1 : 0));
                        }
                        ++nb;
                    }
                }
                ql = p0;
            }
            for (int step = 0; nf < 3 && qr < n1 && step < kReach; ++step) {
                const int p0 = (int)(qr - 1u);                               // inner positions qr .. qr+509 (p0 < n - 1: fits)
                const __amdgpu_buffer_rsrc_t rw = tile_rsrc32(x + p0, ni - p0, sizeof(Tin));
#pragma unroll
                for (int g = 0; g < 4; ++g) tile_load2<false>(rw, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), w[g][0], w[g][1]);
                knot_masks<Tin, 4>(w, 1, min(510, ni - 2 - p0), WE, WO);
                for (int g = 0; g < 4 && nf < 3; ++g) {
                    unsigned long long Eg = WE[g], Og = WO[g];
                    while ((Eg | Og) && nf < 3) {
                        const int be = Eg ? __ffsll((long long)Eg) - 1 : 64, bo = Og ? __ffsll((long long)Og) - 1 : 64;
                        const bool odd = bo < be;
                        const int l = odd ? bo : be;
                        if (odd) Og &= Og - 1; else Eg &= Eg - 1;
                        if (lane == l) {
                            const Tin val = odd ? (g == 0 ? w[0][1] : g == 1 ? w[1][1] : g == 2 ? w[2][1] : w[3][1]) : (g == 0 ? w[0][0] : g == 1 ? w[1][0] : g == 2 ? w[2][0] : w[3][0]);
                            s_hX[2 + nf] = (double)val;
                            s_hI[2 + nf] = (int32_t)(p0 + 128 * g + 2 * l + (odd ? 1 : 0));
                        }
                        ++nf;
                    }
                }
                qr += 510;
            }
            fail = (nb < 2 && ql > 0) || (nf < 3 && qr < n1);
            wave_sync();
        }
        if (fail && lane == 0) atomicOr(&st->l0_fail, 1);   // the engine repeats level 0 through k_scan0 + the record-driven launch
        PROF_MARK(2)   // ranks, halo knots
    }
    // ---- the tile itself, needed from here on ---------------------------------------------------------------------------
    double xr[G2][2];   // samples beyond the row were read as 0
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        xr[g][0] = (double)pre[kt].q[g][0];
        xr[g][1] = (double)pre[kt].q[g][1];
    }
    // ---- this level's knots inside the tile (the producer's flag words, or FUSE0's own scan); tile-relative ranks, once ----
    const int c = (ITD_ABL_R & 2048) ? 0 : own_c;
    double *rot_t = rot_out + (int64_t)sig * rot_stride + s;
    double *bas_t = FINAL ? nullptr : base_out + (int64_t)sig * base_stride + s;
    const __amdgpu_buffer_rsrc_t r_rot = tile_rsrc32(rot_t, rem, 8);
    const __amdgpu_buffer_rsrc_t r_bas = tile_rsrc32(FINAL ? rot_t : bas_t, FINAL ? 0 : rem, 8);
    const bool tail_tile = rem <= TW;       // holds sample n-1 (or runs past it)
    const bool near_end = rem <= TW + 2;    // holds sample n-2 or later
    const bool interior = full && !near_end && t != 0;   // no end-of-signal rule applies to any sample of the tile
    const int nrem = min(rem, TW + 2);      // samples of the signal from s on, clipped: p < nrem <=> s + p < n
    bool odd_vals = false, own_nan = false;   // wave-uniform: the new baseline tile holds a NaN or an infinity / a NaN
    // s_hX[5], s_hX[6] / s_hI[5], s_hI[6]: the previous run's last two knots; s_bl[0], s_bl[1]: baseline at samples s-1, s+TW

    // ---- passes --------------------------------------------------------------------------------------------------------
    int rb = 0, g0 = 0;
    while (g0 < G2) {
        int g1 = g0, m = 0;
        if (c - rb <= CAP) {   // everything that is left fits (one pass per tile unless the tile is dense)
            g1 = G2;
            m = c - rb;
        } else {
#pragma unroll 1
            for (int g = g0; g < G2; ++g) {
                const int cg = (int)((gcnt >> (8 * g)) & 0xffu);
                if (m + cg > CAP) break;
                m += cg;
                g1 = g + 1;
            }
        }
        const int nfp = min(3, (c - rb - m) + nf);   // real knots behind the run
        const bool has_ends = (rb == 0 && nb == 0) || nfp < 3;
        const bool one_pass = g0 == 0 && g1 == G2;   // the run is the whole tile: every own knot has a slot
      if (ITD_ABL_R & 2048) {   // timing skeleton: one constant segment instead of the knot phases
        if (lane == 0) { s_B[1] = 1.0; s_S[1] = 0.5; s_X[1] = 0.25; }
        wave_sync();
      } else {
        // own knots by rank (owner lanes), slots 0 .. m+4 of this pass
        if (one_pass) {
            // the common case: every own knot has a slot and every halo slot a halo knot.  No exec-masked regions: samples that
            // are not knots write to a spare slot behind the pass's slots
            constexpr int kSpare = RK - 1;
            if (c > 0) {
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    int p = 128 * g + 2 * lane, ki = kinfo[g];
                    asm volatile("" : "+v"(p), "+v"(ki));
                    const int Le = (ki & 0xffff) + 1;               // rb = 0: slot of the even sample's knot (rank ke-1)
                    const int se = (ki & 0x10000) ? Le : kSpare;
                    const int so = (ki & 0x20000) ? Le + 1 : kSpare;   // the odd sample's knot: rank ke
                    s_X[se] = xr[g][0];
                    s_gi[se] = si + p;
                    s_X[so] = xr[g][1];
                    s_gi[so] = si + p + 1;
                }
            }
            if (lane < 5) {   // slots 0, 1 and m+2 .. m+4: the five knots around the tile
                const int L = lane < 2 ? lane : m + lane;
                s_X[L] = s_hX[lane];
                s_gi[L] = s_hI[lane];
            }
        } else {
        if (c > 0) {
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                int p = 128 * g + 2 * lane, ki = kinfo[g];
                asm volatile("" : "+v"(p), "+v"(ki));   // opaque: decoded per use, not kept decoded across the pass loop
                const int ke = ki & 0xffff, bE = (ki >> 16) & 1, bO = (ki >> 17) & 1;
                const int Le = ke - rb + 1;          // slot of the even sample's knot (if it is one): rank ke-1
                const int Lo = Le + bO;              // slot of the odd sample's knot: rank ke+bO-1
                if (bE && (unsigned)(Le - 2) < (unsigned)(m + 3)) {
                    s_X[Le] = xr[g][0];
                    s_gi[Le] = si + p;
                }
                if (bO && (unsigned)(Lo - 2) < (unsigned)(m + 3)) {
                    s_X[Lo] = xr[g][1];
                    s_gi[Lo] = si + p + 1;
                }
            }
        }
        // slots whose rank lies outside the tile: the halo knots; slots 0,1 of a later pass: the previous run's last two
        // knots (their owners' registers hold the baseline by now)
        if (lane < 5) {
            const int r = lane < 2 ? rb - 2 + lane : rb + m + lane - 2;
            const int L = lane < 2 ? lane : m + lane;
            if (r < 0 || r >= c) {
                const int q = r < 0 ? 2 + r : 2 + (r - c);
                s_X[L] = s_hX[q];
                s_gi[L] = s_hI[q];
            } else if (lane < 2) {
                s_X[L] = s_hX[5 + lane];
                s_gi[L] = s_hI[5 + lane];
            }
        }
        }
        wave_sync();
        PROF_MARK(3)   // knots by rank into LDS
        if (g1 < G2 && lane < 2) { s_hX[5 + lane] = s_X[m + lane]; s_hI[5 + lane] = s_gi[m + lane]; }   // ranks rb+m-2, rb+m-1 (before the slopes reuse gi's bytes)
#if ITD_BS_SWEEP
        // ---- knot values, ITD.py:100-110, and per-segment slopes (B_{k+1}-B_k)/(x[e_{k+1}]-x[e_k]), ITD.py:115-116, in ONE sweep over
        //      the slots, highest first: a slot's slope needs the next slot's B — the next lane's (a DPP shift), for lane 63 the first
        //      lane's of the round before (carried on the scalar unit); positions, values and B stay in registers in between.  (The
        //      slopes overwrite the positions' bytes: a round's s_S[L] covers s_gi[2L], s_gi[2L+1], beyond what the lower rounds read.)
        {
            double b_carry = 0.0;
            for (int base = 1 + ((m + 2) & ~63); base >= 1; base -= kWave) {
                const int L = base + lane;
                const bool act = L <= m + 3;
                const int Lr = act ? L : 1;                                      // (lanes beyond the pass read slot 1 and write nothing)
                const int32_t k0 = s_gi[Lr - 1], k1 = s_gi[Lr], k2 = s_gi[Lr + 1];
                const double x0 = s_X[Lr - 1], x1 = s_X[Lr], x2 = s_X[Lr + 1];
                const double frac = int_ratio(k1 - k0, k2 - k0);
                const double tt = frac * (x2 - x0);
                const double u = x0 + tt;
                double Bv = 0.5 * u + 0.5 * x1;                                  // ITD.py:107-110
                bool endn = false;
                if (has_ends) {   // wave-uniform: only next to the signal's ends is a slot of the pass one of the two end knots
                    const bool end0 = (L == 1) && (rb == 0) && (nb == 0);         // e[0]   = sample 0
                    endn = (L >= m + 2) && (L - (m + 2) >= nfp);                  // e[m+1] = sample n-1 (starts no segment)
                    Bv = end0 ? m0 : (endn ? mn : Bv);
                }
                const double Bn = wave_dpp<0x130>(b_carry, Bv);                  // lane l <- lane l + 1; lane 63 keeps the carry
                const double sl = (Bn - Bv) / (x2 - x1);
                wave_sync();                                                     // (the round's position reads are done before its slopes land on them)
                if (act) s_B[L] = Bv;
                if (L <= m + 2 && !endn) s_S[L] = sl;
                b_carry = bits_d((unsigned)__builtin_amdgcn_readlane((int)(unsigned)dbits(Bv), 0), (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(dbits(Bv) >> 32), 0));
            }
        }
        wave_sync();
#else
        // (A/B builds, -DITD_BS_SWEEP=0: round 4's two loops — knot values, a round trip through LDS, slopes)
        for (int L = 1 + lane; L <= m + 3; L += kWave) {
            const int32_t k0 = s_gi[L - 1], k1 = s_gi[L], k2 = s_gi[L + 1];
            const double x0 = s_X[L - 1], x1 = s_X[L], x2 = s_X[L + 1];
            const double frac = (double)(k1 - k0) / (double)(k2 - k0);
            const double tt = frac * (x2 - x0);
            const double u = x0 + tt;
            double Bv = 0.5 * u + 0.5 * x1;
            if (has_ends) {
                const bool end0 = (L == 1) && (rb == 0) && (nb == 0);
                const bool endn = (L >= m + 2) && (L - (m + 2) >= nfp);
                Bv = end0 ? m0 : (endn ? mn : Bv);
            }
            s_B[L] = Bv;
        }
        wave_sync();
        for (int L = 1 + lane; L <= m + 2; L += kWave) {
            const double sl = (s_B[L + 1] - s_B[L]) / (s_X[L + 1] - s_X[L]);
            const bool endn = has_ends && (L >= m + 2) && (L - (m + 2) >= nfp);
            if (!endn) s_S[L] = sl;
        }
        wave_sync();
#endif
        PROF_MARK(4)   // knot values
        PROF_MARK(5)   // slopes
      }
        // ---- baseline at the two samples next to the tile (lane 0) -------------------------------------------------
        if (lane == 0) {
            if (g0 == 0 && t >= 1) {   // sample s-1 = the last sample of tile t-1
                const double xlo = bits_d((unsigned)s_rb[0][kRecDwEdge + 2], (unsigned)s_rb[0][kRecDwEdge + 3]);
                s_bl[0] = s_B[1] + s_S[1] * (xlo - s_X[1]);
            }
            if (g1 == G2 && rem > TW + 1) {   // sample s+TW exists and is not the signal's last
                const int L = (nf >= 1 && s_hI[2] == si + TW) ? m + 2 : 1 + m;   // sample s+TW is itself a knot
                const double xhi = bits_d((unsigned)s_rb[1][kRecDwEdge], (unsigned)s_rb[1][kRecDwEdge + 1]);   // first sample of tile t+1
                s_bl[1] = s_B[L] + s_S[L] * (xhi - s_X[L]);
            }
        }
#if ITD_PRIO
        __builtin_amdgcn_s_setprio(ITD_PRIO);   // a wavefront that has reached its store phase goes first: it retires sooner
#endif
        // ---- baseline map + rotation, ITD.py:114-119; the baseline replaces the input in the registers --------------
        if (interior && g0 == 0 && g1 == G2) {
            // the common case, straight-line: a full, aligned tile away from both ends of the signal, one pass
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                __builtin_amdgcn_sched_barrier(0);   // one group's by-rank reads in flight at a time: registers, not latency, are scarce
                int p = 128 * g + 2 * lane, ki = kinfo[g];
                asm volatile("" : "+v"(p), "+v"(ki));
                const int ke = ki & 0xffff, bO = (ki >> 17) & 1;
                const int Le = ke + 1, Lo = Le + bO;   // rb = 0
                const double xe = xr[g][0], xo = xr[g][1];
                const double be = s_B[Le] + s_S[Le] * (xe - s_X[Le]);
                const double bo = s_B[Lo] + s_S[Lo] * (xo - s_X[Lo]);
                const double re = FINAL ? (xe - be) + be : xe - be;   // FINAL: rotation + baseline, ITD.py:420
                const double rq = FINAL ? (xo - bo) + bo : xo - bo;
                xr[g][0] = be;
                xr[g][1] = bo;
                tile_store2<(ITD_NT & 1) != 0>(r_rot, p * 8, re, rq);
                double se = be, so = bo;
                if (__builtin_expect(__any(nonfinite(be) || nonfinite(bo)), 0)) {   // rare: the stored baseline is the mutated one, ITD.py:50
                    odd_vals = true;
                    own_nan = own_nan || __any(__builtin_isunordered(be, bo));
                    if (!keep_nan) {
                        se = be != be ? inf : be;
                        so = bo != bo ? inf : bo;
                    }
                }
                if constexpr (!FINAL) tile_store2<(ITD_NT & 2) != 0>(r_bas, p * 8, se, so);
            }
        } else
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            if (g >= g0 && g < g1) {
                int p = 128 * g + 2 * lane, ki = kinfo[g];   // tile position of the even sample; its rank word
                asm volatile("" : "+v"(p), "+v"(ki));        // opaque: keeps per-group masks, addresses and decoded ranks out of the pass loop's preheader
                const int ke = ki & 0xffff, bO = (ki >> 17) & 1;
                const int Le = ke - rb + 1, Lo = Le + bO;   // 1 + knots of the run at or before the sample
                const double xe = xr[g][0], xo = xr[g][1];
                double be = s_B[Le] + s_S[Le] * (xe - s_X[Le]);
                double bo = s_B[Lo] + s_S[Lo] * (xo - s_X[Lo]);
                if (tail_tile) {   // baseline[n-1] is never written (stays 0), ITD.py:112-117
                    if (p >= nrem - 1) be = 0.0;
                    if (p + 1 >= nrem - 1) bo = 0.0;
                }
                const double re = FINAL ? (xe - be) + be : xe - be;   // FINAL: rotation + baseline, ITD.py:420
                const double rq = FINAL ? (xo - bo) + bo : xo - bo;
                xr[g][0] = be;
                xr[g][1] = bo;
                tile_store2<(ITD_NT & 1) != 0>(r_rot, p * 8, re, rq);   // samples beyond the row are dropped by the bounds check
                double se = be, so = bo;
                if (__builtin_expect(__any(nonfinite(be) || nonfinite(bo)), 0)) {   // rare: the stored baseline is the mutated one, ITD.py:50
                    odd_vals = true;
                    own_nan = own_nan || __any(__builtin_isunordered(be, bo));
                    if (!keep_nan) {
                        se = be != be ? inf : be;
                        so = bo != bo ? inf : bo;
                    }
                }
                if constexpr (!FINAL) tile_store2<(ITD_NT & 2) != 0>(r_bas, p * 8, se, so);
                if (!FINAL && (near_end || t == 0) && !(ITD_ABL_R & 16384)) {   // the next level's end samples (of the mutated array), ITD.py:101-102
                    if (t == 0 && p == 0) { ends_next[0] = se; ends_next[1] = so; }
                    if (p == nrem - 2) { ends_next[2] = se; ends_next[3] = so; }
                    if (p + 1 == nrem - 2) ends_next[2] = so;
                    if (p == nrem - 1) ends_next[3] = se;
                }
            }
        }
        rb += m;
        g0 = g1;
        wave_sync();
        PROF_MARK(6)   // map + stores issued
    }
    // ---- knots of the baseline just produced = the next level's input, on registers.  A tile that holds (or borders on) a
    //      NaN or an infinity first goes through the reference's NaN rules: what they add to the plain count, and the
    //      NaN -> +inf mutation the next level sees (nan_rules) ------------------------------------------------------------
    if (!(ITD_ABL_R & 65536)) {
        double x_lo = s_bl[0], x_hi = s_bl[1];
        if (__builtin_expect(odd_vals || __any(nonfinite(x_lo) || nonfinite(x_hi)), 0)) {
            const int delta = nan_rules<TW>(xr, x_lo, x_hi, s, nrem);
            if (lane == 0) {
                if (delta) atomicAdd(&st->c_delta[level], delta);
                if (own_nan) atomicOr(&st->nan_mask, 1 << level);
            }
        }
        // TIES on a record-driven launch (the level in front of the fused sparse levels): near ties of the baseline it has just written
        scan_publish<TW, FINAL, TIES && !FUSE0>(xr, x_lo, x_hi, s, nrem, slot0 + t, ((size_t)sig * n_groups + t / kTilesPerGroup) * kGsumPitch,
                                                counts_out, recs_out, gsum_out, s_rec, s_pos, FUSE0 ? own_c : 0,
                                                (TIES && !FUSE0) ? near_out + (slot0 + t) * 8 : nullptr);
    }
    wave_sync();   // the next tile's staging must not overtake this tile's LDS reads
    PROF_MARK(7)   // next level's scan + record
  }
    if (blockIdx.x == 0 && !(ITD_ABL_R & 16384)) {
        // ---- tile 0's wavefront, after its own tile (kept off the path between the loads and their first use: a branch
        //      with memory operations there makes the compiler wait for ALL loads at the join): total knot count of this
        //      level and the stop rule (ITD.py:400-404); nothing in this launch reads what it writes -----------------
        const int lane = lane_id();
        int acc = 0;
        for (int q = lane; q < n_groups; q += kWave) acc += gs[(size_t)q * kGsumPitch];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        int32_t *gc = gsum_clear + (size_t)sig * n_groups * kGsumPitch;
        if (level == 1) {
            // the fused level-0 launch could only accumulate the signal's own knot total (its tiles ran concurrently), in the ints
            // behind the group sums this launch reads (scan_publish's count_hi): complete now.  A record-driven level 0 has set m[0].
            int a0 = 0;
            for (int q = lane; q < n_groups; q += kWave) a0 += gs[(size_t)q * kGsumPitch + 1];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a0 += __shfl_xor(a0, d);
            if (lane == 0 && st->m[0] < 0) st->m[0] = a0;
        }
        if (lane == 0 && !FUSE0) {
            // the baseline this level reads held a NaN: the reference counted its knots under the NaN rules (nan_rules)
            if (level >= 1 && ((st->nan_mask >> (level - 1)) & 1)) acc += st->c_delta[level - 1];
            st->m[level] = acc;
            // the pending baseline is not decomposable: launches from level + 1 on do nothing (this one completes)
            if (level >= 1 && acc < 2 && st->stop_level < 0) st->stop_level = level;
        }
        for (int q = lane; q < n_groups; q += kWave) gc[(size_t)q * kGsumPitch] = 0;
    }
#if ITD_PROF
    if constexpr (FUSE0) {
        PROF_MARK(8)
        if (lane_id() == 0 && g_prof_buf) {
            unsigned long long *o = g_prof_buf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16;
            o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = pt3; o[4] = pt4; o[5] = pt5; o[6] = pt6; o[7] = pt7; o[8] = pt8;
            o[9] = pt_last - pt_begin; o[10] = pt_begin; o[11] = pt_last;
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            o[12] = xcc; o[13] = hwid;
            o[14] = __builtin_amdgcn_s_memrealtime() - pt_rt0; o[15] = pt_rt0;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_finalize: the stop test of the LAST pending baseline (ITD.py:400-404, it takes priority over the "Out of time!"
// branch) and the stop rule's row fix-up (ITD.py:404-416), on the device.
// Natural stop detected at level j (input of extraction j+1 has < 2 knots) means the reference's counter is
// c = j-1: row c of the result is baselines[c-1] (= the input of extraction j, B_c), or the untouched all-zero last
// row when c = 0.  Every workgroup totals the last level's group sums itself (a few KB from L2) so that all of them
// reach the same verdict without reading what workgroup 0 publishes.  grid = (blocks, batch), 256 threads.
// ---------------------------------------------------------------------------------------------
// a signal's state as a decomposition expects to find it (k_init_state, and k_finalize for the call after the next)
__device__ __forceinline__ void sig_state_reset(SigState *st)
{
    for (int j = 0; j < kMaxLevels; ++j) { st->m[j] = -1; st->c_delta[j] = 0; }
    st->stop_level = -1;
    st->nan_mask = 0;
    st->in_nan = 0;
    st->l0_fail = 0;
    st->fin_stopped = 0;
    st->fin_stop_level = -1;
    st->kf_fail = 0;
    st->res_fail = 0;
    st->skip = 0;
}

constexpr int kFinalizeThreads = 256;
__global__ __launch_bounds__(kFinalizeThreads) void k_finalize(double *__restrict__ rows, int64_t rows_stride,
                                                               int64_t n, const double *__restrict__ bases,
                                                               int64_t bases_stride, int64_t bases_row_pitch,
                                                               int bases_rotate, const int32_t *__restrict__ gsum_last,
                                                               int n_tiles, int level_last,
                                                               SigState *__restrict__ state,
                                                               SigState *__restrict__ other_state = nullptr,
                                                               int32_t *__restrict__ other_gsum = nullptr, int64_t other_third = 0)
{
    __shared__ int s_red[kFinalizeThreads / 64];
    const int sig = blockIdx.y;
    // the other set of states / group sums (itd_engine.hip): left in its initial state for the decomposition after this one —
    // the signal's state, and elements 0 and 1 (all that is ever written) of its group-sum slots in the three rotating buffers
    if (other_state) {
        if (blockIdx.x == 0 && threadIdx.x == 0) sig_state_reset(other_state + sig);
        const int n_groups = groups_of(n_tiles);
        int32_t *og = other_gsum + (size_t)sig * n_groups * kGsumPitch;
        for (int k = blockIdx.x * kFinalizeThreads + threadIdx.x; k < 3 * n_groups; k += gridDim.x * kFinalizeThreads) {
            int32_t *slot = og + (size_t)(k / n_groups) * other_third + (size_t)(k % n_groups) * kGsumPitch;
            slot[0] = 0;
            slot[1] = 0;
        }
    }
    SigState *st = state + sig;
    if (st->skip > 0) return;       // (the device-side repair: this signal's first result, rows and state, stands)
    int stop_level = st->stop_level;
    int stopped = stop_level >= 0;
    if (!stopped) {
        const int n_groups = groups_of(n_tiles);
        const int32_t *gs = gsum_last + (size_t)sig * n_groups * kGsumPitch;
        int acc = 0;
        for (int k = threadIdx.x; k < n_groups; k += kFinalizeThreads) acc += gs[(size_t)k * kGsumPitch];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        int m_last = 0;
#pragma unroll
        for (int k = 0; k < kFinalizeThreads / 64; ++k) m_last += s_red[k];
        // the last pending baseline held a NaN: the reference counted its knots under the NaN rules (nan_rules)
        if (level_last >= 1 && ((st->nan_mask >> (level_last - 1)) & 1)) m_last += st->c_delta[level_last - 1];
        if (m_last < 2) { stopped = 1; stop_level = level_last; }
        if (blockIdx.x == 0 && threadIdx.x == 0) st->m[level_last] = m_last;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->fin_stopped = stopped;
        st->fin_stop_level = stop_level;
    }
    if (!stopped) return;
    const int c = stop_level - 1;
    double *dst = rows + (int64_t)sig * rows_stride + (int64_t)c * n;
    // B_c was written by extraction c (level c-1) into baselines row c-1 (rotating slots: (c-1) % 3)
    const double *src = nullptr;
    if (c >= 1) {
        const int row = bases_rotate ? ((c - 1) % bases_rotate) : (c - 1);
        src = bases + (int64_t)sig * bases_stride + (int64_t)row * bases_row_pitch;
    }
    // the copy: 16 bytes per access when both rows allow it (a thread moves several: the grid is sized for >= 4 per thread — with
    // one 8-byte element per thread a batch of short, naturally stopping signals spent most of this launch dispatching wavefronts)
    const int64_t i0 = (int64_t)blockIdx.x * kFinalizeThreads + threadIdx.x, stride = (int64_t)gridDim.x * kFinalizeThreads;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int64_t n2 = n >> 1;
        double2 *d2 = reinterpret_cast<double2 *>(dst);
        const double2 *s2 = reinterpret_cast<const double2 *>(src);
        for (int64_t i = i0; i < n2; i += stride) d2[i] = s2 ? s2[i] : make_double2(0.0, 0.0);
        if ((n & 1) && i0 == 0) dst[n - 1] = src ? src[n - 1] : 0.0;
    } else {
        for (int64_t i = i0; i < n; i += stride) dst[i] = src ? src[i] : 0.0;
    }
}

// knot values of an explicit knot list (host-facing single-level operator, ITD.py:100-110)
template <typename Tin>
__global__ void k_knot_values(const Tin *__restrict__ x, int64_t n, const int32_t *__restrict__ e, int m,
                              double *__restrict__ bk, int interior_only = 0 /* leave bk[0], bk[m+1] as the caller set them */)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > m + 1 || (interior_only && (k == 0 || k == m + 1))) return;
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
