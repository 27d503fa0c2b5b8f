// itd_chain.hpp — the whole level loop of a decomposition (ITD.itd, ITD.py:384-432) in ONE launch.  Opt-in
// (itd_set_chain_mode): bit-exact with the level-by-level engine on the whole test suite, measured 15 % slower than it on the
// benchmark signal because it is bound by instruction issue — kept as the working base of that direction.
//
// The level-by-level engine (itd_kernels.hpp) streams every level's input once: 8 B read + 16 B written per sample and
// level, and it sits at the floor of that traffic.  The only bytes left to remove are the baseline's round trip between two
// levels.  Here a wavefront keeps its tile IN REGISTERS through all levels: level 0 reads the caller's signal (the fused
// level-0 front end of k_extract), every level stores its rotation row (and the baseline row only if the caller asked for the
// reference's `baselines` buffer), and the next level starts from the baseline still held in the registers:
// 4 + 8 (L + 1) bytes per sample for L + 1 rows instead of 20 + 24 L.
//
// What a tile needs from its neighbours at level j — the two knots in front of it, the three behind it, the two samples
// next to it (the first 64 bytes of a TileRec) — exists only once those tiles have finished level j-1.  So the launch is a
// wave-granular pipeline:
//   * work is handed out by tickets (one returning atomic per tile), per XCD in chunks of 256 consecutive tiles, so a
//     wavefront only ever waits for tiles whose tickets were drawn before or shortly after its own: running wavefronts
//     hold a sliding window of the tile sequence, no co-residency of the whole grid is assumed;
//   * a tile publishes the record of its new baseline as sixteen 8-byte {tag, dword} granules, each ONE store, and its knot
//     count as one more; readers re-load granules with agent-scope (sc1) loads until every tag matches — the data is the
//     flag, no fence and no ordering between stores is needed (cdna_hip_programming.md, Guideline 16, form R2).  The tag is
//     the call's generation number, kept in device memory and advanced by the init kernel (replay-safe); every level has
//     granule arrays of its own, so nothing is overwritten while a slower neighbour may still need it; every array exists
//     in a same-XCD (L2) and a write-through copy (gran_store2);
//   * a wavefront holds two tiles: a level is attempted (one round of polls) and either runs to completion or leaves the tile
//     untouched, and the wavefront turns to its other tile instead of spinning;
//   * every wait is bounded (wall clock + a launch-wide give-up word): a tile whose halo knots lie beyond kChainWindows
//     x 64 tiles, or a wavefront that cannot move for ITD_CHAIN_SPIN_US, sets ChainCtl::give_up, every wavefront leaves, and
//     the engine repeats the call level by level (itd_get_summary), as it does when the stop rule `num_extrema < 2`
//     (ITD.py:404) fires inside the requested levels: the chain runs all levels optimistically and k_chain_finalize checks
//     the totals afterwards.
// Arithmetic, association order, NaN rules: the code of k_extract, unchanged.
#pragma once
#include "itd_kernels.hpp"

#pragma clang fp contract(off)

namespace itd {

#ifndef ITD_CHAIN_CAP
#define ITD_CHAIN_CAP 264   // by-rank slots per pass (level 0 is the densest level: see ITD_RANK_CAP0)
#endif
constexpr int kChainCap = ITD_CHAIN_CAP;
#ifndef ITD_CHAIN_TILE
#define ITD_CHAIN_TILE 1024  // samples per tile of the chain launch (its own choice: the level-by-level engine keeps ITD_TILE);
                             // 1024 measured 0.69 ms against 0.88 ms at 512 for the 2^24-sample benchmark signal: fewer, longer steps
#endif
constexpr int kChainTile = ITD_CHAIN_TILE;
constexpr int kChainChunk = 256;       // consecutive tiles per ticket chunk (one XCD's L2 sees a chunk's records)
constexpr int kChainWindows = 8;       // 64-tile count windows a tile walks per side before it gives up (well inside what an
                                       // XCD's wavefronts hold in flight: a tile only ever waits for tiles that are running)
#ifndef ITD_CHAIN_SPIN_US
#define ITD_CHAIN_SPIN_US 5000         // a wavefront that cannot move for this long gives up (s_memrealtime: 100 MHz)
#endif

struct ChainCtl {
    int32_t ticket[8][32];   // next ticket of XCD list x, one 128-byte line each
    int32_t give_up;         // != 0: some wavefront left early (bit 0 walk limit, bit 1 spin limit): repeat level by level
    uint32_t gen;            // tag of this call's granules (k_chain_init: ++gen, never 0)
    int32_t pad[30];
    unsigned long long prof[16];   // diagnostic builds (ITD_CHAIN_PROF): s_memrealtime ticks / event counts summed over all wavefronts
};
#ifndef ITD_CHAIN_SPARSE
#define ITD_CHAIN_SPARSE 6       // a tile with fewer knots than this polls the count windows together with the four records
#endif
#ifndef ITD_CHAIN_ABL
#define ITD_CHAIN_ABL 0          // diagnostics: 1 no rotation-row stores (timing only, results wrong), 2 granules are taken as
                                 // published whatever their tag (timing only), 4 every read from the write-through copy B
#endif
#ifndef ITD_CHAIN_CTX
#define ITD_CHAIN_CTX 2          // tile contexts per wavefront (1: a wavefront spins on its one tile)
#endif
// ISA markers for tools/isa_regions.py (comments in the assembly; no instructions)
#define CHAIN_MARK(n) asm volatile("; CHAIN_MARK " #n)
#ifndef ITD_CHAIN_PROF
#define ITD_CHAIN_PROF 0
#endif
#if ITD_CHAIN_PROF
#define CHAIN_T(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()
#define CHAIN_ACC(k, v) do { prof_acc[k] += (unsigned)(v); } while (0)   // per wavefront, flushed when it leaves
#else
#define CHAIN_T(var) do { } while (0)
#define CHAIN_ACC(k, v) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long gran_load(const unsigned long long *p, unsigned gen = 0)
{
    const unsigned long long g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ITD_CHAIN_ABL & 2) return ((unsigned long long)gen << 32) | (unsigned)g;
    return g;
}
__device__ __forceinline__ void gran_store(unsigned long long *p, unsigned tag, unsigned v)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// tile_rsrc with 32-bit arithmetic: `elems` elements of `elem_bytes` bytes from `base` on (0 = an empty descriptor: loads read 0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void *base, int elems, unsigned elem_bytes)
{
    const unsigned e = elems < 0 ? 0u : (unsigned)elems;
    const unsigned lim = e > 0x7fffffffu / elem_bytes ? 0x7fffffffu : e * elem_bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)lim, 0x00020000);
}

// Every granule array exists twice.  Copy A is written with PLAIN stores: the line stays (dirty) in the writer's XCD's L2,
// where an agent-scope (sc1: past L1, L2-served) load of another wavefront of the SAME XCD finds it at L2 latency.  Copy B is
// written with agent-scope (sc1, write-through) stores, which every XCD can read — from memory, several times slower, and the
// latency of a poll decides how long a tile waits for its neighbours.  A tile's ticket list belongs to one XCD (no stealing),
// so all tiles of a 256-tile chunk are produced on one XCD: a reader takes copy A for a tile of its own chunk, copy B
// otherwise.  A granule carries its tag, so a stale or never-written copy only ever reads as "not published yet".
__device__ __forceinline__ void gran_store2(unsigned long long *p, int64_t copy, unsigned tag, unsigned v)
{
    const unsigned long long g = ((unsigned long long)tag << 32) | v;
    __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);        // copy A: plain global_store
    __hip_atomic_store(p + copy, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // copy B: global_store sc1
}

// the launch-wide give-up word, wave-uniform for the compiler too (a vector load: the value is the same in every lane)
__device__ __forceinline__ bool chain_gave_up(ChainCtl *ctl)
{
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl->give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0;
}

// Setting it: a plain agent-scope store by every lane (same address, same value), not an atomic OR by lane 0 — a divergent
// branch in front of the wave-uniform `return` that follows would make that return, and with it every value that leaves the
// spin loops, divergent for the compiler (VGPRs and exec-mask control flow for what are scalars).  The last writer's code
// survives; the host only tests for non-zero.
__device__ __forceinline__ void chain_give_up(ChainCtl *ctl, int code)
{
    __hip_atomic_store(&ctl->give_up, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one step of a bounded spin: false = give up (the launch-wide word is set, or this spin has lasted too long)
struct ChainSpin {
    unsigned t0 = 0;    // low word of s_memrealtime (100 MHz) at the first check; 32-bit arithmetic throughout: gfx9 has no
    unsigned n = 0;     // scalar 64-bit ordered compare, and a vector compare would make the spin's exit divergent for the compiler
};
__device__ __forceinline__ bool chain_spin(ChainSpin &sp, ChainCtl *ctl)
{
    __builtin_amdgcn_s_sleep(2);
    if ((++sp.n & 31u) != 0) return true;
    if (chain_gave_up(ctl)) return false;
    const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime() | 1u;
    if (sp.t0 == 0) { sp.t0 = now; return true; }
    if (now - sp.t0 > (unsigned)ITD_CHAIN_SPIN_US * 100u) {
        chain_give_up(ctl, 2);
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------------------------
// One 64-tile count window (lane l holds the count granule of tile base + DIR * l, out-of-range tiles as published and empty):
// adds the window's non-empty tiles, nearest first, to the candidates until they hold `need` knots.  A window is usable up to
// its first tile that has not published yet.  Returns 1 = enough knots found, 2 = the whole window has published and holds
// too few (walk on), 0 = an unpublished tile is in the way (poll again; nothing is changed).
// Scalars only (no arrays: an indexed local array lives in scratch memory, and what comes back from there is no longer
// wave-uniform for the compiler).
template <int DIR>
__device__ __forceinline__ int chain_window(unsigned long long g, unsigned gen, int base, int need, int &acc, int &k,
                                            int &u0, int &u1, int &u2, int &c0, int &c1, int &c2)
{
    const unsigned long long rdy = __ballot((unsigned)(g >> 32) == gen);
    const int run = rdy == ~0ull ? 64 : __ffsll((long long)~rdy) - 1;      // tiles published, counted from the near end
    const int cnt = (int)(unsigned)g & 0x7ff;
    unsigned long long nz = __ballot(cnt != 0) & (run == 64 ? ~0ull : ((1ull << run) - 1ull));
    int a2 = acc, k2 = k, x0 = u0, x1 = u1, x2 = u2, y0 = c0, y1 = c1, y2 = c2;
    while (nz && a2 < need) {
        const int l = __ffsll((long long)nz) - 1;
        nz &= nz - 1;
        const int cl = __builtin_amdgcn_readlane(cnt, l);
        if (k2 == 0) { x0 = base + DIR * l; y0 = cl; }
        else if (k2 == 1) { x1 = base + DIR * l; y1 = cl; }
        else { x2 = base + DIR * l; y2 = cl; }
        a2 += cl;
        ++k2;
    }
    if (a2 < need && run < 64) return 0;
    u0 = x0; u1 = x1; u2 = x2;
    c0 = y0; c1 = y1; c2 = y2;
    acc = a2;
    k = k2;
    return a2 >= need ? 1 : 2;
}

// Nearest non-empty tiles of this level in direction DIR from tile t, enough of them to hold `need` knots (2 in front,
// 3 behind): u* / c* = tile and knot count, in walking order; polls the count granules window by window.  Returns the
// number of tiles found (the signal's end may come first), -2 if an unpublished tile is in the way (try again later), or -1
// to give up (walk limit).
template <int DIR>
__device__ __forceinline__ int chain_walk(const unsigned long long *__restrict__ cg, int64_t copy, int gu, int t, int n_tiles,
                                          unsigned gen, int need, int &u0, int &u1, int &u2, int &c0, int &c1, int &c2,
                                          ChainCtl *ctl, int lane)
{
    int acc = 0, k = 0;
    for (int w = 0;; ++w) {
        const int base = t + DIR * (1 + 64 * w);
        if (base < 0 || base >= n_tiles) return k;   // the signal's end: the missing knots are e[0] / e[m+1]
        if (w >= kChainWindows) {
            chain_give_up(ctl, 1);
            return -1;
        }
        const int tt = base + DIR * lane;
        const bool in = tt >= 0 && tt < n_tiles;
        const bool own_chunk = !(ITD_CHAIN_ABL & 4) && ((gu + (tt - t)) >> 8) == (gu >> 8);
        const unsigned long long g = in ? gran_load(cg + (own_chunk ? 0 : copy) + tt, gen) : ((unsigned long long)gen << 32);
        const int r = chain_window<DIR>(g, gen, base, need, acc, k, u0, u1, u2, c0, c1, c2);
        if (r == 1) return k;
        if (r == 0) return -2;
    }
}

// ---------------------------------------------------------------------------------------------
// chain_publish: scan_publish for the chain — knot scan of the baseline tile in registers, the count (group sum for the
// level's total; count granule for the neighbours' walks) and the neighbours' half of the record as sixteen granules.
// The flag words stay in registers (nm): the same wavefront consumes them at the next level.
// ---------------------------------------------------------------------------------------------
template <int TW>
__device__ __forceinline__ int chain_publish(int lane, const double (&xr)[TW / 128][2], double x_lo, double x_hi, bool first, int nrem,
                                             bool count_only, unsigned gen, unsigned long long *__restrict__ cnt_g,
                                             unsigned long long *__restrict__ rec_g, int64_t copy, int32_t *__restrict__ gsum_slot,
                                             int32_t *s_rec, int32_t *s_pos, WaveMasks &nm)
{
    constexpr int G2 = TW / 128;
    double d0[G2];
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        const double fill = (g == 0) ? x_lo : wave_dpp<0x13C>(0.0, xr[g > 0 ? g - 1 : 0][1]);
        const double left = wave_dpp<0x138>(fill, xr[g][1]);
        d0[g] = xr[g][0] - left;
    }
    const bool edge = first || (nrem <= TW + 1);   // the tile holds sample 0, or sample n-2 or later
    int total = 0;
    int gcount[G2];
#pragma unroll
    for (int g = 0; g < G2; ++g) {
        const double d1 = xr[g][1] - xr[g][0];
        const double fill = (g == G2 - 1) ? (x_hi - xr[g][1]) : wave_dpp<0x134>(0.0, d0[g < G2 - 1 ? g + 1 : g]);
        const double d2 = wave_dpp<0x130>(fill, d0[g]);
        bool fe = ((d1 > 0.0) && (d0[g] <= 0.0)) || ((d1 < 0.0) && (d0[g] >= 0.0));
        bool fo = ((d2 > 0.0) && (d1 <= 0.0)) || ((d2 < 0.0) && (d1 >= 0.0));
        if (edge) {   // first and last sample are never knots (ITD.py:70-73); nothing beyond sample n-2
            const int p = 128 * g + 2 * lane;
            fe = fe && (!first || p >= 1) && (p <= nrem - 2);
            fo = fo && (p + 1 <= nrem - 2);
        }
        const unsigned long long E = __ballot(fe), O = __ballot(fo);
        nm.set(2 * g, E);
        nm.set(2 * g + 1, O);
        gcount[g] = __popcll(E) + __popcll(O);
        total += gcount[g];
    }
    CHAIN_MARK(31);
    if (lane == 0) {
        if (total) atomicAdd(gsum_slot, total);
        if (!count_only) gran_store2(cnt_g, copy, gen, (unsigned)total);
    }
    if (count_only) return total;   // the "Out of time!" level: only the stop test reads this level's knots
    TileRec *lrec = reinterpret_cast<TileRec *>(s_rec);   // its first 64 bytes only
    if (lane < 5) s_pos[lane] = 0;
    if (lane == 0) lrec->edge[0] = xr[0][0];
    if (lane == 63) lrec->edge[1] = xr[G2 - 1][1];
    if (total > 0) {
        int gbase = 0;
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            if (gcount[g] > 0 && (gbase < 3 || gbase + gcount[g] > total - 2)) {
                const unsigned long long E = nm.get(2 * g), O = nm.get(2 * g + 1);
                const int bE = lane_bit(E), bO = lane_bit(O);
                const int re = mbcnt64(O, mbcnt64(E, gbase));
                const int ro_ = re + bE;
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
    }
    CHAIN_MARK(32);
    wave_sync();
    if (lane == 0) lrec->packed = rec_pack(total, s_pos[0], s_pos[1], s_pos[2], s_pos[3], s_pos[4]);
    wave_sync();
    if (lane < 16) gran_store2(rec_g + lane, 16 * copy, gen, (unsigned)s_rec[lane]);
    return total;
}

// ---------------------------------------------------------------------------------------------
// k_chain: grid = (workgroups, 1), 64 threads; any grid size is correct (tickets), ~the chip's residency is fastest.
//   cntg  [2][...][n_levels][batch * n_tiles]       count granules of level j's input (j >= 1); copy B starts gran_copy
//   recg  [2][...][n_levels][batch * n_tiles][16]   record granules of level j's input                  (16 gran_copy) elements on
//   gsum  [n_levels + 1][batch][groups * pitch]   knot totals of level j's input per 64-tile group (k_chain_finalize)
//   endg  [n_levels][batch][8]              the four end samples of level j's input (ITD.py:101-102), as granules
//   rows  rotation j+1 -> row j; the last level (n_levels - 1) writes rotation + baseline (ITD.py:420)
//   bases (WANT_BASES) baseline j+1 -> row j
// ---------------------------------------------------------------------------------------------
template <typename Tin, int TW, int CAP, bool WANT_BASES>
__global__ __launch_bounds__(kWave) void k_chain(const Tin *__restrict__ xin, int64_t x_stride, int64_t n, int n_tiles,
                                                   int batch, unsigned long long *__restrict__ cntg,
                                                   unsigned long long *__restrict__ recg, int64_t gran_copy,
                                                   int32_t *__restrict__ gsum,
                                                   int64_t gsum_level_pitch, unsigned long long *__restrict__ endg,
                                                   double *__restrict__ rows, int64_t rows_stride,
                                                   double *__restrict__ bases, int64_t bases_stride,
                                                   SigState *__restrict__ state, ChainCtl *__restrict__ ctl, int n_levels)
{
    constexpr int G2 = TW / 128;
    static_assert(TW % 128 == 0 && TW <= 1024 && CAP >= 128, "tile geometry: whole groups, 10-bit knot positions");
    constexpr int RK = CAP + 8;
    __shared__ double s_X[RK];
    __shared__ double s_B[RK];
    __shared__ double s_S[RK];
    __shared__ __attribute__((aligned(16))) int32_t s_rec[16];
    __shared__ double s_bl[2];
    __shared__ int32_t s_pos[8];
    __shared__ __attribute__((aligned(16))) int32_t s_rb[9][16];
    __shared__ double s_hX[8];
    __shared__ int32_t s_hI[8];
    int32_t *s_gi = reinterpret_cast<int32_t *>(s_S);

    const unsigned gen = ctl->gen;
    const int n_groups = groups_of(n_tiles);
    // 32-bit sample and tile indices throughout (the host sends only n <= 2^31 - 2^16 and batch * tiles < 2^31 here): gfx9 has
    // no scalar 64-bit ordered compare, so an int64 `<` on wave-uniform values becomes a vector compare and its users divergent
    const int ni = (int)n;
    const int lv_tiles = batch * n_tiles;
    const int n_chunks = (lv_tiles + kChainChunk - 1) / kChainChunk;
    const double inf = __builtin_huge_val();
#if ITD_CHAIN_PROF
    unsigned prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;

    // ---- two tile contexts per wavefront.  A tile waits at every level for its neighbours' records (a few microseconds at
    //      best, longer where knots are sparse and the tiles it needs drew their tickets later).  Instead of spinning, the
    //      wavefront parks the tile — 8 samples per lane, two flag-word registers, a few scalars — and works on its other tile:
    //      a level is attempted (one round of polls), and either runs to completion or leaves the tile untouched.
    double xr[G2][2];     // current tile: the level's input; after the map: the baseline = the next level's input
    WaveMasks own;        // lane j < 2*G2: flag word j of the knots of xr (levels >= 1: left by chain_publish)
    int own_total = 0, level = 0, gu = -1;        // gu: tile index over the whole batch, < 0 = no tile
    double pxr[G2][2];    // the parked tile
    WaveMasks pown;
    int pown_total = 0, plevel = 0, pgu = -1;
#pragma unroll
    for (int g = 0; g < G2; ++g) { xr[g][0] = xr[g][1] = pxr[g][0] = pxr[g][1] = 0.0; }
    bool exhausted = false;
    int idle = 0;
    ChainSpin sp;
    const int xl = xcc;   // this XCD's ticket list only: what a chunk's tiles publish stays readable in this XCD's L2 (gran_store2)

  for (;;) {
    if (gu < 0 && !exhausted) {
        if (chain_gave_up(ctl)) return;
        int tk = 0;
        if (lane_id() == 0) tk = atomicAdd(&ctl->ticket[xl][0], 1);   // (the only lane_id() outside a level's code)
        tk = __builtin_amdgcn_readfirstlane(tk);
        const int chunk = (int)((unsigned)tk >> 8) * 8 + xl;
        static_assert(kChainChunk == 256, "ticket -> chunk arithmetic");
        if (chunk >= n_chunks) exhausted = true;
        else {
            const int g_new = chunk * kChainChunk + (tk & (kChainChunk - 1));
            if (g_new >= lv_tiles) continue;   // the last chunk's unused tickets
            gu = g_new;
            level = 0;
        }
    }
    // one level of the current tile: 1 = done, 0 = its neighbours have not published yet (nothing was changed), -1 = give up
    auto step = [&]() __attribute__((always_inline)) -> int {
    const int sig = (int)((unsigned)gu / (unsigned)n_tiles);
    const int t = gu - sig * n_tiles;
    SigState *st = state + sig;
    const Tin *x = xin + (int64_t)sig * x_stride;
    const int si = t * TW;              // the tile's first sample
    const int rem = ni - si;            // samples of the signal from there on (>= 1)
    const bool tail_tile = rem <= TW;       // holds sample n-1 (or runs past it)
    const bool near_end = rem <= TW + 2;    // holds sample n-2 or later
    const int nrem = min(rem, TW + 2);      // clipped: p < nrem <=> s + p < n
    const size_t grp = ((size_t)sig * n_groups + t / kTilesPerGroup) * kGsumPitch;
    {

        const bool fin = (level == n_levels - 1);
        int lane_v = lane_id();
        asm volatile("" : "+v"(lane_v));   // opaque per level: masks and offsets derived from the lane index are not kept across levels
        const int lane = lane_v;
        const int q4 = lane >> 4, w16 = lane & 15;
        CHAIN_T(pt_lv0);
        int nb = 0, nf = 0, own_c = 0;
        int kinfo[G2];
        unsigned long long gcnt = 0;   // the groups' knot counts, 8 bits each
        double e0, e1, e2, e3;
        if (level == 0) {
            // ---- the caller's signal: tile + one 128-sample group either side, knots from the registers (k_extract, FUSE0) ----
            Tin q[G2][2], hq[2][2];
            {
                const __amdgpu_buffer_rsrc_t rl = chain_rsrc(x + (si >= 128 ? si - 128 : 0), si >= 128 ? 128 : 0, sizeof(Tin));
                const __amdgpu_buffer_rsrc_t rh = chain_rsrc(x + (rem > TW ? si + TW : 0), rem > TW ? rem - TW : 0, sizeof(Tin));
                const __amdgpu_buffer_rsrc_t rx = chain_rsrc(x + si, rem, sizeof(Tin));
                tile_load2<false>(rl, lane * 2 * (int)sizeof(Tin), 0, hq[0][0], hq[0][1]);
#pragma unroll
                for (int g = 0; g < G2; ++g)
                    tile_load2<true>(rx, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), q[g][0], q[g][1]);
                tile_load2<false>(rh, lane * 2 * (int)sizeof(Tin), 0, hq[1][0], hq[1][1]);
            }
            e0 = (double)x[0]; e1 = (double)x[1]; e2 = (double)x[ni - 2]; e3 = (double)x[ni - 1];
            Tin v[G2 + 2][2];
            v[0][0] = hq[0][0]; v[0][1] = hq[0][1];
#pragma unroll
            for (int g = 0; g < G2; ++g) { v[g + 1][0] = q[g][0]; v[g + 1][1] = q[g][1]; }
            v[G2 + 1][0] = hq[1][0]; v[G2 + 1][1] = hq[1][1];
            {   // NaN input is rejected (ITD_ERR_NONFINITE): the reference would write +inf into the caller's array, ITD.py:46-51
                bool nan_in = false;
#pragma unroll
                for (int g = 0; g < G2; ++g) nan_in = nan_in || __builtin_isunordered(q[g][0], q[g][1]);
                if (__any(nan_in) && lane == 0) st->in_nan = 1;
            }
            unsigned long long E[G2 + 2], O[G2 + 2];
            knot_masks<Tin, G2 + 2>(v, max(1, 129 - si), min(TW + 255, rem + 126), E, O);
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                const int bE = lane_bit(E[g + 1]), bO = lane_bit(O[g + 1]);
                const int ke = mbcnt64(O[g + 1], mbcnt64(E[g + 1], own_c)) + bE;
                kinfo[g] = ke | (bE << 16) | (bO << 17);
                const int cg = __popcll(E[g + 1]) + __popcll(O[g + 1]);
                gcnt |= (unsigned long long)cg << (8 * g);
                own_c += cg;
            }
            if (lane == 0 && own_c) atomicAdd(&gsum[grp], own_c);   // level 0's total (m[0])
            const int cL = __popcll(E[0]) + __popcll(O[0]), cR = __popcll(E[G2 + 1]) + __popcll(O[G2 + 1]);
            nb = min(cL, 2);
            nf = min(cR, 3);
            if (lane < 5) {   // defaults: the end knots e[0] = 0, e[m+1] = n-1 (ITD.py:96,98)
                s_hX[lane] = lane < 2 ? e0 : e3;
                s_hI[lane] = lane < 2 ? 0 : ni - 1;
            }
            if (lane < 2) s_bl[lane] = 0.0;
            if (lane == 63) *reinterpret_cast<double *>(&s_rb[0][kRecDwEdge + 2]) = (double)v[0][1];
            if (lane == 0) *reinterpret_cast<double *>(&s_rb[1][kRecDwEdge]) = (double)v[G2 + 1][0];
            wave_sync();
            {
                const int bE = lane_bit(E[0]), bO = lane_bit(O[0]);
                const int re = mbcnt64(O[0], mbcnt64(E[0], 0));
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
            int ql = si - 128, qr = si + TW + 127;
            bool fail = false;
            if ((nb < 2 && ql > 0) || (nf < 3 && qr < ni - 1)) {
                Tin w[4][2];
                unsigned long long WE[4], WO[4];
                for (int step = 0; nb < 2 && ql > 0 && step < kReach; ++step) {
                    const int p0 = ql > 510 ? ql - 510 : 0;
                    const __amdgpu_buffer_rsrc_t rw = chain_rsrc(x + p0, ni - p0, sizeof(Tin));
#pragma unroll
                    for (int g = 0; g < 4; ++g) tile_load2<false>(rw, lane * 2 * (int)sizeof(Tin), g * 128 * (int)sizeof(Tin), w[g][0], w[g][1]);
                    knot_masks<Tin, 4>(w, max(1, 1 - p0), ql - p0, WE, WO);
                    for (int g = 3; g >= 0 && nb < 2; --g) {
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
                for (int step = 0; nf < 3 && qr < ni - 1 && step < kReach; ++step) {
                    const int p0 = qr - 1;
                    const __amdgpu_buffer_rsrc_t rw = chain_rsrc(x + p0, ni - p0, sizeof(Tin));
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
                fail = (nb < 2 && ql > 0) || (nf < 3 && qr < ni - 1);
                wave_sync();
            }
            if (fail && lane == 0) atomicOr(&st->l0_fail, 1);
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                xr[g][0] = (double)q[g][0];
                xr[g][1] = (double)q[g][1];
            }
        } else {
            // ---- level j >= 1: the input is the baseline in the registers, its knots the flag words chain_publish left; the
            //      neighbours' records of this level arrive as granules ------------------------------------------------------
            CHAIN_MARK(10);
            own_c = own_total;
            const int64_t lv0 = (int64_t)level * lv_tiles + sig * n_tiles;   // this level's granules of the signal's tile 0
            const unsigned long long *cg = cntg + lv0;          // copy A; copy B = + gran_copy (records: 16 gran_copy)
            const unsigned long long *rg = recg + lv0 * 16;
            const int ch0 = (ITD_CHAIN_ABL & 4) ? -1 : (gu >> 8);   // ablation 4: every read from copy B
            auto cg_of = [&](int tt) { return cg + ((((gu + (tt - t)) >> 8) == ch0) ? 0 : gran_copy) + tt; };
            auto rg_of = [&](int tt) { return rg + ((((gu + (tt - t)) >> 8) == ch0) ? 0 : 16 * gran_copy) + tt * 16; };
            const int uspec = q4 == 0 ? t - 1 : q4 == 1 ? t + 1 : q4 == 2 ? t - 2 : t + 2;
            const bool inr = uspec >= 0 && uspec < n_tiles;
            int specw = 0;
            bool v2 = false, v3 = false;
            // candidate tiles: the nearest non-empty ones, two knots' worth in front of the tile, three behind it
            int ub0 = -1, ub1 = -1, ubx = -1, cb0 = 0, cb1 = 0, cbx = 0;
            int uf0 = -1, uf1 = -1, uf2 = -1, cf0 = 0, cf1 = 0, cf2 = 0;
            bool walk_b = false, walk_f = false;
            // Where knots are sparse (guess: this tile holds few) the nearest non-empty tiles are rarely t-1, t+1, t-2, t+2: the
            // two 64-tile count windows are polled together with those four records, one round trip instead of two.
            const bool sparse = own_total < ITD_CHAIN_SPARSE;
            {
                    const unsigned long long g = inr ? gran_load(rg_of(uspec) + w16, gen) : ((unsigned long long)gen << 32);
                    unsigned long long gb = (unsigned long long)gen << 32, gf = gb;
                    if (sparse) {
                        if (t - 1 - lane >= 0) gb = gran_load(cg_of(t - 1 - lane), gen);
                        if (t + 1 + lane < n_tiles) gf = gran_load(cg_of(t + 1 + lane), gen);
                    }
                    specw = (int)(unsigned)g;
                    const unsigned long long okm = __ballot((unsigned)(g >> 32) == gen);
                    const unsigned ok_lo = (unsigned)okm, ok_hi = (unsigned)(okm >> 32);   // 32-bit tests: scalar compares
                    const bool v0 = (ok_lo & 0xffffu) == 0xffffu, v1 = (ok_lo >> 16) == 0xffffu;
                    v2 = (ok_hi & 0xffffu) == 0xffffu;
                    v3 = (ok_hi >> 16) == 0xffffu;
                    bool ready = false;
                    if (v0 && v1) {   // tiles t-1 and t+1 have published (out-of-range neighbours count as published, empty)
                        const int c_m1 = __builtin_amdgcn_readlane(specw, 0) & 0x7ff;
                        const int c_p1 = __builtin_amdgcn_readlane(specw, 16) & 0x7ff;
                        const int c_m2 = v2 ? (__builtin_amdgcn_readlane(specw, 32) & 0x7ff) : 0;
                        const int c_p2 = v3 ? (__builtin_amdgcn_readlane(specw, 48) & 0x7ff) : 0;
                        const bool know_b = c_m1 >= 2 || v2, know_f = c_p1 >= 3 || v3;
                        bool ok_b = false, ok_f = false;
                        if (know_b && ((c_m1 + c_m2 >= 2) || (t - 3 < 0))) {
                            ok_b = true;
                            ub0 = ub1 = -1;
                            if (c_m1 > 0) {
                                ub0 = t - 1; cb0 = c_m1;
                                if (cb0 < 2 && c_m2 > 0) { ub1 = t - 2; cb1 = c_m2; }
                            } else if (c_m2 > 0) { ub0 = t - 2; cb0 = c_m2; }
                        } else if (sparse) {
                            int acc = 0, k = 0;
                            ub0 = ub1 = ubx = -1;
                            const int r = chain_window<-1>(gb, gen, t - 1, 2, acc, k, ub0, ub1, ubx, cb0, cb1, cbx);
                            ok_b = r != 0;
                            walk_b = (r == 2) && (t - 65 >= 0);   // (rare) more tiles beyond the window: the walk starts over
                        } else if (know_b) {
                            ok_b = walk_b = true;
                        }
                        if (know_f && ((c_p1 + c_p2 >= 3) || (t + 3 >= n_tiles))) {
                            ok_f = true;
                            uf0 = uf1 = uf2 = -1;
                            if (c_p1 > 0) {
                                uf0 = t + 1; cf0 = c_p1;
                                if (cf0 < 3 && c_p2 > 0) { uf1 = t + 2; cf1 = c_p2; }
                            } else if (c_p2 > 0) { uf0 = t + 2; cf0 = c_p2; }
                        } else if (sparse) {
                            int acc = 0, k = 0;
                            uf0 = uf1 = uf2 = -1;
                            const int r = chain_window<1>(gf, gen, t + 1, 3, acc, k, uf0, uf1, uf2, cf0, cf1, cf2);
                            ok_f = r != 0;
                            walk_f = (r == 2) && (t + 65 < n_tiles);
                        } else if (know_f) {
                            ok_f = walk_f = true;
                        }
                        ready = ok_b && ok_f;
                    }
                    if (!ready) return 0;
            }
            CHAIN_MARK(11);
            s_rb[q4][w16] = specw;
            if (walk_b) {
                ub0 = ub1 = ubx = -1; cb0 = cb1 = cbx = 0;
                const int r = chain_walk<-1>(cg, gran_copy, gu, t, n_tiles, gen, 2, ub0, ub1, ubx, cb0, cb1, cbx, ctl, lane);
                if (r < 0) return r == -2 ? 0 : -1;
            }
            if (walk_f) {
                uf0 = uf1 = uf2 = -1; cf0 = cf1 = cf2 = 0;
                const int r = chain_walk<1>(cg, gran_copy, gu, t, n_tiles, gen, 3, uf0, uf1, uf2, cf0, cf1, cf2, ctl, lane);
                if (r < 0) return r == -2 ? 0 : -1;
            }
            (void)cb1; (void)cbx; (void)ubx; (void)cf2;
            // staging slots: 0, 1 = tiles t-1, t+1 (always staged: their edge samples); 2, 3 = t-2, t+2 if their records came
            // with the first sweep; 4+k = candidate k fetched now
            auto slot_of = [&](int u, int k) { return u == t - 1 ? 0 : u == t + 1 ? 1 : (u == t - 2 && v2) ? 2 : (u == t + 2 && v3) ? 3 : 4 + k; };
            const int sb0 = slot_of(ub0, 0), sb1 = slot_of(ub1, 1), sf0 = slot_of(uf0, 2), sf1 = slot_of(uf1, 3), sf2 = slot_of(uf2, 4);
            if ((ub0 >= 0 && sb0 >= 4) || (ub1 >= 0 && sb1 >= 4) || (uf0 >= 0 && sf0 >= 4) || (uf1 >= 0 && sf1 >= 4)) {
                const int uk = q4 == 0 ? ub0 : q4 == 1 ? ub1 : q4 == 2 ? uf0 : uf1;
                const int sk = q4 == 0 ? sb0 : q4 == 1 ? sb1 : q4 == 2 ? sf0 : sf1;
                const bool want = uk >= 0 && sk >= 4;
                const unsigned long long g = want ? gran_load(rg_of(uk) + w16, gen) : ((unsigned long long)gen << 32);
                if (!__all((unsigned)(g >> 32) == gen)) return 0;
                if (want) s_rb[sk][w16] = (int)(unsigned)g;
            }
            if (uf2 >= 0 && sf2 >= 4) {
                const unsigned long long g = lane < 16 ? gran_load(rg_of(uf2) + lane, gen) : ((unsigned long long)gen << 32);
                if (!__all((unsigned)(g >> 32) == gen)) return 0;
                if (lane < 16) s_rb[8][lane] = (int)(unsigned)g;
            }
            auto tile_of = [&](int sl) { return sl == 0 ? t - 1 : sl == 1 ? t + 1 : sl == 2 ? t - 2 : sl == 3 ? t + 2 : sl == 4 ? ub0 : sl == 5 ? ub1 : sl == 6 ? uf0 : sl == 7 ? uf1 : uf2; };
            if (ub0 >= 0) nb = (cb0 >= 2 || ub1 >= 0) ? 2 : 1;
            if (uf0 >= 0) nf = min(3, cf0 + cf1 + (uf2 >= 0 ? 1 : 0));
            // the four end samples of this level's input (ITD.py:101-102): only tiles whose halo reaches an end of the signal
            // use them; they were published by the tiles that hold them, as granules
            e0 = e1 = e2 = e3 = 0.0;
            if (nb < 2 || nf < 3) {
                const unsigned long long *eg = endg + ((int64_t)level * batch + sig) * 8;
                const unsigned long long g = lane < 8 ? gran_load(eg + lane, gen) : ((unsigned long long)gen << 32);
                if (!__all((unsigned)(g >> 32) == gen)) return 0;
                const int lo = (int)(unsigned)g;
                e0 = bits_d((unsigned)__builtin_amdgcn_readlane(lo, 0), (unsigned)__builtin_amdgcn_readlane(lo, 1));
                e1 = bits_d((unsigned)__builtin_amdgcn_readlane(lo, 2), (unsigned)__builtin_amdgcn_readlane(lo, 3));
                e2 = bits_d((unsigned)__builtin_amdgcn_readlane(lo, 4), (unsigned)__builtin_amdgcn_readlane(lo, 5));
                e3 = bits_d((unsigned)__builtin_amdgcn_readlane(lo, 6), (unsigned)__builtin_amdgcn_readlane(lo, 7));
            }
            CHAIN_MARK(12);
            wave_sync();
            if (lane < 5) {
                int sl, sh, vw;
                bool real;
                if (lane < 2) {
                    real = (lane == 1) ? (nb >= 1) : (nb >= 2);
                    const bool from0 = (lane == 1) || (cb0 >= 2);
                    const int which = (lane == 1 || !from0) ? 1 : 0;
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
            CHAIN_MARK(13);
            if (own_c > 0) {
                int gbase = 0;
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    const unsigned long long E = own.get(2 * g), O = own.get(2 * g + 1);
                    const int bE = lane_bit(E), bO = lane_bit(O);
                    const int ke = mbcnt64(O, mbcnt64(E, gbase)) + bE;
                    kinfo[g] = ke | (bE << 16) | (bO << 17);
                    const int cgk = __popcll(E) + __popcll(O);
                    gcnt |= (unsigned long long)cgk << (8 * g);
                    gbase += cgk;
                }
            } else {
#pragma unroll
                for (int g = 0; g < G2; ++g) kinfo[g] = 0;
            }
        }
        CHAIN_MARK(14);
        CHAIN_T(pt_front);
        CHAIN_ACC(level == 0 ? 2 : 3, pt_front - pt_lv0);   // whole front end of level 0 / of a later level
#if ITD_CHAIN_PROF
        {   // front end by level (levels 1..8 -> slots 8..15)
            const unsigned d = (unsigned)(pt_front - pt_lv0);
            switch (level) { case 1: prof_acc[8] += d; break; case 2: prof_acc[9] += d; break; case 3: prof_acc[10] += d; break; case 4: prof_acc[11] += d; break;
                             case 5: prof_acc[12] += d; break; case 6: prof_acc[13] += d; break; case 7: prof_acc[14] += d; break; case 8: prof_acc[15] += d; break; default: break; }
        }
#endif
        const double m0 = (e0 + e1) / 2.0;     // numpy.mean(x[:2]),  ITD.py:101
        const double mn = (e2 + e3) / 2.0;     // numpy.mean(x[-2:]), ITD.py:102

        // ---- the extraction on the tile: k_extract's passes ---------------------------------------------------------------
        const int c = own_c;
        double *rot_t = rows + (int64_t)sig * rows_stride + (int64_t)level * n + si;
        const __amdgpu_buffer_rsrc_t r_rot = chain_rsrc(rot_t, rem, 8);
        const __amdgpu_buffer_rsrc_t r_bas = chain_rsrc(WANT_BASES ? bases + (int64_t)sig * bases_stride + (int64_t)level * n + si : rot_t,
                                                        (WANT_BASES && !fin) ? rem : 0, 8);
        unsigned long long *eg_next = endg + ((int64_t)(level + 1) * batch + sig) * 8;   // not touched by the last level
        bool odd_vals = false, own_nan = false;
        constexpr bool kOnePass = CAP >= TW;   // every tile fits the by-rank arrays: no pass loop, no per-group pass tests
        int rb = 0, g0 = 0;
        while (g0 < G2) {
            int g1 = g0, m = 0;
            if (kOnePass || c - rb <= CAP) {
                g1 = G2;
                m = c - rb;
            } else {
#pragma unroll 1
                for (int g = g0; g < G2; ++g) {
                    const int cg = (int)((gcnt >> (8 * g)) & 0xffull);
                    if (m + cg > CAP) break;
                    m += cg;
                    g1 = g + 1;
                }
            }
            const int nfp = min(3, (c - rb - m) + nf);
            CHAIN_MARK(20);
            if (c > 0) {
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    int p = 128 * g + 2 * lane, ki = kinfo[g];
                    asm volatile("" : "+v"(p), "+v"(ki));
                    const int ke = ki & 0xffff, bE = (ki >> 16) & 1, bO = (ki >> 17) & 1;
                    const int Le = ke - rb + 1;
                    const int Lo = Le + bO;
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
            wave_sync();
            CHAIN_MARK(21);
            if (g1 < G2 && lane < 2) { s_hX[5 + lane] = s_X[m + lane]; s_hI[5 + lane] = s_gi[m + lane]; }
            // ---- knot values, ITD.py:100-110 ----
            for (int L = 1 + lane; L <= m + 3; L += kWave) {
                const int32_t k0 = s_gi[L - 1], k1 = s_gi[L], k2 = s_gi[L + 1];
                const double x0 = s_X[L - 1], x1 = s_X[L], x2 = s_X[L + 1];
                const double frac = (double)(k1 - k0) / (double)(k2 - k0);
                const double tt = frac * (x2 - x0);
                const double uu = x0 + tt;
                double Bv = 0.5 * uu + 0.5 * x1;                                  // ITD.py:107-110
                const bool end0 = (L == 1) && (rb == 0) && (nb == 0);             // e[0]   = sample 0
                const bool endn = (L >= m + 2) && (L - (m + 2) >= nfp);           // e[m+1] = sample n-1
                Bv = end0 ? m0 : (endn ? mn : Bv);
                s_B[L] = Bv;
            }
            wave_sync();
            CHAIN_MARK(22);
            // ---- per-segment slope, ITD.py:115-116 ----
            for (int L = 1 + lane; L <= m + 2; L += kWave) {
                const double sl = (s_B[L + 1] - s_B[L]) / (s_X[L + 1] - s_X[L]);
                const bool endn = (L >= m + 2) && (L - (m + 2) >= nfp);
                if (!endn) s_S[L] = sl;
            }
            wave_sync();
            CHAIN_MARK(23);
            // ---- baseline at the two samples next to the tile (lane 0) ----
            if (lane == 0) {
                if (g0 == 0 && t >= 1) {
                    const double xlo = bits_d((unsigned)s_rb[0][kRecDwEdge + 2], (unsigned)s_rb[0][kRecDwEdge + 3]);
                    s_bl[0] = s_B[1] + s_S[1] * (xlo - s_X[1]);
                }
                if (g1 == G2 && rem > TW + 1) {   // sample s+TW exists and is not the signal's last
                    const int L = (nf >= 1 && s_hI[2] == si + TW) ? m + 2 : 1 + m;
                    const double xhi = bits_d((unsigned)s_rb[1][kRecDwEdge], (unsigned)s_rb[1][kRecDwEdge + 1]);
                    s_bl[1] = s_B[L] + s_S[L] * (xhi - s_X[L]);
                }
            }
            CHAIN_MARK(24);
            // ---- baseline map + rotation, ITD.py:114-119; the baseline replaces the input in the registers ----
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (g >= g0 && g < g1) {
                    int p = 128 * g + 2 * lane, ki = kinfo[g];
                    asm volatile("" : "+v"(p), "+v"(ki));
                    const int ke = ki & 0xffff, bO = (ki >> 17) & 1;
                    const int Le = ke - rb + 1, Lo = Le + bO;
                    const double xe = xr[g][0], xo = xr[g][1];
                    double be = s_B[Le] + s_S[Le] * (xe - s_X[Le]);
                    double bo = s_B[Lo] + s_S[Lo] * (xo - s_X[Lo]);
                    if (tail_tile) {   // baseline[n-1] is never written (stays 0), ITD.py:112-117
                        if (p >= nrem - 1) be = 0.0;
                        if (p + 1 >= nrem - 1) bo = 0.0;
                    }
                    const double re = fin ? (xe - be) + be : xe - be;   // last level: rotation + baseline, ITD.py:420
                    const double rq = fin ? (xo - bo) + bo : xo - bo;
                    xr[g][0] = be;
                    xr[g][1] = bo;
                    if (!(ITD_CHAIN_ABL & 1)) tile_store2<true>(r_rot, p * 8, re, rq);
                    double se = be, so = bo;
                    if (__builtin_expect(__any(nonfinite(be) || nonfinite(bo)), 0)) {   // rare: the stored baseline is the mutated one, ITD.py:50
                        odd_vals = true;
                        own_nan = own_nan || __any(__builtin_isunordered(be, bo));
                        se = be != be ? inf : be;
                        so = bo != bo ? inf : bo;
                    }
                    if constexpr (WANT_BASES) tile_store2<true>(r_bas, p * 8, se, so);   // extent 0 at the last level
                    if (!fin && (near_end || t == 0)) {   // the next level's end samples (of the mutated array), ITD.py:101-102
                        const unsigned long long ue = dbits(se), uo = dbits(so);
                        if (t == 0 && p == 0) {
                            gran_store(eg_next + 0, gen, (unsigned)ue); gran_store(eg_next + 1, gen, (unsigned)(ue >> 32));
                            gran_store(eg_next + 2, gen, (unsigned)uo); gran_store(eg_next + 3, gen, (unsigned)(uo >> 32));
                        }
                        if (p == nrem - 2) {
                            gran_store(eg_next + 4, gen, (unsigned)ue); gran_store(eg_next + 5, gen, (unsigned)(ue >> 32));
                            gran_store(eg_next + 6, gen, (unsigned)uo); gran_store(eg_next + 7, gen, (unsigned)(uo >> 32));
                        }
                        if (p + 1 == nrem - 2) { gran_store(eg_next + 4, gen, (unsigned)uo); gran_store(eg_next + 5, gen, (unsigned)(uo >> 32)); }
                        if (p == nrem - 1) { gran_store(eg_next + 6, gen, (unsigned)ue); gran_store(eg_next + 7, gen, (unsigned)(ue >> 32)); }
                    }
                }
            }
            rb += m;
            g0 = g1;
            wave_sync();
        }
        CHAIN_MARK(25);
        CHAIN_T(pt_map);
        CHAIN_ACC(4, pt_map - pt_front);   // passes + map + stores issued
        // ---- knots of the baseline just produced = the next level's input; the reference's NaN rules first where they apply ----
        {
            double x_lo = s_bl[0], x_hi = s_bl[1];
            if (__builtin_expect(odd_vals || __any(nonfinite(x_lo) || nonfinite(x_hi)), 0)) {
                const int delta = nan_rules<TW>(xr, x_lo, x_hi, (int64_t)si, nrem);
                if (lane == 0) {
                    if (delta) atomicAdd(&st->c_delta[level], delta);
                    if (own_nan) atomicOr(&st->nan_mask, 1 << level);
                }
            }
            const int64_t lv1 = (int64_t)(level + 1) * lv_tiles + gu;
            own_total = chain_publish<TW>(lane, xr, x_lo, x_hi, t == 0, nrem, fin, gen, cntg + lv1, recg + lv1 * 16, gran_copy,
                                          gsum + (int64_t)(level + 1) * gsum_level_pitch + grp, s_rec, s_pos, own);
        }
        wave_sync();
        CHAIN_MARK(33);
        CHAIN_T(pt_pub);
        CHAIN_ACC(5, pt_pub - pt_map);   // scan + publish
    }
    return 1;
    };   // step

    int status = 0;
    if (gu >= 0) {
        CHAIN_T(pt_s0);
        status = step();
        if (status < 0) return;
        CHAIN_T(pt_s1);
        if (status == 0) { CHAIN_ACC(0, pt_s1 - pt_s0); CHAIN_ACC(7, 1); }   // a failed attempt: polls only
        else if (++level == n_levels) { gu = -1; CHAIN_ACC(6, 1); }
    }
    if (gu < 0 && pgu < 0 && exhausted) break;
    if (ITD_CHAIN_CTX > 1) {   // the other tile's turn
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const double a0 = xr[g][0], a1 = xr[g][1];
            xr[g][0] = pxr[g][0]; xr[g][1] = pxr[g][1];
            pxr[g][0] = a0; pxr[g][1] = a1;
        }
        const WaveMasks wt = own; own = pown; pown = wt;
        int it = own_total; own_total = pown_total; pown_total = it;
        it = level; level = plevel; plevel = it;
        it = gu; gu = pgu; pgu = it;
    }
    if (status == 1) { idle = 0; sp = ChainSpin(); }
    else if (++idle >= ITD_CHAIN_CTX) {   // neither tile could move: pause, and give up if that lasts
        idle = 0;
        if (!chain_spin(sp, ctl)) return;
    }
  }
#if ITD_CHAIN_PROF
    if (lane_id() < 16) {
        const int k = lane_id();
        unsigned v = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) v = (k == j) ? prof_acc[j] : v;
        atomicAdd(&ctl->prof[k], (unsigned long long)v);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// k_chain_init: per call — tickets, give-up word, the generation tag, the group sums of every level, the signals' state.
// ---------------------------------------------------------------------------------------------
__global__ void k_chain_init(ChainCtl *ctl, SigState *st, int batch, int32_t *gsum, int64_t gsum_elems)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < gsum_elems; i += (int64_t)gridDim.x * blockDim.x) gsum[i] = 0;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < 8) ctl->ticket[b][0] = 0;
    if (b == 8) ctl->give_up = 0;
    if (b == 9) { unsigned g = ctl->gen + 1u; ctl->gen = g ? g : 1u; }
    if (b >= batch) return;
    for (int j = 0; j < kMaxLevels; ++j) { st[b].m[j] = -1; st[b].c_delta[j] = 0; }
    st[b].stop_level = -1;
    st[b].nan_mask = 0;
    st[b].in_nan = 0;
    st[b].l0_fail = 0;
    st[b].fin_stopped = 0;
    st[b].fin_stop_level = -1;
    st[b].chain_stop = 0;
}

// ---------------------------------------------------------------------------------------------
// k_chain_finalize: the levels' knot totals m[0 .. n_levels] under the reference's counting rules (NaN corrections as in
// k_extract / k_finalize) and the stop rule after the fact: if any pending baseline inside the run had < 2 knots
// (ITD.py:404) the chain's rows past that point are not the reference's — chain_stop tells the engine to repeat the call
// level by level (which performs the row fix-up of ITD.py:404-416).  grid = batch, 256 threads.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chain_finalize(const int32_t *__restrict__ gsum, int64_t gsum_level_pitch, int n_tiles,
                                                         int n_levels, SigState *__restrict__ state)
{
    __shared__ int s_red[4];
    const int sig = blockIdx.x;
    SigState *st = state + sig;
    const int n_groups = groups_of(n_tiles);
    int stop = 0;
    for (int j = 0; j <= n_levels; ++j) {
        const int32_t *gs = gsum + (int64_t)j * gsum_level_pitch + (size_t)sig * n_groups * kGsumPitch;
        int acc = 0;
        for (int k = threadIdx.x; k < n_groups; k += 256) acc += gs[(size_t)k * kGsumPitch];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
        __syncthreads();
        int mj = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        if (j >= 1 && ((st->nan_mask >> (j - 1)) & 1)) mj += st->c_delta[j - 1];
        if (threadIdx.x == 0) st->m[j] = mj;
        if (j >= 1 && mj < 2) stop = 1;
    }
    if (threadIdx.x == 0) st->chain_stop = stop;
}

}  // namespace itd
