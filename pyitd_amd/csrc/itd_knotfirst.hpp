// itd_knotfirst.hpp — the sparse levels of a decomposition, fused ("knot first").
//
// One extraction (ITD.py:79-121) maps every sample through an affine function of ITSELF inside its segment,
//     baseline[i] = B_k + S_k * (x[i] - x[e_k])        for i in [e_k, e_k+1)                              (ITD.py:114-117)
// so between two consecutive knots the baseline is a monotone image of the (monotone) input: the NEXT level's knots can only sit
// at this level's knots.  Hence the level recursion of the driver (ITD.py:384-432) can run on the KNOT LIST alone — a few per
// cent of the samples from level 2 or 3 on — and the samples need ONE pass for all of those levels instead of one pass (8 B
// read + 16 B written per sample) per level:
//
//   knot side   k_kf_knots, ONE launch for all fused levels.  A candidate carries the level's values at (e-1, e, e+1) and two
//               flags: it is a knot of the level; it is "sticky".  A level's step computes B_k and S_k of every knot from the
//               knots' positions and values (ITD.py:100-116), writes them to the level's table, moves the three values of every
//               candidate to the next level through the maps of the segments they lie in, and decides with the knot predicate
//               (ITD.py:59 on x and on -x) which candidates are knots of the next level.  Knots and sticky candidates stay in
//               the list.  Sticky for ever: sample n-2 (its right neighbour is forced to 0: baseline[n-1] is never written,
//               ITD.py:112-117), both samples of every NEAR tie of the first fused level's input (near_tie(): two neighbours
//               within 2^-20 of their magnitude — exact ties of quantised data, ties that the first levels have grown or broken
//               by an ulp, neighbours close enough to collapse a few levels on), and every candidate whose triple ever shows a
//               near tie — the places where rounding makes or breaks a plateau and a knot can appear where the previous level
//               had none.  The stop rules (ITD.py:400-426) are decided here, from the list sizes.
//               A workgroup owns kKcTiles consecutive tiles of the signal and keeps its part of the candidate list in LDS from
//               the hand-over to the last level; all it needs from outside at a level are the two knots in front of its range
//               and the three behind it, which its neighbours publish as records of data-tagged 8-byte granules (one sc1 store
//               each, polled with sc1 loads: no flags, no fences) — a nearest-neighbour exchange, never a grid barrier.
//   sample side k_kf_apply: a wavefront takes a 512-sample tile of the first fused level's input, and for every fused level
//               looks its tile's knots up in the level's table (positions as flag words, B / S / X by rank), maps its samples,
//               stores the rotation row, and RE-DERIVES the next level's knots from the values it has just computed — the exact
//               predicate on the actual samples.  Any difference from the knot side's flags means the shortcut missed a knot
//               (a plateau born from rounding somewhere else): the signal's result is discarded (SigState::kf_fail) and the
//               engine repeats the call level by level.  It also VERIFIES every table value it maps its samples with (V0 .. V3 in
//               front of k_kf_apply below: positions and values against its own samples, B and S recomputed, the runs' chain, the
//               levels' knot counts), so whatever this path delivers is, by induction over the levels and by construction, bit
//               for bit what the level-by-level engine delivers; what it cannot deliver it reports.
//
// Traffic of the fused levels L0 .. L: 8 B read + 8 B per row written per sample, against 24 B per sample and level.
// A numpy statement of exactly this algorithm lives with the tests.
#pragma once
#include "itd_kernels.hpp"

#pragma clang fp contract(off)

namespace itd {

constexpr int kKfCap = 128;        // knots of one tile and level the sample pass holds by rank

// ---- geometry of the knot side's launch ----
#ifndef ITD_KC_THREADS
#define ITD_KC_THREADS 512        // eight wavefronts per knot-side workgroup, two candidates per thread (A/B builds: 256 = round 4's four: 70.9
                                  // against 68.6 us per launch at 2^24 fused from level 2, 98.1 against 95.7 us per decomposition at 2^20)
#endif
constexpr int kKcThreads = ITD_KC_THREADS;       // threads of a workgroup
constexpr int kKcWaves = kKcThreads / 64;
constexpr int kKcTiles = 64;                     // tiles a workgroup owns at most (KfWs::tpw; one thread per 128-sample group of them)
constexpr int kKcCap = 1024;                     // candidates a workgroup holds from the second fused level on (more: the signal is left to the level-by-level engine)
constexpr int kKcCapH = 1720;                    // ... and at the hand-over level, whose candidates' triples stay in registers (KcLay)
constexpr int kKcEnt = (kKcCap + kKcThreads - 1) / kKcThreads;      // candidates per thread
constexpr int kKcEntH = (kKcCapH + kKcThreads - 1) / kKcThreads;
constexpr int kKcRound = 2;                      // workgroups whose records a halo search's first round reads (the nearest ones, whatever they hold)
constexpr int kKcSlab = 3200;                    // table entries (all fused levels) a workgroup may write: ~1.7 x the hand-over level's knots
// The workgroup's LDS, two layouts over the same bytes.  From the second fused level on (H = false) a candidate is its position, its
// triple of values, flags and rank (31 B), a knot by rank its position, value, B and S (28 B): 59 B x kKcCap.  The hand-over level is the
// densest by a factor of ~2.6 and its candidates' triples come straight from the level's input in memory: they stay in the registers of
// the threads that fetched them until that level's maps have used them — the hand-over layout (H = true) has no triples and holds
// kKcCapH candidates in the same 60 KB.  (A level's step reads the old lists into registers, crosses a barrier and only then writes the
// next lists: the change of layout happens at that barrier.)
template <bool H>
struct KcLay {
    static constexpr int cap = H ? kKcCapH : kKcCap;
    static constexpr int o_xl = 0, o_xc = 8 * kKcCap, o_xr = 16 * kKcCap;       // (H: none)
    static constexpr int o_kX = H ? 0 : 24 * kKcCap;
    static constexpr int o_kB = o_kX + 8 * (cap + 8), o_kS = o_kB + 8 * (cap + 8), o_kpos = o_kS + 8 * (cap + 8);
    static constexpr int o_cpos = o_kpos + 4 * (cap + 8), o_crk = o_cpos + 4 * cap, o_cfl = o_crk + 2 * cap;
    static constexpr int end = o_cfl + cap;
    unsigned char *base;
    __device__ __forceinline__ double *c_xl() const { return reinterpret_cast<double *>(base + o_xl); }
    __device__ __forceinline__ double *c_xc() const { return reinterpret_cast<double *>(base + o_xc); }
    __device__ __forceinline__ double *c_xr() const { return reinterpret_cast<double *>(base + o_xr); }
    __device__ __forceinline__ double *k_X() const { return reinterpret_cast<double *>(base + o_kX); }
    __device__ __forceinline__ double *k_B() const { return reinterpret_cast<double *>(base + o_kB); }
    __device__ __forceinline__ double *k_S() const { return reinterpret_cast<double *>(base + o_kS); }
    __device__ __forceinline__ int32_t *k_pos() const { return reinterpret_cast<int32_t *>(base + o_kpos); }
    __device__ __forceinline__ int32_t *c_pos() const { return reinterpret_cast<int32_t *>(base + o_cpos); }
    __device__ __forceinline__ unsigned short *c_rk() const { return reinterpret_cast<unsigned short *>(base + o_crk); }
    __device__ __forceinline__ unsigned char *c_fl() const { return base + o_cfl; }
};
constexpr int kKcLds = ((KcLay<true>::end > KcLay<false>::end ? KcLay<true>::end : KcLay<false>::end) + 15) & ~15;
static_assert(KcLay<true>::o_kB - KcLay<true>::o_kX >= kKcTiles * 8 * 8, "the sticky words fit the hand-over layout's k_B");
constexpr int kKcRecGran = 32;                   // 8-byte granules per record slot (20 in use): 256 bytes
constexpr uint32_t kKcPoison = 0xffffffffu;      // a record of a workgroup that has given up
#ifndef ITD_ROW_PAD
#define ITD_ROW_PAD 0             // experiment builds only (profiles/r05/experiments): the result's rows ITD_ROW_PAD elements further apart than n
#endif
#ifndef ITD_KC_GATHER_ORDER
#define ITD_KC_GATHER_ORDER 1     // A/B builds: 0 = the hand-over's gathers in candidate order
#endif
#ifndef ITD_KF_CHECK_STEP
#define ITD_KF_CHECK_STEP 2       // A/B builds: 1 = one table entry per lane and round in the check wavefronts
#endif
#ifndef ITD_KC_PRIO
#define ITD_KC_PRIO 0
#endif
#ifndef ITD_KC_TIMEOUT
#define ITD_KC_TIMEOUT 4000000ll                 // ticks of the 100 MHz wall clock a halo search waits at most (40 ms)
#endif

struct KfEntry { double X, B, S; int32_t pos; int32_t pad; };   // a level's table: the knot in front of a tile range, the range's knots, the knot behind
static_assert(sizeof(KfEntry) == 32, "table entries are two 16-byte words");

// failure bits (KfSig::fail, SigState::kf_fail)
constexpr int kKfFailVerify = 1, kKfFailCapacity = 2, kKfFailNonFinite = 4, kKfFailTies = 8;
constexpr int kKfFailWait = 16;   // a halo wait was given up (a neighbour that never became resident within ITD_KC_TIMEOUT) or met a neighbour that had given up

struct KfSig {
    // ---- the head: what itd_get_summary copies to the host (kKfSigHead bytes).  Written with plain stores by the signal's last
    //      knot-side workgroup (workgroup 0 for a signal the fused levels do not run for); the sample pass ORs into `fail` and adds
    //      to `m_exact` behind that ----
    int32_t fail;
    int32_t active;       // the fused levels run for this signal (it had not stopped before the hand-over, no NaN anywhere)
    int32_t lend;         // the last fused level: rows 0 .. lend are the result (-1: the signal is not active)
    int32_t natural;      // at lend: 1 = natural stop (row lend = the level's input), 0 = "Out of time!" (rotation + baseline)
    int32_t m_exact;      // knots of the last pending baseline, counted by the sample pass
    int32_t fail_lev;     // the lowest level at which anything failed (99: nowhere): verification of that level's knots or tables by the sample
                          // pass, non-finite knot data — what the engine's next calls cap their fused levels at (KfWs::cap)
    int32_t cont;         // 1: the fused levels were capped (KfWs::cap) and did not stop: rows 0 .. lend are rotations, the baseline behind
                          // level lend is in memory and the level launches go on from there
    int32_t pad0[1];
    int32_t mlev[kMaxLevels + 2];   // knots of level j's input
    // ---- the knot side's accumulators: zero between calls (the signal's last workgroup reads and clears them; the workspace starts
    //      zeroed), so no launch has to prepare them ----
    int32_t acc_mlev[kMaxLevels + 2];
    int32_t acc_fail;
    int32_t nf;           // 64 - (the first level whose knot data was not finite); 0 = all finite
    int32_t done;         // workgroups of the signal that have finished
    int32_t ticket;       // (first signal of a launch that hands out tickets) the launch's workgroup tickets ...
    int32_t fin;          // ... and its finished workgroups: the last one clears both
    int32_t pad1;
    uint32_t seq;         // the generation in the tags of a call's records: bumped by the signal's last workgroup
    uint32_t pad2;
};
constexpr size_t kKfSigHead = (8 + (kMaxLevels + 2)) * sizeof(int32_t);
static_assert(offsetof(KfSig, acc_mlev) == kKfSigHead, "KfSig layout");

// what the knot side's launch needs to do k_finalize's work (itd_kernels.hpp) for the hand-over level: the stop test of that level's
// input with its row fix-up, and the other set of states / group sums left initialised for the call after this one
struct KfFin {
    double *rows; int64_t rows_stride;
    const double *bases; int64_t bases_stride, bases_row_pitch; int32_t bases_rotate;
    const int32_t *gsum;                 // the hand-over level's group sums
    SigState *other_state; int32_t *other_gsum; int64_t other_third;
};

struct KfWs {
    KfSig *sig;                   // [batch]
    KfEntry *pool;                // [batch][wgs_max][kKcSlab]       a workgroup's tables, level after level
    int32_t *first;               // [batch][nlev][tiles]            pool index (within the signal) of the table entry in front of the tile's knots
    unsigned long long *tflags;   // [batch][nlev][tiles][8]         the level's knots as flag words (flag_pos format)
    unsigned long long *nearw;    // [batch][tiles][8]               both samples of every near tie of the first fused level's input, as flag words (written
                                  //                                 by the launch that wrote that input: k_extract's TIES form)
    unsigned long long *rec;      // [batch][rec_levels][wgs_max][kKcRecGran]   boundary records
    int32_t wgs_max, rec_levels;  // strides of pool / rec
    int32_t nlev, n_tiles, L0;    // this call's geometry
    int32_t wgs, nb;              // workgroups per signal, signals of the launch
    int32_t tpw;                  // tiles per workgroup (<= kKcTiles; fewer where the first fused level's knots are dense)
    int32_t ticketed;             // 1: the grid may not fit the device at once: workgroup ids are tickets (KfSig::ticket), not blockIdx
    // fault injection (itd_debug_kf_fault, tests only): kind 6 / 7 = a halo knot's value / position as this launch's workgroup `wg`
    // receives it at level `lev` is perturbed (kinds 0 .. 5 are applied to the workspace by k_kf_fault between the two launches)
    int32_t dbg_kind, dbg_lev, dbg_wg, dbg_slot, dbg_delta;
    int32_t dbg_sig;              // ... of signal dbg_sig of the launch
    // the batch pipeline (itd_engine.hip, enqueue_decompose): every knot-side workgroup counts itself here when it starts — a counter that
    // only ever grows; k_kf_gate on the memory-bound launches' stream returns when all of a launch's workgroups are resident.  NULL: not counted
    unsigned long long *started;
    // Capped fused levels (partial fusion): cap != 0 = the fused levels end at level cap - 1 although the call asks for more — a workload
    // whose fused form fails at level cap every time (periodic input whose baseline collapses there) keeps the fused form for the levels
    // in front of it.  The sample pass then stores the baseline behind level cap - 1 (the input of level cap) at xnext and the engine
    // goes on level by level from a scan of it (k_scan0).
    int32_t cap;
    double *xnext; int64_t xnext_stride;
};

__device__ __forceinline__ bool kf_pred(double yl, double yc, double yr)
{
    const double dp = yc - yl, dn = yr - yc;
    return ((dn > 0.0) && (dp <= 0.0)) || ((dn < 0.0) && (dp >= 0.0));      // ITD.py:59 on x and on -x
}

// a data-tagged granule: {data, tag} in one naturally aligned 8-byte word, written by ONE store and read by ONE load (both sc1:
// past the L1s, never torn) — the reader needs no flag and no fence: a granule whose tag is this call's and this level's IS the datum
__device__ __forceinline__ void kc_store(unsigned long long *p, uint32_t data, uint32_t tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long kc_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A workgroup barrier for LDS hand-overs only: __syncthreads() also waits for every global store and atomic the wavefront has in
// flight (its fence covers global memory: s_waitcnt vmcnt(0)) — here that would put the write-through record stores and the table
// stores, which no thread of the workgroup ever reads back, on every level's critical path (~1.5 us each).
__device__ __forceinline__ void kc_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// exclusive prefix of v over the workgroup's threads (in thread order) and the total
__device__ __forceinline__ int kc_scan(int v, int *s_red, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
    kc_barrier();                                   // (the previous scan's reads of s_red are done)
    if (lane == 63) s_red[wave] = inc;
    kc_barrier();
    int wb = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kKcThreads / 64; ++k) { const int s = s_red[k]; if (k < wave) wb += s; tot += s; }
    total = tot;
    return wb + inc - v;
}

// A workgroup's record of a level, for its neighbours: granule 0 the number of its knots; 1..3 / 4..6 its second-last / last
// knot (position, value's low / high word); 7..15 its first three knots; 16..19 two doubles: workgroup 0 publishes the level's
// x[0], x[1], the last workgroup x[n-2], x[n-1] (ITD.py:101-102).  One wavefront, lanes 0 .. 19; dense index = rank + 1.
__device__ __forceinline__ void kc_publish(unsigned long long *slot, uint32_t tag, int lane, int ck, const int32_t *k_pos, const double *k_X,
                                           double ea, double eb, bool poison)
{
    if (lane >= 20) return;
    uint32_t data = 0;
    if (poison) data = kKcPoison;
    else if (lane == 0) data = (uint32_t)ck;
    else if (lane <= 15) {
        const int q = (lane - 1) / 3, f = (lane - 1) % 3;                    // q: 0 second last, 1 last, 2..4 the first three
        const int rank = q == 0 ? ck - 1 : (q == 1 ? ck : q - 1);
        if (rank >= 1 && rank <= ck) {
            if (f == 0) data = (uint32_t)k_pos[rank + 1];
            else { const unsigned long long b = dbits(k_X[rank + 1]); data = f == 1 ? (uint32_t)b : (uint32_t)(b >> 32); }
        }
    } else {
        const unsigned long long b = dbits(lane < 18 ? ea : eb);
        data = (lane & 1) ? (uint32_t)(b >> 32) : (uint32_t)b;
    }
    kc_store(slot + lane, data, tag);
}

// ITD_PROF=1 (diagnostic build, tools/knots_prof.py): the 100 MHz wall clock at a workgroup's phase boundaries, 64 marks per workgroup
#if ITD_PROF
__device__ unsigned long long *g_kc_prof;   // [workgroups][64]
#define KC_MARK(i) do { if (threadIdx.x == 0 && g_kc_prof) kc_marks[(i)] = (unsigned long long)wall_clock64(); } while (0)
#define KC_MARK_T(t, i) do { if (threadIdx.x == (t) && g_kc_prof) g_kc_prof[(size_t)blockIdx.x * 64 + (i)] = (unsigned long long)wall_clock64(); } while (0)   /* marks 52 .. 57: other wavefronts' */
#else
#define KC_MARK(i)
#define KC_MARK_T(t, i)
#endif
// ---- the knot side: hand-over and every fused level in ONE launch.  grid = wgs * nb workgroups of kKcThreads (512) threads; workgroup id ->
//      (signal, range of tpw tiles).  A workgroup waits for its neighbours' records, so it must never keep a neighbour from
//      starting: a grid that fits the device at once (KfWs::ticketed = 0) takes the id from blockIdx; a larger one hands out
//      tickets — whoever starts first takes the lowest range, so whatever a resident workgroup waits for has started already
//      or starts without that workgroup finishing first (dependencies reach a few ranges per level, the device holds hundreds).
template <int TW>
__global__ __launch_bounds__(kKcThreads) void k_kf_knots(KfWs ws, KfFin fin, const double *__restrict__ xl, int64_t xl_stride, int64_t n, int max_iteration,
                                                         const int32_t *__restrict__ counts, const TileRec *__restrict__ recs,
                                                         SigState *__restrict__ state)
{
    static_assert(TW == 512 && kKcTiles * (TW / 128) <= kKcThreads && kKcTiles == 64 && (kKcThreads == 256 || kKcThreads == 512),
                  "one thread per 128-sample group (the first 256 threads), one lane per tile");
    constexpr int HT = kKcTiles * (TW / 128);              // threads that take part in the hand-over's expansion: one per 128-sample group
    // the candidates: position, the level's values at position - 1, position, position + 1,
    // flags, and the number of the range's knots at or before the candidate
    // (c_fl: 1 = a knot of the level, 2 = sticky.  The level's knots by rank, dense index = rank + 1: ranks -1, 0 the two knots in
    //  front of the range, 1 .. ck its own, ck + 1 .. ck + 3 the three behind it)
    __shared__ __attribute__((aligned(16))) unsigned char s_lds[kKcLds];
    __shared__ unsigned long long t_w[kKcTiles * 8];       // the level's knots as the tiles' flag words
    __shared__ uint32_t s_stage[2][4][16];
    __shared__ int s_red[8], s_i[8], s_cnt[kKcEntH * kKcWaves];
    const KcLay<true> LH{s_lds};
    const KcLay<false> LN{s_lds};
    __shared__ double s_ends[4];                           // the level's x[0], x[1], x[n-2], x[n-1] (as far as this workgroup needs them)
    constexpr int NT = kKcThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if ITD_PROF
    unsigned long long kc_marks[64];
    for (int i = 0; i < 64; ++i) kc_marks[i] = 0;
#endif
    KC_MARK(0);
#if ITD_KC_PRIO
    __builtin_amdgcn_s_setprio(ITD_KC_PRIO);       // A/B builds: the knot side's wavefronts issue in front of a co-resident launch's (batches)
#endif
    if (ws.started && tid == 0) atomicAdd(ws.started, 1ull);      // (in front of every return: the gate counts workgroups, whatever they go on to do)
    int id = blockIdx.x;
    if (ws.ticketed) {
        if (tid == 0) s_i[0] = atomicAdd(&ws.sig[0].ticket, 1);
        kc_barrier();
        id = s_i[0];
    }
    KC_MARK(1);
    const int W = ws.wgs;
    const int sig = id / W, w = id - sig * W;
    if (sig >= ws.nb) return;
    KfSig *ks = ws.sig + sig;
    SigState *st = state + sig;
    const int n_tiles = ws.n_tiles, L0 = ws.L0, M = max_iteration;
    const int tpw = ws.tpw;
    const int t0 = w * tpw, t1 = min(n_tiles, t0 + tpw), nt = t1 - t0;
    const int32_t n1 = (int32_t)(n - 1);
    // a launch that hands out tickets counts its finished workgroups: the last one clears both counters for the next call
    auto launch_done = [&]() {
        if (ws.ticketed && tid == 192 && atomicAdd(&ws.sig[0].fin, 1) == W * ws.nb - 1) { ws.sig[0].ticket = 0; ws.sig[0].fin = 0; }
    };
    // everything the hand-over needs is requested before the first use: one round trip
    unsigned long long w_in[2], s_in[2];
    int c_in[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int q = tid + r * HT, t = t0 + (q >> 3);
        w_in[r] = s_in[r] = 0ull; c_in[r] = 0;
        if (tid < HT && t < t1) {
            c_in[r] = counts[(size_t)sig * n_tiles + t];
            w_in[r] = recs[(size_t)sig * n_tiles + t].flags[q & 7];
            s_in[r] = ws.nearw[((size_t)sig * n_tiles + t) * 8 + (q & 7)];
        }
    }
    const int n_groups = groups_of(n_tiles);
    int m_part = 0;                                                          // this level's knots: the launch in front left group sums
    {
        const int32_t *gs = fin.gsum + (size_t)sig * n_groups * kGsumPitch;
        for (int g = tid; g < n_groups; g += NT) m_part += gs[(size_t)g * kGsumPitch];
    }
    const double end_in = st->ends[L0 & 1][tid & 3];
    const uint32_t seq = ks->seq;
    int stop_level = st->stop_level;
    const int nan_mask = st->nan_mask, c_delta = L0 >= 1 ? st->c_delta[L0 - 1] : 0;
    const bool odd_input = nan_mask != 0 || st->in_nan || st->l0_fail;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m_part += __shfl_xor(m_part, d);
    if (lane == 0) s_red[wave] = m_part;
    if (tid == 0) s_i[1] = 0;
    kc_barrier();
    // ---- what k_finalize does behind a level-by-level call (ITD.py:400-416): the stop test of the pending baseline — here the first
    //      fused level's input —, the residual row of a signal that has stopped, the other set of states left ready for the next call
    bool stopped = stop_level >= 0;
    if (!stopped) {
        int m_last = 0;
#pragma unroll
        for (int k = 0; k < kKcWaves; ++k) m_last += s_red[k];
        if (L0 >= 1 && ((nan_mask >> (L0 - 1)) & 1)) m_last += c_delta;     // the baseline held a NaN: counted under the NaN rules (nan_rules)
        if (m_last < 2) { stopped = true; stop_level = L0; }
        if (w == 0 && tid == 0) st->m[L0] = m_last;
    }
    if (w == 0) {
        if (tid == 0) { st->fin_stopped = stopped ? 1 : 0; st->fin_stop_level = stop_level; }
        if (fin.other_state) {
            if (tid == 0) sig_state_reset(fin.other_state + sig);
            int32_t *og = fin.other_gsum + (size_t)sig * n_groups * kGsumPitch;
            for (int k = tid; k < 3 * n_groups; k += NT) {
                int32_t *slot = og + (size_t)(k / n_groups) * fin.other_third + (size_t)(k % n_groups) * kGsumPitch;
                slot[0] = 0;
                slot[1] = 0;
            }
        }
    }
    if (stopped) {      // row c = stop_level - 1 of the result is baselines[c-1], or zeros when c = 0: this workgroup's samples of it
        const int c_row = stop_level - 1;
        double *dst = fin.rows + (int64_t)sig * fin.rows_stride + (int64_t)c_row * n;
        const double *src = nullptr;
        if (c_row >= 1) {
            const int row = fin.bases_rotate ? ((c_row - 1) % fin.bases_rotate) : (c_row - 1);
            src = fin.bases + (int64_t)sig * fin.bases_stride + (int64_t)row * fin.bases_row_pitch;
        }
        const int64_t lo = (int64_t)t0 * TW, hi = min((int64_t)t1 * TW, n);
        for (int64_t i = lo + tid; i < hi; i += NT) dst[i] = src ? src[i] : 0.0;
    }
    if (stopped || odd_input) {                                              // (every workgroup of the signal sees the same)
        if (w == 0 && tid == 0) {
            ks->active = 0;
            ks->lend = -1;
            ks->fail = stopped ? 0 : kKfFailNonFinite;                        // NaN rules / an unfinished level 0: not this path's
            ks->fail_lev = 99;
            ks->cont = 0;
            if (ws.cap && stopped) st->skip = 1;                              // (capped: the level launches behind the sample pass have nothing to do)
        }
        launch_done();
        return;
    }
    unsigned long long *rec_sig = ws.rec + ((size_t)sig * ws.rec_levels) * ws.wgs_max * kKcRecGran;
    KfEntry *slab = ws.pool + ((size_t)sig * ws.wgs_max + w) * kKcSlab;
    const int slab_off = w * kKcSlab;
    int32_t *first_sig = ws.first + (size_t)sig * ws.nlev * n_tiles;
    unsigned long long *tf_sig = ws.tflags + (size_t)sig * ws.nlev * n_tiles * 8;
    const double *xs = xl + (int64_t)sig * xl_stride;
    auto rec_slot = [&](int lev, int v) { return rec_sig + ((size_t)(lev - L0) * ws.wgs_max + v) * kKcRecGran; };
    auto rec_tag = [&](int lev) { return seq * 64u + (uint32_t)(lev - L0 + 1); };
    auto give_up = [&](int code) { s_i[1] = 1; atomicOr(&ks->acc_fail, code); };

    // ---- hand-over: the level-L0 knots of the range (flag words of the records the launch for level L0 - 1 left); sticky
    //      candidates: sample n-2, both samples of every near tie of the level's input (near_tie(): flag words the launch that
    //      wrote it left)
    unsigned long long *s_w = reinterpret_cast<unsigned long long *>(LH.k_B());   // (sticky words: until the first level needs k_B)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int q = tid + r * HT;
        if (tid < HT) {
            t_w[q] = c_in[r] > 0 ? w_in[r] : 0ull;                           // (an empty tile's record holds no words)
            s_w[q] = s_in[r];
        }
    }
    if (tid < 4) s_ends[tid] = end_in;
    if (w == 0 && tid == 0) ks->active = 1;
    kc_barrier();
    KC_MARK(2);
    if (tid == 0) {
        const int64_t p = n - 2;
        const int tp = (int)(p / TW);
        if (p >= 1 && tp >= t0 && tp < t1) {
            const int q = (int)(p - (int64_t)tp * TW);
            s_w[(tp - t0) * 8 + 2 * (q >> 7) + (q & 1)] |= 1ull << ((q & 127) >> 1);
        }
    }
    kc_barrier();
    int c = 0, ck = 0;                                                       // candidates of the range; knots among them
    {
        const int tt = (tid >> 2) & (kKcTiles - 1), g = tid & 3;
        const bool mine = tid < HT;                                          // (more threads than groups: the others hold no words)
        const unsigned long long Ek = mine ? t_w[tt * 8 + 2 * g] : 0ull, Ok = mine ? t_w[tt * 8 + 2 * g + 1] : 0ull;
        const unsigned long long Es = mine ? s_w[tt * 8 + 2 * g] : 0ull, Os = mine ? s_w[tt * 8 + 2 * g + 1] : 0ull;
        unsigned long long E = Ek | Es, O = Ok | Os;
        int tot;
        int o = kc_scan((__popcll(E) + __popcll(O)) | ((__popcll(Ek) + __popcll(Ok)) << 16), s_red, tot);
        c = tot & 0xffff; ck = tot >> 16;
        int r = o >> 16;                                                     // knots in front of the thread's group
        o &= 0xffff;
        if (c <= kKcCapH) {
            int32_t *c_pos = LH.c_pos(), *k_pos = LH.k_pos();
            unsigned char *c_fl = LH.c_fl();
            unsigned short *c_rk = LH.c_rk();
            const int32_t s = (int32_t)((t0 + tt) * TW + 128 * g);
            while (E | O) {                                                  // within a group the candidates run E_0, O_0, E_1, O_1, ...
                const int l = __builtin_ctzll(E | O);
                const unsigned long long bit = 1ull << l;
                if (E & bit) {
                    const int kf = (Ek & bit) ? 1 : 0;
                    r += kf;
                    c_pos[o] = s + 2 * l; c_fl[o] = (unsigned char)(kf | ((Es & bit) ? 2 : 0)); c_rk[o] = (unsigned short)r;
                    if (kf) k_pos[r + 1] = s + 2 * l;
                    ++o;
                }
                if (O & bit) {
                    const int kf = (Ok & bit) ? 1 : 0;
                    r += kf;
                    c_pos[o] = s + 2 * l + 1; c_fl[o] = (unsigned char)(kf | ((Os & bit) ? 2 : 0)); c_rk[o] = (unsigned short)r;
                    if (kf) k_pos[r + 1] = s + 2 * l + 1;
                    ++o;
                }
                E &= ~bit; O &= ~bit;
            }
        } else if (tid == 0) give_up(kKfFailCapacity);
    }
    kc_barrier();
    // the three values around every candidate, one round trip; they stay in these registers until the first level's maps have used them
    double va[kKcEntH], vb[kKcEntH], vd[kKcEntH];
#pragma unroll
    for (int i = 0; i < kKcEntH; ++i) va[i] = vb[i] = vd[i] = 0.0;
    if (!s_i[1] && c > 0) {
        // (every load unconditional — a thread without a candidate in a round repeats the list's last one — so that the compiler counts
        //  them: the knots' values by rank, which the first level's knot values need, wait for the middle samples only; the outer two
        //  arrive while the halo exchange runs.  At level 2 these gathers touch about half of the input's lines.)
        int32_t pp[kKcEntH];
#pragma unroll
        for (int i = 0; i < kKcEntH; ++i) pp[i] = LH.c_pos()[min(i * NT + tid, c - 1)];
#if ITD_KC_GATHER_ORDER
#pragma unroll
        for (int i = 0; i < kKcEntH; ++i) vb[i] = xs[pp[i]];
#pragma unroll
        for (int i = 0; i < kKcEntH; ++i) { va[i] = xs[pp[i] - 1]; vd[i] = xs[pp[i] + 1]; }
#else       // (A/B builds: the three samples of a candidate together, as up to round 4)
#pragma unroll
        for (int i = 0; i < kKcEntH; ++i) { va[i] = xs[pp[i] - 1]; vb[i] = xs[pp[i]]; vd[i] = xs[pp[i] + 1]; }
#endif
#pragma unroll
        for (int i = 0; i < kKcEntH; ++i) {
            const int j = i * NT + tid;
            if (j < c && (LH.c_fl()[j] & 1)) LH.k_X()[LH.c_rk()[j] + 1] = vb[i];
        }
        if (tid == 192) atomicAdd(&ks->acc_mlev[L0], ck);
    }
    kc_barrier();
    KC_MARK(3);

    // ---- the levels.  At the top of a level the candidates, the knots' positions and values by rank (k_pos, k_X) and the tiles'
    //      flag words (t_w) are in place: the hand-over's for the first level, the previous level's compaction for the others.
    int toff = 0;                     // table entries this workgroup has written
    int published = L0 - 1;           // the last level whose record is out
    // one level; `lay`: the layout its lists are in (the hand-over's for the first, whose triples are va / vb / vd).  false = given up
    auto level = [&](auto lay, const int lev) -> bool {
        using Lay = decltype(lay);
        constexpr bool FIRST = Lay::cap == kKcCapH;
        constexpr int ENT = FIRST ? kKcEntH : kKcEnt;
        double *const k_X = lay.k_X(), *const k_B = lay.k_B(), *const k_S = lay.k_S();
        int32_t *const k_pos = lay.k_pos();
        const int32_t *const c_pos = lay.c_pos();
        const unsigned short *const c_rk = lay.c_rk();
        const unsigned char *const c_fl = lay.c_fl();
        const int li = lev - L0;
        if (toff + ck + 2 > kKcSlab) { if (tid == 0) give_up(kKfFailCapacity); kc_barrier(); return false; }
        KC_MARK(4 + 4 * min(li, 12));
        // A. the record for the neighbours (wavefront 2); the tiles' structures of this level for the sample pass (wavefront 3);
        //    the knots around the range from the neighbours' records (wavefront 0: in front, wavefront 1: behind)
        if (wave == 2) {
            if (lev == L0) {                                                 // (the later levels' records leave from the compaction below)
                const bool head = w == 0, tail = w == W - 1;
                kc_publish(rec_slot(lev, w), rec_tag(lev), lane, ck, k_pos, k_X, head ? s_ends[0] : (tail ? s_ends[2] : 0.0),
                           head ? s_ends[1] : (tail ? s_ends[3] : 0.0), false);
            }
        } else if (wave == 3) {
            int cnt = 0;
            unsigned long long wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { wv[q] = t_w[lane * 8 + q]; cnt += __popcll(wv[q]); }
            int inc = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
            // (the level's knot count — what the stop rules are drawn from — is the population of the words the sample pass verifies)
            if (lane == nt - 1 && inc != ck) give_up(kKfFailVerify);
            if (lane < nt) {
                first_sig[(size_t)li * n_tiles + t0 + lane] = slab_off + toff + inc - cnt;
                using U2 = unsigned long long __attribute__((ext_vector_type(2)));
                U2 *dst = reinterpret_cast<U2 *>(tf_sig + ((size_t)li * n_tiles + t0 + lane) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const U2 v = {wv[2 * q], wv[2 * q + 1]}; dst[q] = v; }
            }
            if (li == 1) KC_MARK_T(192, 53);
        } else if (wave < 2) {
            const int side = wave;                                           // 0: in front (two knots), 1: behind (three)
            const int want = side == 0 ? 2 : 3;
            const int a = lane >> 4, g = lane & 15;
            const uint32_t tag = rec_tag(lev);
            const int v_end = side == 0 ? 0 : W - 1;                         // the workgroup that publishes this side's end samples
            auto nb = [&](int dist) { return side == 0 ? w - dist : w + dist; };
            // one granule per lane, polled until it carries this call's and this level's tag (or the wait is given up)
            auto poll = [&](const unsigned long long *src, bool wanted, bool &bad) -> uint32_t {
                bool ok = !wanted;
                uint32_t data = 0;
                const long long t_begin = wall_clock64();
                for (;;) {
                    if (!ok) {                                   // two loads in flight, half a round trip apart: the granule is seen sooner
                        const unsigned long long q1 = kc_load(src);
                        __builtin_amdgcn_s_sleep(12);
                        const unsigned long long q2 = kc_load(src);
                        if ((uint32_t)(q1 >> 32) == tag) { ok = true; data = (uint32_t)q1; }
                        else if ((uint32_t)(q2 >> 32) == tag) { ok = true; data = (uint32_t)q2; }
                    }
                    if (__all(ok)) break;
                    if (wall_clock64() - t_begin > ITD_KC_TIMEOUT) { bad = true; break; }
                }
                bad = __any(bad);
                return data;
            };
            int have = 0, dist0 = 1;
            bool edge = false, bad = false, first_round = true, ends_read = false;
            if (li == 1) KC_MARK(49);
            while (have < want && !edge && !bad) {
                // the workgroups whose records this round reads, nearest first (all lanes agree)
                // (lane k of vsel = the k-th workgroup selected, wave-uniform values, read with v_readlane — as four variables chosen
                //  by nsel the compiler kept them as an array in scratch memory)
                int vsel = -1, nsel = 0, ends_only = -1;
                bool reach_edge = false;
                auto select = [&](int v) { vsel = lane == nsel ? v : vsel; ++nsel; };
                auto sel = [&](int k) { return __builtin_amdgcn_readlane(vsel, k); };
                if (first_round) {                                           // the nearest, whatever they hold
                    for (int k = 0; k < kKcRound; ++k) {
                        const int v = nb(dist0 + k);
                        if (v < 0 || v >= W) { reach_edge = true; break; }
                        select(v);
                    }
                    dist0 += kKcRound;
                    first_round = false;
                } else {                                                     // knots are sparse here: the counts of the next 64, then those that hold knots
                    const int v = nb(dist0 + lane);
                    const bool inr = v >= 0 && v < W;
                    const uint32_t cnt = poll(rec_slot(lev, inr ? v : w), inr, bad);
                    if (bad) break;
                    if (__any(inr && cnt == kKcPoison)) { bad = true; break; }
                    unsigned long long nz = __ballot(inr && cnt != 0);
                    reach_edge = __ballot(!inr) != 0ull;
                    int pend = 0;
                    while (nz && nsel < 3 && have + pend < want) {
                        const int l = __builtin_ctzll(nz);
                        nz &= nz - 1;
                        select(nb(dist0 + l));
                        pend += min((int)__shfl((int)cnt, l), want - have - pend);
                    }
                    if (reach_edge && have + pend < want && !ends_read) {    // the walk will end at the signal's end: its end samples
                        const int last = nsel == 0 ? -1 : sel(nsel - 1);
                        // (not among the workgroups that hold knots: read for the end samples only — if an earlier round has passed
                        //  it, its knots are in the halo already and so are the end samples)
                        if (last != v_end) { ends_only = nsel; select(v_end); }
                    }
                    dist0 += 64;
                }
                // their granules: 16 lanes per workgroup
                {
                    const int v = __shfl(vsel, a);
                    int gi = -1;
                    if (g == 0) gi = 0;
                    else if (side == 0) { if (g <= 6) gi = g; else if (g <= 10 && v == 0) gi = 16 + (g - 7); }
                    else { if (g <= 9) gi = 6 + g; else if (g <= 13 && v == W - 1) gi = 16 + (g - 10); }
                    const bool wanted = a < nsel && gi >= 0;
                    const uint32_t data = poll(rec_slot(lev, wanted ? v : w) + (wanted ? gi : 0), wanted, bad);
                    if (bad) break;
                    s_stage[side][a][g] = data;
                }
                if (li == 1) KC_MARK(50);
                wave_sync();
                for (int k = 0; k < nsel && have < want; ++k) {              // (every lane follows the walk; lane 0 writes)
                    const int vv = sel(k);
                    const uint32_t *sg = s_stage[side][k];
                    const uint32_t cnt = sg[0];
                    if (cnt == kKcPoison) { bad = true; break; }
                    if (lane == 0) {
                        if (side == 0 && vv == 0) { s_ends[0] = bits_d(sg[7], sg[8]); s_ends[1] = bits_d(sg[9], sg[10]); }
                        if (side == 1 && vv == W - 1) { s_ends[2] = bits_d(sg[10], sg[11]); s_ends[3] = bits_d(sg[12], sg[13]); }
                    }
                    if (vv == v_end) ends_read = true;
                    const int take = k == ends_only ? 0 : min((int)cnt, want - have);
                    if (lane == 0) {
                        for (int h = 0; h < take; ++h) {
                            // in front: the neighbour's last knot first (granules 4..6), then its second last (1..3); behind: its first three
                            const int gq = side == 0 ? (h == 0 ? 4 : 1) : 1 + 3 * h;
                            const int d = side == 0 ? 1 - have - h : ck + 2 + have + h;
                            k_pos[d] = (int32_t)sg[gq];
                            k_X[d] = bits_d(sg[gq + 1], sg[gq + 2]);
                        }
                    }
                    have += take;
                }
                wave_sync();                                                 // (the staging rows are read before the next round rewrites them)
                if (reach_edge && have < want) edge = true;
            }
            if (li == 1) KC_MARK(51);
            if (li == 1) KC_MARK_T(64, 52);
            if (lane == 0) {
                if (bad) give_up(kKfFailWait);
                else if (have < want) {                                      // the signal's end: the virtual knot at sample 0 / n-1 (ITD.py:96,98)
                    const int d = side == 0 ? 1 - have : ck + 2 + have;
                    k_pos[d] = side == 0 ? 0 : n1;
                    k_X[d] = side == 0 ? s_ends[0] : s_ends[3];
                }
                s_i[2 + side] = have;
                if (!bad) {
                    // what the walk has assembled must be knots strictly outside the range, in order (a guard on the exchange
                    // itself: the sample pass verifies knots, not table values)
                    const int32_t lo_pos = (int32_t)(t0 * TW), hi_pos = (int32_t)min((int64_t)t1 * TW, n);
                    bool okh = true;
                    for (int h = 0; h < have; ++h) {
                        if (side == 0) { const int d = 1 - h; okh = okh && k_pos[d] >= 1 && k_pos[d] < lo_pos && (h == 0 || k_pos[d] < k_pos[d + 1]); }
                        else { const int d = ck + 2 + h; okh = okh && k_pos[d] >= hi_pos && k_pos[d] < n1 && (h == 0 || k_pos[d] > k_pos[d - 1]); }
                    }
                    if (!okh) give_up(kKfFailVerify);
                }
            }
        }
        if (lev == L0) published = lev;
        kc_barrier();
        if (s_i[1]) return false;
        if ((ws.dbg_kind == 6 || ws.dbg_kind == 7) && lev == ws.dbg_lev && w == ws.dbg_wg && sig == ws.dbg_sig) {   // fault injection (tests): a halo knot as received
            if (tid == 0) {
                const int d = ws.dbg_slot < 2 ? ws.dbg_slot : ck + ws.dbg_slot;     // slots 0, 1: in front; 2 .. 4: behind
                if (ws.dbg_kind == 6) k_X[d] = __builtin_bit_cast(double, dbits(k_X[d]) + (unsigned long long)(long long)ws.dbg_delta);
                else k_pos[d] += ws.dbg_delta;
            }
            kc_barrier();
        }
        KC_MARK(5 + 4 * min(li, 12));
        // B. knot values (ITD.py:100-110) and slopes (ITD.py:115-116): a thread computes B of its knot and of the next one (no
        //    exchange in between), the level's table entry leaves at once
        const int nl = s_i[2], nr = s_i[3];
        const int d_vs = nl < 2 ? 1 - nl : -8, d_ve = nr < 3 ? ck + 2 + nr : 1 << 30;     // the virtual end knots' dense indices
        const double e0 = s_ends[0], e1 = s_ends[1], e2 = s_ends[2], e3 = s_ends[3];
        const double m0 = (e0 + e1) / 2.0, mn = (e2 + e3) / 2.0;                          // ITD.py:101-102
        const int dmaxS = min(ck + 2, d_ve);
        auto knot_value = [&](int d) {
            if (d == d_vs) return m0;
            if (d == d_ve) return mn;
            const double frac = int_ratio(k_pos[d] - k_pos[d - 1], k_pos[d + 1] - k_pos[d - 1]);   // (the check wavefronts divide in full)
            const double tt = frac * (k_X[d + 1] - k_X[d - 1]);
            const double u = k_X[d - 1] + tt;
            return 0.5 * u + 0.5 * k_X[d];                                                // ITD.py:107-110
        };
        bool nonfin = false;
        for (int d = 1 + tid; d <= dmaxS; d += NT) {
            const double Bv = knot_value(d);
            const double Sv = d == d_ve ? 0.0 : (knot_value(d + 1) - Bv) / (k_X[d + 1] - k_X[d]);
            k_B[d] = Bv;
            k_S[d] = Sv;
            // the level's table: the knot in front of the range, its own, the one behind
            KfEntry en; en.X = k_X[d]; en.B = Bv; en.S = Sv; en.pos = k_pos[d]; en.pad = 0;
            slab[toff + d - 1] = en;
            nonfin = nonfin || !(Bv - Bv == 0.0) || !(Sv - Sv == 0.0);
        }
        if (tid < kKcTiles) {                                                // (wavefront 3 has taken this level's words)
#pragma unroll
            for (int q = 0; q < 8; ++q) t_w[tid * 8 + q] = 0ull;
        }
        kc_barrier();
        KC_MARK(6 + 4 * min(li, 12));
        // C. every candidate's three values through the maps of the segments they lie in (ITD.py:114-117); the next level's knots.
        //    Thread t takes candidates t, t + 256, ...: the lists shrink ~2.6x per level, one candidate per thread from the second level on
        int32_t pos[ENT];
        double yl[ENT], yc[ENT], yr[ENT];
        int nfl[ENT];
#pragma unroll
        for (int i = 0; i < ENT; ++i) {
            const int j = i * NT + tid;
            nfl[i] = 0; pos[i] = 0; yl[i] = yc[i] = yr[i] = 0.0;
            if (j < c) {
                const int fl = c_fl[j];
                pos[i] = c_pos[j];
                const int dc = c_rk[j] + 1;                     // the segment of the candidate's own sample
                const int dl = dc - (fl & 1);                   // of the sample in front of it: a knot's left neighbour lies in the previous segment
                const int dr = (k_pos[dc + 1] == pos[i] + 1) ? dc + 1 : dc;
                double xl_, xc_, xr_;                           // the candidate's triple at this level
                if constexpr (FIRST) { xl_ = va[i]; xc_ = vb[i]; xr_ = vd[i]; }
                else { xl_ = LN.c_xl()[j]; xc_ = LN.c_xc()[j]; xr_ = LN.c_xr()[j]; }
                yl[i] = k_B[dl] + k_S[dl] * (xl_ - k_X[dl]);
                yc[i] = k_B[dc] + k_S[dc] * (xc_ - k_X[dc]);
                yr[i] = (pos[i] + 1 == n1) ? 0.0 : k_B[dr] + k_S[dr] * (xr_ - k_X[dr]);   // baseline[n-1] stays 0, ITD.py:112-117
                const bool knot = kf_pred(yl[i], yc[i], yr[i]);
                const bool sticky = (fl & 2) || near_tie(yl[i], yc[i]) || near_tie(yc[i], yr[i]);   // a (near) tie: rounding may move the knot
                nfl[i] = (knot ? 1 : 0) | (sticky ? 2 : 0);
                nonfin = nonfin || !(yl[i] - yl[i] == 0.0) || !(yc[i] - yc[i] == 0.0) || !(yr[i] - yr[i] == 0.0);
            }
        }
        if (li == 1) KC_MARK(44);
        if (nonfin) atomicMax(&ks->nf, 64 - lev);
        // the next level's end samples: workgroup 0 / the last one (ITD.py:101-102 through this level's maps).  Thread 192: the
        // wavefront with the fewest candidates also publishes the record's fixed part below
        double ne0 = 0.0, ne1 = 0.0, ne2 = 0.0;
        if (tid == 192) {
            if (w == 0) {                                       // dense index 1 is the virtual knot at sample 0
                ne0 = k_B[1] + k_S[1] * (e0 - e0);
                ne1 = (2 != d_ve && k_pos[2] == 1) ? k_B[2] + k_S[2] * (e1 - k_X[2]) : k_B[1] + k_S[1] * (e1 - e0);
            }
            if (w == W - 1) ne2 = k_B[ck + 1] + k_S[ck + 1] * (e2 - k_X[ck + 1]);   // sample n-2 lies in the last knot's segment (knots are <= n-2)
        }
        if (li == 1) KC_MARK(45);
        // the survivors, in order: positions in the next level's list from ballots — per 256-candidate chunk and wavefront the
        // counts of survivors and of knots among them, one barrier, sixteen sums
        int pk[ENT], pn[ENT];
#pragma unroll
        for (int i = 0; i < ENT; ++i) {
            const unsigned long long kb = __ballot(nfl[i] != 0), nb = __ballot((nfl[i] & 1) != 0);
            pk[i] = mbcnt64(kb, 0);
            pn[i] = mbcnt64(nb, 0);
            if (lane == 0) s_cnt[i * kKcWaves + wave] = __popcll(kb) | (__popcll(nb) << 16);
        }
        kc_barrier();                                           // (also: every read of the old list and of the knot arrays is done)
        int tot = 0, base[ENT];
#pragma unroll
        for (int i = 0; i < ENT; ++i) {
#pragma unroll
            for (int q = 0; q < kKcWaves; ++q) {
                if (q == wave) base[i] = tot;
                tot += s_cnt[i * kKcWaves + q];
            }
        }
        if (li == 1) KC_MARK(46);
        const int ckn = tot >> 16;
        if ((tot & 0xffff) > kKcCap) {                          // (only behind the hand-over level: its list may be longer than the later layout holds)
            if (tid == 0) give_up(kKfFailCapacity);
            kc_barrier();
            return false;
        }
        const bool more = lev < M + 1;                          // the next level's record for the neighbours leaves first, from the registers
        unsigned long long *slot_n = rec_slot(more ? lev + 1 : lev, w);
        const uint32_t tag_n = rec_tag(lev + 1);
        auto publish_knot = [&](int first_granule, int32_t p, double v) {
            const unsigned long long b = dbits(v);
            kc_store(slot_n + first_granule, (uint32_t)p, tag_n);
            kc_store(slot_n + first_granule + 1, (uint32_t)b, tag_n);
            kc_store(slot_n + first_granule + 2, (uint32_t)(b >> 32), tag_n);
        };
        if (more && tid == 192) {
            kc_store(slot_n, (uint32_t)ckn, tag_n);
            const double ea = w == 0 ? ne0 : (w == W - 1 ? ne2 : 0.0), eb = w == 0 ? ne1 : 0.0;
            const unsigned long long ba = dbits(ea), bb = dbits(eb);
            kc_store(slot_n + 16, (uint32_t)ba, tag_n); kc_store(slot_n + 17, (uint32_t)(ba >> 32), tag_n);
            kc_store(slot_n + 18, (uint32_t)bb, tag_n); kc_store(slot_n + 19, (uint32_t)(bb >> 32), tag_n);
            if (ckn < 2) publish_knot(1, 0, 0.0);               // (slots of knots that do not exist carry the tag too: the readers poll them)
            if (ckn < 1) publish_knot(4, 0, 0.0);
            for (int h = ckn; h < 3; ++h) publish_knot(7 + 3 * h, 0, 0.0);
        }
        if (li == 1) KC_MARK(47);
#pragma unroll
        for (int i = 0; i < ENT; ++i) {
            if (nfl[i]) {
                const int o = (base[i] & 0xffff) + pk[i];
                const int r = (base[i] >> 16) + pn[i] + (nfl[i] & 1);     // knots of the range at or before the candidate
                if (more && (nfl[i] & 1)) {
                    if (r <= 3) publish_knot(4 + 3 * r, pos[i], yc[i]);          // the first three: granules 7.., 10.., 13..
                    if (r == ckn - 1) publish_knot(1, pos[i], yc[i]);            // the second last
                    if (r == ckn) publish_knot(4, pos[i], yc[i]);                // the last
                }
                // (the next level's lists: always the later layout)
                LN.c_pos()[o] = pos[i]; LN.c_xl()[o] = yl[i]; LN.c_xc()[o] = yc[i]; LN.c_xr()[o] = yr[i]; LN.c_fl()[o] = (unsigned char)nfl[i]; LN.c_rk()[o] = (unsigned short)r;
                if (nfl[i] & 1) {
                    LN.k_pos()[r + 1] = pos[i];
                    LN.k_X()[r + 1] = yc[i];
                    const int tt = pos[i] / TW - t0, q = pos[i] & (TW - 1);
                    atomicOr(&t_w[tt * 8 + 2 * (q >> 7) + (q & 1)], 1ull << ((q & 127) >> 1));
                }
            }
        }
        if (li == 1) KC_MARK(48);
        if (tid == 192) {
            atomicAdd(&ks->acc_mlev[lev + 1], ckn);
            if (w == 0) { s_ends[0] = ne0; s_ends[1] = ne1; }
            if (w == W - 1) { s_ends[2] = ne2; s_ends[3] = 0.0; }
        }
        toff += ck + 2;
        c = tot & 0xffff;
        ck = ckn;
        if (more) published = lev + 1;
        kc_barrier();
        KC_MARK(7 + 4 * min(li, 12));
        return true;
    };
    if (!s_i[1] && level(LH, L0))
        for (int lev = L0 + 1; lev <= M + 1; ++lev)
            if (!level(LN, lev)) break;
    // a workgroup that gave up tells everyone who may still wait for it
    if (s_i[1] && wave == 0) {
        for (int lev = published + 1; lev <= M + 1; ++lev) kc_publish(rec_slot(lev, w), rec_tag(lev), lane, 0, LN.k_pos(), LN.k_X(), 0.0, 0.0, true);
    }
    // the signal's last workgroup draws the stop rules (ITD.py:400-426) from the list sizes, publishes the verdict and leaves the
    // accumulators cleared for the next call (every workgroup's additions to them are complete — each wavefront's wait below —
    // before its arrival is counted)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    kc_barrier();
    if (tid == 192) {
        if (atomicAdd(&ks->done, 1) == W - 1) {
            int lend = M + 1, natural = 0;
            for (int lev = L0; lev <= M + 1; ++lev) {
                if (__hip_atomic_load(&ks->acc_mlev[lev + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2) { lend = lev; natural = 1; break; }   // "No more decompositions possible"
            }
            const int nf = __hip_atomic_load(&ks->nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int fail = __hip_atomic_load(&ks->acc_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nf && 64 - nf <= lend) fail |= kKfFailNonFinite;
            for (int j = 0; j < kMaxLevels + 2; ++j) {
                ks->mlev[j] = __hip_atomic_load(&ks->acc_mlev[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ks->acc_mlev[j] = 0;
            }
            ks->fail = fail;
            ks->m_exact = 0;
            ks->natural = natural;
            ks->lend = lend;
            ks->fail_lev = nf ? 64 - nf : 99;
            // capped fused levels: a signal that stops inside them is finished here (the level launches behind the sample pass return at
            // once: SigState::skip); one that does not goes on from the baseline the sample pass leaves
            ks->cont = (ws.cap && !natural) ? 1 : 0;
            if (ws.cap && natural) st->skip = 1;
            ks->acc_fail = 0;
            ks->nf = 0;
            ks->done = 0;
            ks->seq = seq + 1;
        }
    }
    launch_done();
#if ITD_PROF
    KC_MARK(60);
    if (tid == 0 && g_kc_prof) {
        kc_marks[61] = (unsigned long long)(sig * W + w);
        kc_marks[62] = (unsigned long long)blockIdx.x;
        unsigned long long *o = g_kc_prof + (size_t)blockIdx.x * 64;
        for (int i = 0; i < 64; ++i) if (i < 52 || i > 57) o[i] = kc_marks[i];
    }
#endif
}

// ---- the sample pass: every fused level of a tile in registers.  grid = (tiles, batch), 64 threads. -----------------------
// What it takes from the knot side it VERIFIES, every level, so that a delivered row is the reference's by construction (induction
// over the levels: the level's input in the registers is exact — the first fused level's comes from the level-by-level launches):
//   V0  the level's knots (flag words) = the exact predicate on the level's input: the first fused level's here, every later
//       level's on the values the level before has just computed;
//   V1  table entries 1 .. kn of the tile's run ARE the tile's knots: position and value equal the owner lane's own sample, bit
//       for bit; tile 0 owns the virtual knot at sample 0, the last tile the one at sample n-1 (ITD.py:96,98,101-102);
//   V2  B of every knot's entry recomputed from the positions and values of entries r-1, r, r+1 in the association of
//       ITD.py:107-110, S from B and X of entries r, r+1 (ITD.py:115-116), compared bit for bit — by wavefronts of their own at the
//       end of the grid, one per level and slab, a knot per lane (in the tiles' wavefronts the two divisions ran on a handful of
//       lanes: 180 us against 144 without any check);
//   V3  the runs chain (the check wavefronts): inside a knot-side workgroup's slab the next tile's run starts where this one's knots
//       end (first[t+1] == first[t] + kn), so entry 0 / kn+1 of a run is physically the entry its owner tile verifies — the nearest
//       knot in front of / behind the tile; across two slabs the next slab's copy of the knot in front of it and this slab's copy
//       of the knot behind it are compared with the owners' entries bit for bit, all four fields (the check wavefronts).
//   counts  the levels' knot counts (what the knot side's stop rules read: KfSig::mlev) are counted again from the verified flag words
//       — one more check wavefront sums what the slabs' chained runs hold, a slab per lane — and have to agree, level by level.
// Every field of every entry a tile uses is thereby checked by its owner or equals, through a chain of bitwise equalities, an
// entry that is; any mismatch refuses the signal (kKfFailVerify).  tests/test_gpu_fused.py injects faults into every one of
// these (itd_debug_kf_fault) and requires the refusal.
// != 0 <=> the bit patterns differ — as integer arithmetic on the vector unit (compares would go through lane masks in scalar
// registers, which the sample pass has none to spare of)
__device__ __forceinline__ int ne_d(double a, double b)
{
    const unsigned long long x = dbits(a) ^ dbits(b);
    return (int)((unsigned)x | (unsigned)(x >> 32));
}
__device__ __forceinline__ int ne_i(int32_t a, int32_t b) { return a ^ b; }
#ifndef ITD_KF_VERIFY
#define ITD_KF_VERIFY 31          // timing-only A/B builds: bit k = check Vk is compiled in, bit 4 = the levels' knot counts (the shipped build: all)
#endif
// ---- V2, V3 and the counts: what can be verified on the tables themselves.  One wavefront per knot-side workgroup (slab) and level.
//      Entries 1 .. cnt of the slab's table are the range's knots, 0 and cnt + 1 the copies of the knots around the range.
//      V3 inside   first[t + 1] == first[t] + (population of tile t's flag words) for the range's tiles: the runs chain;
//      V2          B of entries 1 .. cnt from the positions and values of entries r-1, r, r+1 (ITD.py:107-110), S of entries
//                  0 .. cnt from B and X of entries r, r+1 (ITD.py:115-116), bit for bit;
//      V3 across   the next slab's entry 0 (its copy of the knot in front of it) equals this slab's entry cnt, its entry 1
//      two slabs   (the first knot at or behind its first sample) this slab's entry cnt + 1: all four fields, bit for bit.
//      Returns != 0 in some lane if anything differs.
__device__ __forceinline__ int kf_check_slab(const KfWs &ws, int sig, int w, int li, int lane)
{
    const int n_tiles = ws.n_tiles;
    const int t0 = w * ws.tpw, t1 = min(n_tiles, t0 + ws.tpw);
    const KfEntry *pool = ws.pool + (size_t)sig * ws.wgs_max * kKcSlab;
    const int pool_n = ws.wgs_max * kKcSlab;
    int vb = 0;
    const int32_t *firstl = ws.first + ((size_t)sig * ws.nlev + li) * n_tiles;
    const unsigned long long *tfl = ws.tflags + ((size_t)sig * ws.nlev + li) * n_tiles * 8;
    const int start = firstl[t0];
    const int f_next = t1 < n_tiles ? firstl[t1] : -1;
    int cnt = 0;
    for (int tb = t0; tb < t1; tb += kWave) {                // a tile per lane: the chain of the runs; the range's knot count
        const int tt = tb + lane;
        int f = 0, fn = 0, kt = 0;
        if (tt < t1) {
            using U2 = unsigned long long __attribute__((ext_vector_type(2)));
            const U2 *wp = reinterpret_cast<const U2 *>(tfl + (size_t)tt * 8);
            f = firstl[tt];
            fn = tt + 1 < t1 ? firstl[tt + 1] : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const U2 v = wp[q]; kt += __popcll(v.x) + __popcll(v.y); }
            if ((ITD_KF_VERIFY & 8) && tt + 1 < t1) vb |= ne_i(fn, f + kt);
        }
        const int l_last = t1 - 1 - tb;                      // the lane of the range's last tile, if it is in this round
        if (l_last < kWave) cnt = __shfl(f + kt, l_last) - start;
    }
    if (start < 0 || cnt < 0 || start + cnt + 2 > pool_n || (t1 < n_tiles && (f_next < 0 || f_next + 2 > pool_n))) vb = 1;
    else if (ITD_KF_VERIFY & 4) {
        const KfEntry *e = pool + start;
        auto check = [&](const KfEntry &em, const KfEntry &e0, const KfEntry &e1, bool with_b) {
            const double Sv = (e1.B - e0.B) / (e1.X - e0.X);                         // ITD.py:115-116
            int bad_ = ne_d(Sv, e0.S);
            const double frac = (double)(e0.pos - em.pos) / (double)(e1.pos - em.pos);
            const double tt = frac * (e1.X - em.X);
            const double u = em.X + tt;
            const double Bv = 0.5 * u + 0.5 * e0.X;                                  // ITD.py:107-110
            if (with_b) bad_ |= ne_d(Bv, e0.B);
            return bad_;
        };
        for (int r = lane; r <= cnt; r += ITD_KF_CHECK_STEP * kWave) {       // two entries per lane and round: their loads fly together
            const int r2 = r + kWave, q2 = min(r2, cnt);     // (beyond the slab's entries: the last one again, not counted)
            const KfEntry a0 = e[r], a1 = e[r + 1], am = e[max(r - 1, 0)];
            const KfEntry b0 = e[q2], b1 = e[q2 + 1], bm = e[q2 - 1 < 0 ? 0 : q2 - 1];
            vb |= check(am, a0, a1, r >= 1);
            if (ITD_KF_CHECK_STEP == 2 && r2 <= cnt) vb |= check(bm, b0, b1, true);
        }
        if (t1 < n_tiles && lane < 2) {
            const KfEntry a = pool[f_next + lane], o = e[cnt + lane];
            vb |= ne_i(a.pos, o.pos) | ne_d(a.X, o.X) | ne_d(a.B, o.B) | ne_d(a.S, o.S);
        }
    }
    return vb;
}
// the levels' knot counts: the stop rules and the rows' number are drawn from the knot side's mlev[] — it has to be the population of
// the flag words the sample pass verifies.  A slab per lane: what its chained runs hold (the slab's own check verifies the chain:
// first[last tile] + that tile's population - first[first tile]), summed over the slabs.  (Added up by the slabs' wavefronts with one
// atomic each the same sum cost the launch 27 us: 3 584 read-modify-writes behind a memory system full of the tiles' stores —
// profiles/r05/experiments.)  Wave-uniform result: != 0 if level index li's count differs.
__device__ __forceinline__ int kf_check_counts(const KfWs &ws, int sig, const KfSig *ks, int li, int lane)
{
    const int n_tiles = ws.n_tiles;
    const int32_t *firstl = ws.first + ((size_t)sig * ws.nlev + li) * n_tiles;
    const unsigned long long *tfl = ws.tflags + ((size_t)sig * ws.nlev + li) * n_tiles * 8;
    int tot = 0;
    for (int wb = 0; wb < ws.wgs; wb += kWave) {
        const int a = (wb + lane) * ws.tpw, b = min(n_tiles, a + ws.tpw);
        if (wb + lane < ws.wgs && a < b) {
            using U2 = unsigned long long __attribute__((ext_vector_type(2)));
            const U2 *wp = reinterpret_cast<const U2 *>(tfl + (size_t)(b - 1) * 8);
            int kt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const U2 v = wp[q]; kt += __popcll(v.x) + __popcll(v.y); }
            tot += firstl[b - 1] + kt - firstl[a];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    return ne_i(tot, ks->mlev[ws.L0 + li]);
}
// workgroups of the sample pass's grid that verify the tables (V2, V3), in front of the tiles': one per knot-side workgroup,
// padded to a multiple of 8 so that a tile's workgroup lands on the same XCD as in every other launch (xcd_item)
__host__ __device__ constexpr int kf_check_blocks(int wgs) { return (wgs + 1 + 7) & ~7; }     // (+ 1: the wavefront that sums the slabs' counts)
#ifndef ITD_KF_APPLY_WAVES
#define ITD_KF_APPLY_WAVES 8      // wavefronts per SIMD the sample pass is compiled for (0 = the compiler's choice; 8: 64 VGPRs, no spill; fused from
                                  // level 2 at 2^24: 181.9 us at 6, 177.1 at 7, 176.3 at 8 — profiles/r05/ab_sample_pass_occupancy.txt)
#endif
#ifndef ITD_KF_FASTGROUP
#define ITD_KF_FASTGROUP 1        // A/B builds: 0 = every 128-sample group takes the by-rank path
#endif
template <int TW, int CAP, bool BASES, bool PART>      // BASES: the caller wants the baselines too (get_baselines()): a second row store per level
                                                       // PART: capped fused levels (KfWs::cap): the last fused level's baseline is stored for the level launches behind
__global__ __launch_bounds__(kWave)
#if ITD_KF_APPLY_WAVES
__attribute__((amdgpu_waves_per_eu(ITD_KF_APPLY_WAVES, ITD_KF_APPLY_WAVES)))
#endif
void k_kf_apply(KfWs ws, const double *__restrict__ xl, int64_t xl_stride, int64_t n,
                                                    const TileRec *__restrict__ recs_l0, double *__restrict__ rows, int64_t rows_stride,
                                                    double *__restrict__ bases, int64_t bases_stride)
{
    constexpr int G2 = TW / 128;
    __shared__ double s_X[CAP + 2], s_B[CAP + 2], s_S[CAP + 2];
    __shared__ int32_t s_P[CAP + 2];
    // The grid is (n_tiles + n_chk, signals); workgroups are dispatched in the order of their linear id (x fastest).  The FIRST n_chk x signals
    // of them are the check wavefronts of ALL the launch's signals (their dependent loads, ~40 round trips, take 40-70 us: in front of
    // every signal's own tiles the last signals' check wavefronts outlived the launch's tiles — a batch chunk's sample pass took 103 us
    // against 88 for the same samples as one signal), the rest the tiles, signal after signal.
    const int lane = lane_id();
    const int n_tiles = ws.n_tiles;
    const int n_chk = kf_check_blocks(ws.wgs);
    const int lin = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, n_chk_all = n_chk * (int)gridDim.y;
    const bool is_chk = lin < n_chk_all;
    const int sig = is_chk ? lin / n_chk : (lin - n_chk_all) / n_tiles;
    const int item = is_chk ? lin - sig * n_chk : (lin - n_chk_all) - sig * n_tiles;
    KfSig *ks = ws.sig + sig;
    if (!ks->active || ks->fail) return;
    const int lend = ks->lend, natural = ks->natural;
    if (lend < 0) return;
    const bool cont = PART && ks->cont != 0;      // the level launches go on behind level lend: its row is a rotation like any other
    if (is_chk) {
        // ---- one wavefront per knot-side workgroup, level after level: what can be verified on the tables themselves (kf_check_slab,
        //      kf_check_counts above), a knot (or a tile) per lane instead of a handful of lanes in every tile's wavefront
        if (!(ITD_KF_VERIFY & 12)) return;
        const int w = item;
        int vb = 0;
        int fl = 99;                                             // the first level at which something differs
        if (w == ws.wgs && (ITD_KF_VERIFY & 16)) {
            for (int li = 0; li <= lend - ws.L0; ++li) { vb |= kf_check_counts(ws, sig, ks, li, lane); if (vb != 0 && fl == 99) fl = ws.L0 + li; }
            if (vb != 0 && lane == 0) { atomicOr(&ks->fail, kKfFailVerify); atomicMin(&ks->fail_lev, fl); }
            return;
        }
        if (w >= ws.wgs) return;                                 // (the blocks that only pad the count to a multiple of 8)
        // (one workgroup per slab, all its levels: every workgroup of this grid costs dispatch time — 3072 of them, one per level and
        //  slab, cost 6 us)
        for (int li = 0; li <= lend - ws.L0; ++li) { vb |= kf_check_slab(ws, sig, w, li, lane); if (fl == 99 && __any(vb != 0)) fl = ws.L0 + li; }
        if (fl != 99 && lane == 0) { atomicOr(&ks->fail, kKfFailVerify); atomicMin(&ks->fail_lev, fl); }
        return;
    }
    const int t = xcd_item(item, n_tiles);
    const int ni = (int)n, si = t * TW, rem = ni - si;
    const double *xs = xl + (int64_t)sig * xl_stride;
    const __amdgpu_buffer_rsrc_t rx = tile_rsrc32(xs + si, rem, 8);
    double xr[G2][2];
#pragma unroll
    for (int g = 0; g < G2; ++g) tile_load2<true>(rx, lane * 16, g * 1024, xr[g][0], xr[g][1]);
    // the samples next to the tile: the neighbours' records of the first fused level's input (their first / last sample)
    const TileRec *rc = recs_l0 + (size_t)sig * n_tiles;
    double x_lo = t > 0 ? rc[t - 1].edge[1] : 0.0;
    double x_hi = rem > TW ? rc[t + 1].edge[0] : 0.0;
    const int nrem = min(rem, TW + 2);
    const int32_t *first = ws.first + ((size_t)sig * ws.nlev) * n_tiles;
    const unsigned long long *tf = ws.tflags + ((size_t)sig * ws.nlev) * n_tiles * 8;
    const KfEntry *pool = ws.pool + (size_t)sig * ws.wgs_max * kKcSlab;
    double *rows_s = rows + (int64_t)sig * rows_stride + si;
    double *bases_s = (BASES && bases) ? bases + (int64_t)sig * bases_stride + si : nullptr;
    unsigned long long wcur = lane < 2 * G2 ? tf[(size_t)t * 8 + lane] : 0ull;       // this level's flag words, lane j holds word j
    bool bad = false;            // wave-uniform findings
    int vbad = 0;                // per-lane findings
    int fl = 99;                 // the first level at which either showed (KfSig::fail_lev)
    // the knots of the values in the registers (x_lo / x_hi: the samples next to the tile): the exact predicate, ITD.py:59 on x and -x
    auto knots_of = [&](double lo_v, double hi_v, unsigned long long (&E2)[G2], unsigned long long (&O2)[G2]) {
        double d0[G2], d1[G2];
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const double fill = (g == 0) ? lo_v : wave_dpp<0x13C>(0.0, xr[g > 0 ? g - 1 : 0][1]);
            const double left = wave_dpp<0x138>(fill, xr[g][1]);
            d0[g] = xr[g][0] - left;
            d1[g] = xr[g][1] - xr[g][0];
        }
        knot_predicate<double, G2>(d0, d1, hi_v - xr[G2 - 1][1], E2, O2);
        if (si == 0 || nrem <= TW + 1) {       // first and last sample are never knots (ITD.py:70-73), nothing beyond n-2
            const int lo = si > 0 ? 0 : 1, hi = nrem - 2;
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                E2[g] &= bit_range((lo - 128 * g + 1) >> 1, (hi - 128 * g) >> 1);
                O2[g] &= bit_range((lo - 128 * g) >> 1, (hi - 128 * g - 1) >> 1);
            }
        }
    };
    // Loop over the fused levels.  At the top of a level the level's input is in the registers (xr, x_lo, x_hi) and its flag words
    // in wcur: V0 first — the exact predicate on those values must reproduce the words — then the level's map.  The pass after the
    // last level only counts the pending baseline's knots (the stop test, ITD.py:400-404).
    for (int lev = ws.L0; ; ++lev) {
        const int li = lev - ws.L0;
        const bool last = lev == lend;
        {
            unsigned long long E2[G2], O2[G2];
            knots_of(x_lo, x_hi, E2, O2);
            if (lev > lend) {
                int tot = 0;
#pragma unroll
                for (int g = 0; g < G2; ++g) tot += __popcll(E2[g]) + __popcll(O2[g]);
                if (lane == 0 && tot) atomicAdd(&ks->m_exact, tot);
                break;
            }
            if ((ITD_KF_VERIFY & 1) || lev > ws.L0) {
                WaveMasks wc;
                wc.lo = (unsigned)wcur; wc.hi = (unsigned)(wcur >> 32);
#pragma unroll
                for (int g = 0; g < G2; ++g) bad = bad || (wc.get(2 * g) != E2[g]) || (wc.get(2 * g + 1) != O2[g]);
            }
        }
        const int kfst = first[(size_t)li * n_tiles + t];        // the table entry of the knot in front of the tile
        WaveMasks wm;
        wm.lo = (unsigned)wcur; wm.hi = (unsigned)(wcur >> 32);
        int kn = 0;                                                // the tile's knots at this level
#pragma unroll
        for (int q = 0; q < 2 * G2; ++q) kn += __popcll(wm.get(q));
        if (kn > CAP) { if (lane == 0) atomicOr(&ks->fail, kKfFailCapacity); return; }
        if (kfst < 0 || kfst + kn + 2 > ws.wgs_max * kKcSlab) { if (lane == 0) { atomicOr(&ks->fail, kKfFailVerify); atomicMin(&ks->fail_lev, lev); } return; }   // (never read beyond the pool)
        // the next level's flag words (what the values computed below must reproduce)
        const unsigned long long wnext = (!last && lane < 2 * G2) ? tf[((size_t)(li + 1) * n_tiles + t) * 8 + lane] : 0ull;
        const KfEntry *tab = pool + kfst;
        for (int r = lane; r < kn + 2; r += kWave) {
            const KfEntry en = tab[r];
            s_X[r] = en.X; s_B[r] = en.B; s_S[r] = en.S; s_P[r] = en.pos;
        }
        wave_sync();
        double *row = rows_s + (int64_t)lev * (n + ITD_ROW_PAD);
        const __amdgpu_buffer_rsrc_t r_row = tile_rsrc32(row, rem, 8);
        const __amdgpu_buffer_rsrc_t r_bas = tile_rsrc32(bases_s ? bases_s + (int64_t)lev * n : row, (bases_s && (!last || cont)) ? rem : 0, 8);
        // (PART, no caller's baselines: the last fused level's baseline goes where the level launch behind expects its input)
        const __amdgpu_buffer_rsrc_t r_nx = tile_rsrc32((PART && !BASES) ? ws.xnext + (int64_t)sig * ws.xnext_stride + si : row, (PART && !BASES && last && cont) ? rem : 0, 8);
        int gbase = 0;
        bool nonfin = false;
        if (ITD_KF_VERIFY & 2) {           // V1 for the two tiles that own a virtual knot (rare paths, kept out of the groups' loop)
            if (t == 0 && lane == 0)      // the virtual knot at sample 0: x[0], mean(x[:2]) (ITD.py:96,101)
                vbad |= ne_i(s_P[0], 0) | ne_d(s_X[0], xr[0][0]) | ne_d(s_B[0], (xr[0][0] + xr[0][1]) / 2.0);
            if (rem <= TW) {              // the last tile: the virtual knot at sample n-1: x[n-1], mean(x[-2:]), no segment (ITD.py:98,102)
                auto pick = [&](int q) {  // the level's input at tile position q (wave-uniform): every group's candidates through
                    const int gq = q >> 7, lq = (q & 127) >> 1, par = q & 1;   // v_readlane, the choice on the scalar unit (registers
                    unsigned lo = 0, hi = 0;                                     // indexed by a variable would move the tile to scratch)
#pragma unroll
                    for (int g = 0; g < G2; ++g) {
                        const unsigned long long ve = dbits(xr[g][0]), vo = dbits(xr[g][1]);
                        const unsigned le = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ve, lq), he = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ve >> 32), lq);
                        const unsigned lo_ = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)vo, lq), ho = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(vo >> 32), lq);
                        if (gq == g) { lo = par ? lo_ : le; hi = par ? ho : he; }
                    }
                    return bits_d(lo, hi);
                };
                const double x_n1 = pick(nrem - 1), x_n2 = nrem >= 2 ? pick(nrem - 2) : x_lo;   // (rem == 1: sample n-2 is the previous tile's last)
                if (lane == 0)
                    vbad |= ne_i(s_P[kn + 1], ni - 1) | ne_d(s_X[kn + 1], x_n1) | ne_d(s_B[kn + 1], (x_n2 + x_n1) / 2.0) | ne_d(s_S[kn + 1], 0.0);
            }
            asm volatile("" : "+v"(vbad));
        }
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            __builtin_amdgcn_sched_barrier(0);   // one group's by-rank reads in flight at a time: registers, not latency, are scarce
            const unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
            int p = 128 * g + 2 * lane;
            asm volatile("" : "+v"(p));          // opaque: nothing derived from the position is kept across the levels' loop
            const double xe = xr[g][0], xo = xr[g][1];
            double be, bo;
            if (ITD_KF_FASTGROUP && (E | O) == 0ull) {
                // no knot among the group's 128 samples (most groups from the third fused level on): one segment, one table entry
                // for every lane, nothing to verify
                const double Xs = s_X[gbase], Bs = s_B[gbase], Ss = s_S[gbase];
                be = Bs + Ss * (xe - Xs);                            // ITD.py:114-117
                bo = Bs + Ss * (xo - Xs);
            } else {
                const int bE = lane_bit(E), bO = lane_bit(O);
                const int Le = mbcnt64(O, mbcnt64(E, gbase)) + bE;  // knots of the tile at or before the even sample = its slot
                const int Lo = Le + bO;
                const double Xe = s_X[Le], Xo = s_X[Lo];
                if (ITD_KF_VERIFY & 2) {                             // V1: a knot's entry is the owner lane's own sample
                    vbad |= -bE & (ne_i(s_P[Le], si + p) | ne_d(Xe, xe));
                    vbad |= -bO & (ne_i(s_P[Lo], si + p + 1) | ne_d(Xo, xo));
                    asm volatile("" : "+v"(vbad));                   // (decided here: the operands do not stay alive until the tile's verdict)
                }
                be = s_B[Le] + s_S[Le] * (xe - Xe);                  // ITD.py:114-117
                bo = s_B[Lo] + s_S[Lo] * (xo - Xo);
            }
            if (rem <= TW) {                                         // baseline[n-1] is never written (stays 0), ITD.py:112-117
                if (p >= nrem - 1) be = 0.0;
                if (p + 1 >= nrem - 1) bo = 0.0;
            }
            double re, ro;
            if (last && natural) { re = xe; ro = xo; }               // row c = baselines[c-1], ITD.py:404-416
            else if (last && !cont) { re = (xe - be) + be; ro = (xo - bo) + bo; }   // "Out of time!": rotation + baseline, ITD.py:420
            else { re = xe - be; ro = xo - bo; }                     // ITD.py:119
            tile_store2<true>(r_row, p * 8, re, ro);
            if constexpr (BASES) tile_store2<true>(r_bas, p * 8, be, bo);      // (without: not even a bounds-checked-away store is issued)
            if constexpr (PART && !BASES) tile_store2<false>(r_nx, p * 8, be, bo);   // (an empty descriptor except at the last level of a capped call)
            nonfin = nonfin || nonfinite(be) || nonfinite(bo);
            xr[g][0] = be;
            xr[g][1] = bo;
            gbase += __popcll(E) + __popcll(O);
        }
        if (__any(nonfin)) { if (lane == 0) atomicOr(&ks->fail, kKfFailNonFinite); return; }
        // the samples next to the tile, through the same maps
        double n_lo = 0.0, n_hi = 0.0;
        if (t > 0) n_lo = s_B[0] + s_S[0] * (x_lo - s_X[0]);
        if (rem > TW) {
            const int Lh = (s_P[kn + 1] == si + TW) ? kn + 1 : kn;
            n_hi = (si + TW == ni - 1) ? 0.0 : s_B[Lh] + s_S[Lh] * (x_hi - s_X[Lh]);
        }
        wave_sync();
        x_lo = n_lo;
        x_hi = n_hi;
        wcur = wnext;
        if (fl == 99 && (bad || __any(vbad != 0))) fl = lev;
    }
    if ((bad || __any(vbad != 0)) && lane == 0) { atomicOr(&ks->fail, kKfFailVerify); atomicMin(&ks->fail_lev, fl == 99 ? lend : fl); }
}

// The gate of the batch pipeline: ONE wavefront on the stream of the memory-bound launches.  The knot side of the next chunk has just become
// ready on the engine's second stream (an event behind the launch that wrote its input); this launch returns when every workgroup of
// it has started (KfWs::started has reached `target`), so that the launch behind the gate — tens of thousands of one-wavefront
// workgroups — cannot take the LDS and the wave slots the knot side's 65 KB workgroups need: side by side from then on, the
// latency-bound launch hides behind the memory-bound ones (tools/anyorder_probe.hip: no gate 229 us, gate 192, the parts 62 + 2 x 87).
// Gives up after `timeout` ticks of the 100 MHz clock — and says so in started[1]: the gate is also what orders a sample pass behind ITS
// knot side (which ended before the awaited one could start), so a gate that gave up voids the call: the engine reads the word with
// the call's summary (or k_verdict does, on the device) and repeats the call level by level.
__global__ __launch_bounds__(kWave) void k_kf_gate(unsigned long long *started, unsigned long long target, long long timeout)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > timeout) { if (threadIdx.x == 0) __hip_atomic_store(started + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}

// fault injection for the tests (itd_debug_kf_fault): ONE field of the workspace the sample pass is about to read is perturbed,
// between the knot side's launch and the sample pass.  kind 0 / 1 / 2: X / B / S of entry `slot` (mod the run's length) of tile
// `tile`'s run at fused level index `li` (delta added to the bit pattern: ulps); 3: its position; 4: first[li][tile]; 5: bit
// (delta & 63) of flag word `slot` & 7 of the tile flipped; 8: the level's knot count in the signal's head.
__global__ void k_kf_fault(KfWs ws, int sig, int kind, int li, int tile, int slot, int delta)
{
    if (threadIdx.x || blockIdx.x) return;
    const int n_tiles = ws.n_tiles;
    int32_t *first = ws.first + ((size_t)sig * ws.nlev + li) * n_tiles;                 // (signal `sig` of the launch)
    unsigned long long *tf = ws.tflags + (((size_t)sig * ws.nlev + li) * n_tiles + tile) * 8;
    if (kind == 8) { ws.sig[sig].mlev[ws.L0 + li] += delta; return; }    // the knot side's count of the level's knots (what the stop rules read)
    if (kind == 4) { first[tile] += delta; return; }
    if (kind == 5) { tf[slot & 7] ^= 1ull << (delta & 63); return; }
    int kn = 0;
    for (int q = 0; q < 8; ++q) kn += __popcll(tf[q]);
    KfEntry *en = ws.pool + (size_t)sig * ws.wgs_max * kKcSlab + first[tile] + slot % (kn + 2);
    if (kind == 0) en->X = __builtin_bit_cast(double, dbits(en->X) + (unsigned long long)(long long)delta);
    if (kind == 1) en->B = __builtin_bit_cast(double, dbits(en->B) + (unsigned long long)(long long)delta);
    if (kind == 2) en->S = __builtin_bit_cast(double, dbits(en->S) + (unsigned long long)(long long)delta);
    if (kind == 3) en->pos += delta;
}

// The verdict of the fused levels — what k_kf_finish did as a launch of its own — is drawn on the host by itd_get_summary from
// the head of KfSig (itd_engine.hip: kf_verdict).

}  // namespace itd
