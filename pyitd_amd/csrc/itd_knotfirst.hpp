// itd_knotfirst.hpp — the sparse levels of a decomposition, fused ("knot first").
//
// One extraction (ITD.py:79-121) maps every sample through an affine function of ITSELF inside its segment,
//     baseline[i] = B_k + S_k * (x[i] - x[e_k])        for i in [e_k, e_k+1)                              (ITD.py:114-117)
// so between two consecutive knots the baseline is a monotone image of the (monotone) input: the NEXT level's knots can only sit
// at this level's knots.  Hence the level recursion of the driver (ITD.py:384-432) can run on the KNOT LIST alone — a few per
// cent of the samples from level 2 or 3 on — and the samples need ONE pass for all of those levels instead of one pass (8 B
// read + 16 B written per sample) per level:
//
//   knot side   a candidate carries the level's values at (e-1, e, e+1).  A step computes B_k and S_k of every knot from the
//               knots' positions and values (ITD.py:100-116), writes them to the level's table, moves the three values to the
//               next level through the maps of the segments they lie in, and keeps the candidates on which the knot predicate
//               (ITD.py:59 on x and on -x) holds.  Sample n-2 (its right neighbour is forced to 0: baseline[n-1] is never
//               written, ITD.py:112-117), the samples next to exact ties of the caller's signal, and every candidate whose
//               triple ever shows an exact tie stay candidates for ever ("sticky"): those are the places where rounding makes
//               or breaks a plateau and a knot can appear where the previous level had none.  The stop rules (ITD.py:400-426)
//               are decided here, from the list sizes.
//   sample side k_kf_apply: a wavefront takes a 512-sample tile of the first fused level's input, and for every fused level
//               looks its tile's knots up in the level's table (positions as flag words, B / S / X by rank), maps its samples,
//               stores the rotation row, and RE-DERIVES the next level's knots from the values it has just computed — the exact
//               predicate on the actual samples.  Any difference from the knot side's flags means the shortcut missed a knot
//               (a plateau born from rounding somewhere else): the signal's result is discarded (SigState::kf_fail) and the
//               engine repeats the call level by level.  So whatever this path delivers is, by induction over the levels, bit
//               for bit what the level-by-level engine delivers; what it cannot deliver it reports.
//
// Traffic of the fused levels L0 .. L: 8 B read + 8 B per row written per sample, against 24 B per sample and level.
// Smooth or quantised input (plateaus everywhere) does not take this path for long: too many ties (> kKfSticky) or a failed
// verification send it back to the level-by-level engine.  A numpy statement of exactly this algorithm lives with the tests.
#pragma once
#include "itd_kernels.hpp"

#pragma clang fp contract(off)

namespace itd {

constexpr int kKfSticky = 64;      // sticky candidates per signal (more: the signal is left to the level-by-level engine)
constexpr int kKfCap = 128;        // knots of one tile and level the sample pass holds by rank
constexpr int kKfChunk = 256;      // list entries per workgroup of a knot-side step
constexpr int kKfPre = 4096;       // chunk-count prefixes the sticky block of a step keeps in LDS (lists of up to 2^20 entries)

struct KfEntry { double X, B, S; int32_t pos; int32_t pad; };   // a level's table: entry 0 = sample 0, entries 1..m the knots, m+1 = sample n-1
static_assert(sizeof(KfEntry) == 32, "table entries are two 16-byte words");

// failure bits (KfSig::fail, SigState::kf_fail)
constexpr int kKfFailVerify = 1, kKfFailCapacity = 2, kKfFailNonFinite = 4, kKfFailTies = 8;

struct KfSig {
    // ---- the head: what itd_get_summary copies to the host (kKfSigHead bytes) ----
    int32_t fail;
    int32_t active;       // the fused levels run for this signal (it had not stopped before the hand-over, no NaN anywhere)
    int32_t lend;         // -1 while the steps run; then the last fused level: rows 0 .. lend are the result
    int32_t natural;      // at lend: 1 = natural stop (row lend = the level's input), 0 = "Out of time!" (rotation + baseline)
    int32_t m_exact;      // knots of the last pending baseline, counted by the sample pass
    int32_t n_sticky;     // entries of spos (the hand-over's appends; then = n_sticky_lev)
    int32_t n_sticky_lev; // of which the first n_sticky_lev are sorted, unique and carry triples of the level in flight
    int32_t n_app;        // appends of the step in flight: slots kKfSticky + i
    int32_t mlev[kMaxLevels + 2];   // knots of level j's input
    // ---- device only ----
    int32_t toff[kMaxLevels + 2];   // level j's table starts at this entry of the signal's pool
    double ends[2][4];              // [level & 1]: x[0], x[1], x[n-2], x[n-1] of the level's input
    int32_t spos[kKfSticky + 64];   // sticky positions
    int32_t ssk[kKfSticky + 64];    // knots of the level's list at or before the position; for a fresh append: its list index
    int32_t sins[kKfSticky + 64];   // scratch of a step: 1 = a knot of the next level that is not in the list
    double stri[2][kKfSticky + 64][3];
};
constexpr size_t kKfSigHead = (8 + kMaxLevels + 2) * sizeof(int32_t);

struct KfWs {
    KfSig *sig;                   // [batch]
    int32_t *candP;               // [batch][2][cap]      1-based: entry k at [k]
    double *candT;                // [batch][2][cap][3]
    double *newT;                 // [batch][cap][3]      the next level's triples of the list entries
    int32_t *flag;                // [batch][cap]         1 = entry k is a knot of the next level
    int32_t *pref;                // [batch][cap]         survivors among the entries of k's chunk in front of k
    int32_t *cnt;                 // [batch][nchunk]      survivors per chunk
    KfEntry *pool;                // [batch][pool_cap]
    int32_t *first;               // [batch][nlev][tiles + 1]   knots of the level in front of every tile
    unsigned long long *tflags;   // [batch][nlev][tiles][8]    the level's knots as flag words (flag_pos format)
    int32_t *tie;                 // [batch][tiles]             the fused level-0 launch saw a zero difference in the tile (self-cleaning)
    int32_t *gpre;                // [batch][groups]            knots of level L0 in front of every 64-tile group
    int32_t cap, pool_cap, nchunk, nlev, n_tiles, L0;
};

__device__ __forceinline__ bool kf_pred(double yl, double yc, double yr)
{
    const double dp = yc - yl, dn = yr - yc;
    return ((dn > 0.0) && (dp <= 0.0)) || ((dn < 0.0) && (dp >= 0.0));      // ITD.py:59 on x and on -x
}

// Before the hand-over k_finalize (itd_kernels.hpp) clears the signals' KfSig and the flag words of the levels behind L0 and
// leaves the exclusive prefix of level L0's group sums in gpre.

// ---- hand-over: the level-L0 knots of every tile (flag words of the records the launch for level L0 - 1 left) with the values
//      of that level's input around them; exact ties of the caller's signal; the tile bases.  grid = (tiles, batch), 64 threads.
template <typename Tin, int TW>
__global__ __launch_bounds__(kWave) void k_kf_gather(KfWs ws, const Tin *__restrict__ x0, int64_t x0_stride, const double *__restrict__ xl,
                                                     int64_t xl_stride, int64_t n, const int32_t *__restrict__ counts,
                                                     const TileRec *__restrict__ recs, const SigState *__restrict__ state)
{
    // grid = (ceil(tiles / 2), batch): a wavefront takes TWO consecutive tiles, 32 lanes each (sparse levels hold a handful of knots
    // per tile: with a wavefront per tile the launch was four rounds of two dependent round trips; a lane takes knot hl, hl + 32, ...)
    constexpr int G2 = TW / 128;
    const int sig = blockIdx.y, lane = lane_id(), half = lane >> 5, hl = lane & 31;
    const int n_tiles = ws.n_tiles, n_groups = groups_of(n_tiles);
    const int t0 = 2 * blockIdx.x, t = t0 + half;
    const bool have = t < n_tiles;
    const SigState *st = state + sig;
    KfSig *ks = ws.sig + sig;
    // everything the tiles need is requested before the first use (one round trip in front of the knots' values)
    const int32_t *cnts = counts + (size_t)sig * n_tiles;
    const int g = t0 / kTilesPerGroup;                                   // (t0 is even: both tiles lie in the same 64-tile group)
    int acc = (lane < t0 - g * kTilesPerGroup) ? cnts[g * kTilesPerGroup + lane] : 0;
    const int c = have ? cnts[t] : 0, gp = ws.gpre[(size_t)sig * n_groups + g];
    const unsigned long long *rf = recs[(size_t)sig * n_tiles + (have ? t : t0)].flags;
    unsigned long long w[2 * G2];
#pragma unroll
    for (int q = 0; q < 2 * G2; ++q) w[q] = rf[q];
    const int tie_here = have ? ws.tie[(size_t)sig * n_tiles + t] : 0;
    const bool active = !st->fin_stopped && st->nan_mask == 0 && !st->in_nan && !st->l0_fail;
    if (t0 == 0 && lane == 0) {
        ks->active = active ? 1 : 0;
        ks->lend = -1;
        if (!active && !st->fin_stopped) ks->fail = kKfFailNonFinite;     // NaN rules / an unfinished level 0: not this path's
    }
    if (!active) return;
    if (t0 == 0 && lane < 4) ks->ends[ws.L0 & 1][lane] = st->ends[ws.L0 & 1][lane];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    const int c_first = __shfl(c, 0);                                    // tile t0's count
    const int base = acc + gp + (half ? c_first : 0);
    if (c == 0) {                                                        // an empty tile's record holds no flag words
#pragma unroll
        for (int q = 0; q < 2 * G2; ++q) w[q] = 0ull;
    }
    int32_t *first = ws.first + ((size_t)sig * ws.nlev) * (n_tiles + 1);
    if (have) {
        // the first fused level's flag words as the records hold them; the words of the levels behind it cleared (the steps set them
        // with atomic ORs; the sample pass reads them)
        if (hl == 0) first[t] = base;
        if (hl < 2 * G2) {
            unsigned long long mine = 0ull;
#pragma unroll
            for (int q = 0; q < 2 * G2; ++q) if (hl == q) mine = w[q];
            unsigned long long *tf = ws.tflags + ((size_t)sig * ws.nlev) * n_tiles * 8 + (size_t)t * 8;
            tf[hl] = mine;
            for (int li = 1; li < ws.nlev; ++li) tf[(size_t)li * n_tiles * 8 + hl] = 0ull;
        }
        if (t == n_tiles - 1 && hl == 0) {
            first[n_tiles] = base + c;
            ks->mlev[ws.L0] = base + c;
            ks->toff[ws.L0] = 0;
            if (base + c + kKfSticky + 4 > ws.cap) atomicOr(&ks->fail, kKfFailCapacity);
        }
    }
    const int64_t s = (int64_t)t * TW;
    const double *xs = xl + (int64_t)sig * xl_stride;
    if (c > 0 && base + c + 2 <= ws.cap) {
        // a lane per knot: the j-th knot of the tile from the flag words (within a 128-sample group the knots run E_0, O_0, E_1, O_1, ...),
        // so that the three values around every knot are requested in ONE round trip
        int32_t *P = ws.candP + ((size_t)sig * 2) * ws.cap;
        double *Tr = ws.candT + ((size_t)sig * 2) * ws.cap * 3;
        int cum[G2 + 1];
        cum[0] = 0;
#pragma unroll
        for (int gg = 0; gg < G2; ++gg) cum[gg + 1] = cum[gg] + __popcll(w[2 * gg]) + __popcll(w[2 * gg + 1]);
        for (int j = hl; j < c; j += 32) {
            int gg = 0;
            unsigned long long E = w[0], O = w[1];
#pragma unroll
            for (int q = 1; q < G2; ++q) if (j >= cum[q]) { gg = q; E = w[2 * q]; O = w[2 * q + 1]; }
            int jr = j;
#pragma unroll
            for (int q = 1; q < G2; ++q) if (gg == q) jr = j - cum[q];
            // smallest l with (knots of the group at pairs 0 .. l) > jr
            int lo = 0, hi = 63;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int mid = (lo + hi) >> 1;
                const unsigned long long mk = mid >= 63 ? ~0ull : ((1ull << (mid + 1)) - 1ull);
                const int f = __popcll(E & mk) + __popcll(O & mk);
                if (f > jr) hi = mid; else lo = mid + 1;
            }
            const int l = lo;
            const unsigned long long mb = (1ull << l) - 1ull;
            const int before = __popcll(E & mb) + __popcll(O & mb);
            const int odd = (((E >> l) & 1ull) && jr == before) ? 0 : 1;
            const int64_t pos = s + 128 * gg + 2 * l + odd;
            const size_t k = (size_t)base + 1 + j;
            const double a = xs[pos - 1], bb = xs[pos], cc = xs[pos + 1];
            P[k] = (int32_t)pos;
            Tr[3 * k] = a; Tr[3 * k + 1] = bb; Tr[3 * k + 2] = cc;
        }
    }
    // sticky candidates: sample n-2, and both samples of every exact tie of the caller's signal (rare: the tile was flagged)
    auto append = [&](int64_t p) {
        if (p < 1 || p > n - 2) return;
        const int idx = atomicAdd(&ks->n_sticky, 1);
        if (idx < kKfSticky) ks->spos[idx] = (int32_t)p;
        else atomicOr(&ks->fail, kKfFailTies);
    };
    if (have && hl == 0 && s <= n - 2 && n - 2 < s + TW) append(n - 2);
    for (int h = 0; h < 2; ++h) {
        if (!__shfl(tie_here, 32 * h)) continue;                         // (wave-uniform)
        const int tt = t0 + h;
        if (lane == 0) ws.tie[(size_t)sig * n_tiles + tt] = 0;           // the flags clean themselves: the fused level-0 launch only ever sets them
        const Tin *x = x0 + (int64_t)sig * x0_stride;
        const int64_t st_ = (int64_t)tt * TW;
        for (int i = lane; i < TW; i += kWave) {
            const int64_t p = st_ + i;
            if (p >= 1 && p <= n - 1 && x[p] == x[p - 1]) { append(p - 1); append(p); }
        }
    }
}

// knots of tile-relative positions <= q in a tile's flag words (flag_pos format)
__device__ __forceinline__ int kf_rank_in_tile(const unsigned long long *__restrict__ w, int q)
{
    const int g = q >> 7, r = q & 127;
    int acc = 0;
    for (int k = 0; k < 2 * g; ++k) acc += __popcll(w[k]);
    const int le = r >> 1;                                          // even positions 2 l <= r
    acc += __popcll(w[2 * g] & (le >= 63 ? ~0ull : ((1ull << (le + 1)) - 1ull)));
    if (r >= 1) {
        const int lo = (r - 1) >> 1;                                // odd positions 2 l + 1 <= r
        acc += __popcll(w[2 * g + 1] & (lo >= 63 ? ~0ull : ((1ull << (lo + 1)) - 1ull)));
    }
    return acc;
}

// sorts the sticky entries — the first `nsl` and the `napp` appends of the step in flight (slots kKfSticky + i) — by position
// and drops duplicates (one wavefront; nsl + napp <= 128); payload: ssk and the triples of parity `par`.  They end up in slots
// 0 .. kept-1.  Returns the number kept.
__device__ __forceinline__ int kf_sticky_sort(KfSig *ks, int nsl, int napp, int par, int lane, int32_t *s_p, int32_t *s_k, double (*s_t)[3])
{
    __shared__ int32_t s_first[kKfSticky + 64];
    const int cnt = nsl + napp;
    for (int i = lane; i < cnt; i += kWave) {        // everything is staged before anything is written back
        const int src = i < nsl ? i : kKfSticky + (i - nsl);
        s_p[i] = ks->spos[src];
        s_k[i] = ks->ssk[src];
        s_t[i][0] = ks->stri[par][src][0]; s_t[i][1] = ks->stri[par][src][1]; s_t[i][2] = ks->stri[par][src][2];
    }
    wave_sync();
    for (int i = lane; i < cnt; i += kWave) {        // 1 = the first occurrence of its position
        bool firsto = true;
        for (int j = 0; j < i; ++j) firsto = firsto && (s_p[j] != s_p[i]);
        s_first[i] = firsto ? 1 : 0;
    }
    wave_sync();
    int kept = 0;
    for (int i0 = 0; i0 < cnt; i0 += kWave) {
        const int i = i0 + lane;
        const bool keep = i < cnt && s_first[i] != 0;
        if (keep) {
            const int p = s_p[i];
            int r = 0;
            for (int j = 0; j < cnt; ++j) r += (s_first[j] != 0 && s_p[j] < p) ? 1 : 0;
            ks->spos[r] = p;
            ks->ssk[r] = s_k[i];
            ks->stri[par][r][0] = s_t[i][0]; ks->stri[par][r][1] = s_t[i][1]; ks->stri[par][r][2] = s_t[i][2];
        }
        kept += __popcll(__ballot(keep));
    }
    return kept;
}

// one wavefront: the sticky candidates sorted, with their list ranks and triples at level L0
template <int TW>
__device__ __forceinline__ void kf_sticky_init(const KfWs &ws, int sig, const double *__restrict__ xl, int64_t xl_stride, int lane)
{
    __shared__ int32_t s_p[kKfSticky + 64], s_k[kKfSticky + 64];
    __shared__ double s_t[kKfSticky + 64][3];
    KfSig *ks = ws.sig + sig;
    const int L0 = ws.L0, par = L0 & 1;
    const double *xs = xl + (int64_t)sig * xl_stride;
    const int cnt = min(ks->n_sticky, kKfSticky);
    const int32_t *first = ws.first + ((size_t)sig * ws.nlev) * (ws.n_tiles + 1);
    const unsigned long long *tf = ws.tflags + ((size_t)sig * ws.nlev) * ws.n_tiles * 8;
    for (int i = lane; i < cnt; i += kWave) {
        const int p = ks->spos[i];
        const int tt = p / TW;
        ks->ssk[i] = first[tt] + kf_rank_in_tile(tf + (size_t)tt * 8, p - tt * TW);
        ks->stri[par][i][0] = xs[p - 1]; ks->stri[par][i][1] = xs[p]; ks->stri[par][i][2] = xs[p + 1];
    }
    wave_sync();
    const int kept = kf_sticky_sort(ks, cnt, 0, par, lane, s_p, s_k, s_t);
    if (lane == 0) { ks->n_sticky = kept; ks->n_sticky_lev = kept; }
}

// ---- one knot-side step, part a: the level's table (B, S), the next level's triples and survival flags of the list entries, the
//      per-tile structures of THIS level for the sample pass, the next level's end samples; the sticky candidates' evaluation.
//      The signal's blocks take the list's 256-entry chunks in turn; the last block also evaluates the sticky ones.
struct KfList {       // a level's list with its two virtual end knots
    const int32_t *P; const double *Tr; int m; int32_t n1; double e0, e3;
    __device__ __forceinline__ int32_t pos(int k) const { return k <= 0 ? 0 : (k > m ? n1 : P[k]); }
    __device__ __forceinline__ double X(int k) const { return k <= 0 ? e0 : (k > m ? e3 : Tr[3 * (size_t)k + 1]); }
};
__device__ __forceinline__ double kf_B(const KfList &L, int k, double m0, double mn)
{
    if (k <= 0) return m0;
    if (k > L.m) return mn;
    const int32_t k0 = L.pos(k - 1), k1 = L.pos(k), k2 = L.pos(k + 1);
    const double x0 = L.X(k - 1), x1 = L.X(k), x2 = L.X(k + 1);
    const double frac = (double)(k1 - k0) / (double)(k2 - k0);
    const double tt = frac * (x2 - x0);
    const double u = x0 + tt;
    return 0.5 * u + 0.5 * x1;                               // ITD.py:107-110
}

// what a level's step works on (the same in every block of the signal)
struct KfLevel {
    KfSig *ks; int m, par, lp, toff; int32_t n1; double e0, e1, e2, e3, m0, mn; KfList L; KfEntry *pool;
};
__device__ __forceinline__ bool kf_level(const KfWs &ws, int sig, int lev, int64_t n, KfLevel &V)
{
    KfSig *ks = ws.sig + sig;
    V.ks = ks;
    V.m = ks->mlev[lev]; V.par = (lev - ws.L0) & 1; V.lp = lev & 1; V.toff = ks->toff[lev];
    if (V.toff + V.m + 2 > ws.pool_cap || V.m + kKfSticky + 4 > ws.cap) return false;
    V.n1 = (int32_t)(n - 1);
    V.e0 = ks->ends[V.lp][0]; V.e1 = ks->ends[V.lp][1]; V.e2 = ks->ends[V.lp][2]; V.e3 = ks->ends[V.lp][3];
    V.m0 = (V.e0 + V.e1) / 2.0; V.mn = (V.e2 + V.e3) / 2.0;                  // ITD.py:101-102
    V.L = KfList{ws.candP + ((size_t)sig * 2 + V.par) * ws.cap, ws.candT + ((size_t)sig * 2 + V.par) * ws.cap * 3, V.m, V.n1, V.e0, V.e3};
    V.pool = ws.pool + (size_t)sig * ws.pool_cap + V.toff;
    return true;
}

// the sticky candidates of a level (one wavefront): their next-level triples and whether they become knots that the list lacks;
// the segments their three samples lie in come from the list itself (a handful of entries each)
__device__ __forceinline__ void kf_sticky_eval(const KfLevel &V, int lane)
{
    KfSig *ks = V.ks;
    const KfList &L = V.L;
    const int m = V.m, lp = V.lp;
    const int cnt = ks->n_sticky_lev;
    for (int i = lane; i < cnt; i += kWave) {
        const int32_t s = ks->spos[i];
        const int sk = ks->ssk[i];
        const bool is_knot = sk >= 1 && L.pos(sk) == s;
        const int seg_l = is_knot ? sk - 1 : sk;
        const bool adj = sk + 1 <= m && L.pos(sk + 1) == s + 1;
        const int seg_r = adj ? sk + 1 : sk;
        const double xl_ = ks->stri[lp][i][0], xc = ks->stri[lp][i][1], xr = ks->stri[lp][i][2];
        auto mapv = [&](int k, double v) {
            const double Bk = kf_B(L, k, V.m0, V.mn), Bn = kf_B(L, k + 1, V.m0, V.mn);
            const double Xk = L.X(k);
            const double Sk = (Bn - Bk) / (L.X(k + 1) - Xk);           // ITD.py:115-116
            return Bk + Sk * (v - Xk);
        };
        const double yl = mapv(seg_l, xl_), yc = mapv(sk, xc);
        const double yr = (s + 1 == V.n1) ? 0.0 : mapv(seg_r, xr);     // baseline[n-1] stays 0, ITD.py:112-117
        ks->stri[lp ^ 1][i][0] = yl; ks->stri[lp ^ 1][i][1] = yc; ks->stri[lp ^ 1][i][2] = yr;
        ks->sins[i] = (!is_knot && kf_pred(yl, yc, yr)) ? 1 : 0;
        if (!(yl - yl == 0.0) || !(yc - yc == 0.0) || !(yr - yr == 0.0)) atomicOr(&ks->fail, kKfFailNonFinite);
    }
}

// One 256-entry chunk of a level's list (all threads of the block): the table entries, the next level's triple of the thread's
// entry (tri), its survival flag, the survivors of the chunk in front of it (before) and in all (returned); the per-tile
// structures of THIS level; chunk 0 / the last knot's chunk: the virtual end entries and the next level's end samples.
template <int TW>
__device__ __forceinline__ int kf_chunk_a(const KfWs &ws, int sig, int lev, const KfLevel &V, int c, int &fl_out, int &before_out,
                                          double (&tri)[3], int32_t &pos_out)
{
    __shared__ int32_t sP[kKfChunk + 4];
    __shared__ double sX[kKfChunk + 4], sB[kKfChunk + 4], sS[kKfChunk + 4];
    __shared__ int s_red[kKfChunk / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    KfSig *ks = V.ks;
    const KfList &L = V.L;
    const int m = V.m, lp = V.lp;
    const int32_t n1 = V.n1;
    const double e0 = V.e0, e1 = V.e1, e2 = V.e2, e3 = V.e3, m0 = V.m0, mn = V.mn;
    KfEntry *pool = V.pool;
    const int k0 = 1 + c * kKfChunk;
    __syncthreads();                                            // the previous chunk's LDS reads are done
    // entries k0 - 2 .. k0 + 257 (LDS index j <-> entry k0 - 2 + j)
    for (int j = tid; j < kKfChunk + 4; j += kKfChunk) {
        const int k = k0 - 2 + j;
        sP[j] = L.pos(k);
        sX[j] = L.X(k);
    }
    const int k = k0 + tid, j = tid + 2;
    double xl_ = 0.0, xr = 0.0;
    if (k >= 1 && k <= m) { xl_ = L.Tr[3 * (size_t)k]; xr = L.Tr[3 * (size_t)k + 2]; }   // (requested with the list: one round trip)
    __syncthreads();
    for (int jj = tid; jj < kKfChunk + 4; jj += kKfChunk) {
        const int kk = k0 - 2 + jj;
        double Bv = 0.0;
        if (jj >= 1 && jj <= kKfChunk + 2 && kk >= 0 && kk <= m + 1) {
            if (kk == 0) Bv = m0;
            else if (kk == m + 1) Bv = mn;
            else {
                const double frac = (double)(sP[jj] - sP[jj - 1]) / (double)(sP[jj + 1] - sP[jj - 1]);
                const double tt = frac * (sX[jj + 1] - sX[jj - 1]);
                const double u = sX[jj - 1] + tt;
                Bv = 0.5 * u + 0.5 * sX[jj];
            }
        }
        sB[jj] = Bv;
    }
    __syncthreads();
    for (int jj = tid; jj < kKfChunk + 4; jj += kKfChunk) {
        const int kk = k0 - 2 + jj;
        double Sv = 0.0;
        if (jj >= 1 && jj <= kKfChunk + 1 && kk >= 0 && kk <= m) Sv = (sB[jj + 1] - sB[jj]) / (sX[jj + 1] - sX[jj]);
        sS[jj] = Sv;
    }
    __syncthreads();
    int fl = 0;
    pos_out = 0;
    tri[0] = tri[1] = tri[2] = 0.0;
    if (k >= 1 && k <= m) {
        const int32_t pos = sP[j];
        const double Bk = sB[j], Sk = sS[j], Xk = sX[j];
        KfEntry en; en.X = Xk; en.B = Bk; en.S = Sk; en.pos = pos; en.pad = 0;
        pool[k] = en;
        const double yl = sB[j - 1] + sS[j - 1] * (xl_ - sX[j - 1]);
        const double yc = Bk + Sk * (Xk - Xk);
        double yr;
        if (pos + 1 == n1) yr = 0.0;                                        // baseline[n-1] stays 0, ITD.py:112-117
        else if (pos + 1 == sP[j + 1]) yr = sB[j + 1] + sS[j + 1] * (xr - sX[j + 1]);
        else yr = Bk + Sk * (xr - Xk);
        tri[0] = yl; tri[1] = yc; tri[2] = yr;
        pos_out = pos;
        fl = kf_pred(yl, yc, yr) ? 1 : 0;
        if (!(Bk - Bk == 0.0) || !(Sk - Sk == 0.0) || !(yl - yl == 0.0) || !(yr - yr == 0.0)) atomicOr(&ks->fail, kKfFailNonFinite);
        if (yl == yc || yc == yr) {                                         // an exact tie: rounding may move the knot next time
            const int ia = atomicAdd(&ks->n_app, 1);            // (a region of its own: the sticky block may be sorting the others)
            if (ia < 64) {
                const int idx = kKfSticky + ia;
                ks->spos[idx] = pos; ks->ssk[idx] = k; ks->sins[idx] = 0;
                ks->stri[lp ^ 1][idx][0] = yl; ks->stri[lp ^ 1][idx][1] = yc; ks->stri[lp ^ 1][idx][2] = yr;
            } else atomicOr(&ks->fail, kKfFailTies);
        }
        // this level's knots for the sample pass: flag word bit, knots in front of every tile
        const int li = lev - ws.L0;
        int32_t *first = ws.first + ((size_t)sig * ws.nlev + li) * (ws.n_tiles + 1);
        unsigned long long *tf = ws.tflags + ((size_t)sig * ws.nlev + li) * ws.n_tiles * 8;
        const int tt = pos / TW, q = pos - tt * TW;
        atomicOr(&tf[(size_t)tt * 8 + 2 * (q >> 7) + (q & 1)], 1ull << ((q & 127) >> 1));
        const int t_lo = k == 1 ? 0 : sP[j - 1] / TW + 1;
        for (int u = t_lo; u <= tt; ++u) first[u] = k - 1;
        if (k == m) for (int u = tt + 1; u <= ws.n_tiles; ++u) first[u] = m;
    }
    ws.flag[(size_t)sig * ws.cap + (k <= ws.cap - 1 ? k : 0)] = (k >= 1 && k <= m) ? fl : 0;
    // survivors of the chunk in front of the entry, and the chunk's total
    const unsigned long long bal = __ballot(fl);
    const int inw = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_red[wave] = __popcll(bal);
    __syncthreads();
    int before = inw, total = 0;
#pragma unroll
    for (int w = 0; w < kKfChunk / 64; ++w) { if (w < wave) before += s_red[w]; total += s_red[w]; }
    if (k >= 1 && k <= m) ws.pref[(size_t)sig * ws.cap + k] = before;
    if (c == 0 && tid == 0) {
        KfEntry en; en.X = e0; en.B = sB[1]; en.S = sS[1]; en.pos = 0; en.pad = 0;     // entry 0: the virtual knot at sample 0
        pool[0] = en;
        double *ne = ks->ends[lp ^ 1];
        ne[0] = sB[1] + sS[1] * (e0 - e0);
        ne[1] = (m >= 1 && sP[2] == 1) ? sB[2] + sS[2] * (e1 - sX[2]) : sB[1] + sS[1] * (e1 - e0);
        if (m == 0) {
            int32_t *first = ws.first + ((size_t)sig * ws.nlev + (lev - ws.L0)) * (ws.n_tiles + 1);
            for (int u = 0; u <= ws.n_tiles; ++u) first[u] = 0;
        }
    }
    if (k == m || (m == 0 && c == 0 && tid == 0)) {          // the thread of the last knot (or of entry 0 when there is none)
        const int jm = m == 0 ? 1 : j;                       // LDS index of entry m
        KfEntry en; en.X = e3; en.B = mn; en.S = 0.0; en.pos = n1; en.pad = 0;          // entry m + 1: the virtual knot at sample n-1
        pool[m + 1] = en;
        double *ne = ks->ends[lp ^ 1];
        ne[2] = sB[jm] + sS[jm] * (e2 - sX[jm]);             // sample n-2 lies in the last knot's segment (knots are <= n-2)
        ne[3] = 0.0;
    }
    fl_out = fl;
    before_out = before;
    return total;
}

// part a as a launch of its own (the two-launch form of a step): every chunk's results go through memory
template <int TW>
__device__ __forceinline__ void kf_step_a(const KfWs &ws, int sig, int lev, const KfLevel &V, bool fits, int blk, int nblk,
                                          const double *__restrict__ xl, int64_t xl_stride)
{
    if (!fits) {
        if (threadIdx.x == 0) atomicOr(&ws.sig[sig].fail, kKfFailCapacity);
        return;
    }
    if (blk == nblk - 1 && threadIdx.x < kWave) {
        if (lev == ws.L0) {                   // the hand-over's sticky candidates: sorted, ranked, with their triples
            kf_sticky_init<TW>(ws, sig, xl, xl_stride, threadIdx.x);
            __threadfence_block();
            wave_sync();
        }
        kf_sticky_eval(V, threadIdx.x);
    }
    const int nck = max(1, (V.m + kKfChunk - 1) / kKfChunk);      // chunk 0 runs even for an empty list (entry 0, the end samples)
    // (with more than one block the last one works on the sticky candidates only)
    const int nwork = nblk > 1 ? nblk - 1 : 1;
    for (int c = (nblk > 1 && blk == nblk - 1) ? nck : blk; c < nck; c += nwork) {
        int fl, before;
        double tri[3];
        int32_t pos;
        const int total = kf_chunk_a<TW>(ws, sig, lev, V, c, fl, before, tri, pos);
        const int k = 1 + c * kKfChunk + (int)threadIdx.x;
        if (k >= 1 && k <= V.m) {
            double *nt = ws.newT + ((size_t)sig * ws.cap) * 3 + 3 * (size_t)k;
            nt[0] = tri[0]; nt[1] = tri[1]; nt[2] = tri[2];
        }
        if (threadIdx.x == 0) ws.cnt[(size_t)sig * ws.nchunk + c] = total;
    }
}

// ---- part b: the survivors, and the sticky candidates that became knots, in order into the other list buffer; the stop rules
//      (ITD.py:400-426); the sticky candidates' ranks in the new list.  grid = (blocks + 1, batch), 256 threads.
__device__ __forceinline__ void kf_step_b(const KfWs &ws, int sig, int lev, int max_iteration, int blk, int nblk)
{
    __shared__ int32_t s_ins[kKfSticky + 64];
    __shared__ int s_nins, s_red[kKfChunk / 64];
    __shared__ int32_t s_p[kKfSticky + 64], s_k[kKfSticky + 64];
    __shared__ double s_t[kKfSticky + 64][3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    KfSig *ks = ws.sig + sig;
    const int m = ks->mlev[lev], par = (lev - ws.L0) & 1, lp = lev & 1;
    const int nck = (m + kKfChunk - 1) / kKfChunk;
    const int32_t *P = ws.candP + ((size_t)sig * 2 + par) * ws.cap;
    int32_t *nP = ws.candP + ((size_t)sig * 2 + (par ^ 1)) * ws.cap;
    double *nT = ws.candT + ((size_t)sig * 2 + (par ^ 1)) * ws.cap * 3;
    const double *newT = ws.newT + ((size_t)sig * ws.cap) * 3;
    const int32_t *flag = ws.flag + (size_t)sig * ws.cap, *pref = ws.pref + (size_t)sig * ws.cap;
    const int32_t *cnt = ws.cnt + (size_t)sig * ws.nchunk;
    // the sticky candidates to insert, in position order (spos[0 .. n_sticky_lev) is sorted)
    if (tid == 0) s_nins = 0;
    __syncthreads();
    const int nsl = ks->n_sticky_lev;
    if (wave == 0) {
        int base = 0;
        for (int i0 = 0; i0 < nsl; i0 += kWave) {
            const int i = i0 + lane;
            const bool ins = i < nsl && ks->sins[i] != 0;
            const unsigned long long bal = __ballot(ins);
            if (ins) s_ins[base + __popcll(bal & ((1ull << lane) - 1ull))] = ks->spos[i];
            base += __popcll(bal);
        }
        if (lane == 0) s_nins = base;
    }
    __syncthreads();
    const int nins = s_nins;
    {
      const int nwork = nblk > 1 ? nblk - 1 : 1;
      for (int c = (nblk > 1 && blk == nblk - 1) ? nck : blk; c < nck; c += nwork) {
        __syncthreads();
        int acc = 0;
        for (int q = tid; q < c; q += kKfChunk) acc += cnt[q];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
        if (lane == 0) s_red[wave] = acc;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < kKfChunk / 64; ++w) base += s_red[w];
        const int k = 1 + c * kKfChunk + tid;
        if (k <= m && flag[k]) {
            const int32_t pos = P[k];
            int ib = 0;
            for (int i = 0; i < nins; ++i) ib += s_ins[i] < pos ? 1 : 0;
            const int out = base + pref[k] + ib + 1;
            nP[out] = pos;
            nT[3 * (size_t)out] = newT[3 * (size_t)k]; nT[3 * (size_t)out + 1] = newT[3 * (size_t)k + 1]; nT[3 * (size_t)out + 2] = newT[3 * (size_t)k + 2];
        }
      }
      if (blk != nblk - 1) return;
      __syncthreads();
    }
    // ---- the sticky block: inserts, bookkeeping, stop rules ----
    // exclusive prefix of the chunk counts, in place in global memory's place: LDS holds up to kKfPre of them, the rest is summed
    __shared__ int s_cpre[kKfPre];
    int carry = 0;
    for (int q0 = 0; q0 < nck; q0 += kKfChunk) {
        const int q = q0 + tid;
        const int v = q < nck ? cnt[q] : 0;
        __syncthreads();
        // inclusive scan of v over the block through wave ballots / shuffles
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 63) s_red[wave] = inc;
        __syncthreads();
        int wb = 0;
#pragma unroll
        for (int w = 0; w < kKfChunk / 64; ++w) if (w < wave) wb += s_red[w];
        if (q < nck && q < kKfPre) s_cpre[q] = carry + wb + inc - v;
        int tot = 0;
#pragma unroll
        for (int w = 0; w < kKfChunk / 64; ++w) tot += s_red[w];
        carry += tot;
        __syncthreads();
    }
    const int total = carry;
    if (wave != 0) return;
    const int napp = min(ks->n_app, 64), ns_all = nsl + napp;
    for (int i0 = 0; i0 < ns_all; i0 += kWave) {       // a lane per candidate
        const int i = i0 + lane;
        if (i < ns_all) {
            const int src = i < nsl ? i : kKfSticky + (i - nsl);
            const int32_t s = ks->spos[src];
            // old entries: ssk = list knots at or before s; fresh appends (i >= nsl): ssk = their own list index (they ARE list knots)
            const int K = ks->ssk[src];
            int sv = 0;
            if (K > 0) {
                const int ck = (K - 1) / kKfChunk;
                int a = ck < kKfPre ? s_cpre[ck] : 0;
                if (ck >= kKfPre) for (int q = 0; q < ck; ++q) a += cnt[q];
                sv = a + pref[K] + flag[K];                     // survivors among the list entries 1 .. K
            }
            int ile = 0, ilt = 0;
            for (int q = 0; q < nins; ++q) { ile += s_ins[q] <= s ? 1 : 0; ilt += s_ins[q] < s ? 1 : 0; }
            if (i < nsl && ks->sins[i]) {
                const int out = sv + ilt + 1;
                nP[out] = s;
                nT[3 * (size_t)out] = ks->stri[lp ^ 1][i][0]; nT[3 * (size_t)out + 1] = ks->stri[lp ^ 1][i][1]; nT[3 * (size_t)out + 2] = ks->stri[lp ^ 1][i][2];
            }
            ks->ssk[src] = sv + ile;      // knots of the NEW list at or before s
        }
    }
    __threadfence_block();
    wave_sync();
    const int kept = kf_sticky_sort(ks, nsl, napp, lp ^ 1, lane, s_p, s_k, s_t);
    if (lane == 0) {
        if (kept > kKfSticky) atomicOr(&ks->fail, kKfFailTies);
        ks->n_sticky = kept;
        ks->n_sticky_lev = kept;
        ks->n_app = 0;
        const int m_next = total + nins;
        ks->mlev[lev + 1] = m_next;
        ks->toff[lev + 1] = ks->toff[lev] + m + 2;
        if (m_next < 2) { ks->lend = lev; ks->natural = 1; }                 // "No more decompositions possible", ITD.py:404-416
        else if (lev > max_iteration) { ks->lend = lev; ks->natural = 0; }  // "Out of time!", ITD.py:418-426
    }
}

// ---- the knot side as launches: hand-over (sticky candidates), then two launches per level.  grid = (blocks, batch), 256
//      threads; the blocks take the list's chunks in turn, the last block works on the sticky candidates.  Measured alternatives,
//      both correct and both slower than these 12 launches of ~7.5 us (2^24 samples, 200 k knots at level 3): ONE persistent launch
//      with grid barriers between the phases (188 us with 128 blocks, 864 us with 768), and a step as one launch with single-pass
//      compaction (ticketed work items, decoupled look-back over level-tagged status words, agent-scope release / acquire:
//      32 us per level, 93 us at level 3) — the phases are short chains of dependent loads, and in-launch synchronisation across
//      8 XCDs costs more than the launch boundary it replaces (profiles/r03/kf_knot_side_alternatives.txt).
template <int TW>
__global__ __launch_bounds__(kKfChunk) void k_kf_step_a(KfWs ws, int lev, int64_t n, const double *__restrict__ xl, int64_t xl_stride)
{
    const KfSig *ks = ws.sig + blockIdx.y;
    // (the level's bookkeeping is requested together with the three words of the test: one round trip instead of two)
    KfLevel V;
    const bool fits = kf_level(ws, blockIdx.y, lev, n, V);
    if (!ks->active || ks->lend >= 0 || ks->fail) return;      // (a failure raised by a running block of this launch: the others
    kf_step_a<TW>(ws, blockIdx.y, lev, V, fits, blockIdx.x, gridDim.x, xl, xl_stride);   //  may or may not see it — either way the result is discarded)
}
__global__ __launch_bounds__(kKfChunk) void k_kf_step_b(KfWs ws, int lev, int max_iteration)
{
    const KfSig *ks = ws.sig + blockIdx.y;
    if (!ks->active || ks->lend >= 0 || ks->fail) return;
    kf_step_b(ws, blockIdx.y, lev, max_iteration, blockIdx.x, gridDim.x);
}

// ---- the sample pass: every fused level of a tile in registers.  grid = (tiles, batch), 64 threads. -----------------------
#ifndef ITD_KF_APPLY_WAVES
#define ITD_KF_APPLY_WAVES 0      // A/B builds: wavefronts per SIMD the sample pass is compiled for (0 = the compiler's choice: 6)
#endif
template <int TW, int CAP>
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
    const int sig = blockIdx.y, lane = lane_id();
    KfSig *ks = ws.sig + sig;
    if (!ks->active || ks->fail) return;
    const int lend = ks->lend, natural = ks->natural;
    if (lend < 0) return;
    const int n_tiles = ws.n_tiles;
    const int t = xcd_item(blockIdx.x, n_tiles);
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
    const int32_t *first = ws.first + ((size_t)sig * ws.nlev) * (n_tiles + 1);
    const unsigned long long *tf = ws.tflags + ((size_t)sig * ws.nlev) * n_tiles * 8;
    const KfEntry *pool = ws.pool + (size_t)sig * ws.pool_cap;
    double *rows_s = rows + (int64_t)sig * rows_stride + si;
    double *bases_s = bases ? bases + (int64_t)sig * bases_stride + si : nullptr;
    unsigned long long wcur = lane < 2 * G2 ? tf[(size_t)t * 8 + lane] : 0ull;       // this level's flag words, lane j holds word j
    bool bad = false;
    for (int lev = ws.L0; lev <= lend; ++lev) {
        const int li = lev - ws.L0;
        const bool last = lev == lend;
        const int kfst = first[(size_t)li * (n_tiles + 1) + t], kn = first[(size_t)li * (n_tiles + 1) + t + 1] - kfst;
        if (kn > CAP) { if (lane == 0) atomicOr(&ks->fail, kKfFailCapacity); return; }
        // the next level's flag words (what the values computed below must reproduce)
        const unsigned long long wnext = (!last && lane < 2 * G2) ? tf[((size_t)(li + 1) * n_tiles + t) * 8 + lane] : 0ull;
        const KfEntry *tab = pool + ks->toff[lev] + kfst;
        for (int r = lane; r < kn + 2; r += kWave) {
            const KfEntry en = tab[r];
            s_X[r] = en.X; s_B[r] = en.B; s_S[r] = en.S; s_P[r] = en.pos;
        }
        wave_sync();
        WaveMasks wm;
        wm.lo = (unsigned)wcur; wm.hi = (unsigned)(wcur >> 32);
        double *row = rows_s + (int64_t)lev * n;
        const __amdgpu_buffer_rsrc_t r_row = tile_rsrc32(row, rem, 8);
        const __amdgpu_buffer_rsrc_t r_bas = tile_rsrc32(bases_s ? bases_s + (int64_t)lev * n : row, (bases_s && !last) ? rem : 0, 8);
        int gbase = 0;
        bool nonfin = false;
#pragma unroll
        for (int g = 0; g < G2; ++g) {
            const unsigned long long E = wm.get(2 * g), O = wm.get(2 * g + 1);
            const int bE = lane_bit(E), bO = lane_bit(O);
            const int Le = mbcnt64(O, mbcnt64(E, gbase)) + bE;      // knots of the tile at or before the even sample = its slot
            const int Lo = Le + bO;
            const int p = 128 * g + 2 * lane;
            const double xe = xr[g][0], xo = xr[g][1];
            double be = s_B[Le] + s_S[Le] * (xe - s_X[Le]);          // ITD.py:114-117
            double bo = s_B[Lo] + s_S[Lo] * (xo - s_X[Lo]);
            if (rem <= TW) {                                         // baseline[n-1] is never written (stays 0), ITD.py:112-117
                if (p >= nrem - 1) be = 0.0;
                if (p + 1 >= nrem - 1) bo = 0.0;
            }
            double re, ro;
            if (last && natural) { re = xe; ro = xo; }               // row c = baselines[c-1], ITD.py:404-416
            else if (last) { re = (xe - be) + be; ro = (xo - bo) + bo; }   // "Out of time!": rotation + baseline, ITD.py:420
            else { re = xe - be; ro = xo - bo; }                     // ITD.py:119
            tile_store2<true>(r_row, p * 8, re, ro);
            tile_store2<true>(r_bas, p * 8, be, bo);
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
        // the next level's knots from the values just computed: the exact predicate on the actual samples
        unsigned long long E2[G2], O2[G2];
        {
            double d0[G2], d1[G2];
#pragma unroll
            for (int g = 0; g < G2; ++g) {
                const double fill = (g == 0) ? n_lo : wave_dpp<0x13C>(0.0, xr[g > 0 ? g - 1 : 0][1]);
                const double left = wave_dpp<0x138>(fill, xr[g][1]);
                d0[g] = xr[g][0] - left;
                d1[g] = xr[g][1] - xr[g][0];
            }
            knot_predicate<double, G2>(d0, d1, n_hi - xr[G2 - 1][1], E2, O2);
            if (si == 0 || nrem <= TW + 1) {       // first and last sample are never knots (ITD.py:70-73), nothing beyond n-2
                const int lo = si > 0 ? 0 : 1, hi = nrem - 2;
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    E2[g] &= bit_range((lo - 128 * g + 1) >> 1, (hi - 128 * g) >> 1);
                    O2[g] &= bit_range((lo - 128 * g) >> 1, (hi - 128 * g - 1) >> 1);
                }
            }
        }
        if (!last) {
            WaveMasks wn;
            wn.lo = (unsigned)wnext; wn.hi = (unsigned)(wnext >> 32);
#pragma unroll
            for (int g = 0; g < G2; ++g) bad = bad || (wn.get(2 * g) != E2[g]) || (wn.get(2 * g + 1) != O2[g]);
        } else {
            int tot = 0;
#pragma unroll
            for (int g = 0; g < G2; ++g) tot += __popcll(E2[g]) + __popcll(O2[g]);
            if (lane == 0 && tot) atomicAdd(&ks->m_exact, tot);
        }
        x_lo = n_lo;
        x_hi = n_hi;
        wcur = wnext;
    }
    if (bad && lane == 0) atomicOr(&ks->fail, kKfFailVerify);
}

// The verdict of the fused levels — what k_kf_finish did as a launch of its own — is drawn on the host by itd_get_summary from
// the head of KfSig (itd_engine.hip: kf_verdict).

}  // namespace itd
