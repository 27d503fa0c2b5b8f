// itd_engine.hip — host side of libpyitd_hip.so: the C ABI declared in include/pyitd_hip.h.
//
// The level loop of the reference driver (ITD.itd, ITD.py:384-432) is enqueued here as a fixed
// sequence of launches on one HIP stream with NO host synchronisation between levels: the stop rule
// (`num_extrema < 2`, ITD.py:404) is evaluated on the device by block 0 of k_extract and later launches of a
// stopped signal return at once; k_finalize performs the row fix-up.  The host reads one small
// per-signal summary at the end (itd_get_summary).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <atomic>
#include <chrono>
#include <dlfcn.h>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/pyitd_hip.h"
#include "itd_kernels.hpp"
#include "itd_resident.hpp"
#include "itd_knotfirst.hpp"
#include "itd_cubic.hpp"
#include "itd_detect_fast.hpp"
#include "itd_stream.hpp"
#include "itd_tfe.hpp"
#include "itd_spline.hpp"
#include "itd_nak.hpp"
#include "itd_wpe.hpp"
#include "itd_meitd.hpp"

#ifndef ITD_TILE
#define ITD_TILE 512
#endif

using namespace itd;

namespace {
constexpr int T = ITD_TILE;
#ifndef ITD_SLOT_PAD
#define ITD_SLOT_PAD 0
#endif
constexpr int64_t kSlotPad = ITD_SLOT_PAD;   // elements (multiple of 2 keeps the slots 16-byte aligned)
static_assert(T % 128 == 0 && T / 64 <= kMaxGroups, "tile geometry: whole 8/16-byte loads per lane, <= 8 flag words per record");

// per-signal state + the (padded) group sums of all three rotating buffers, one launch
__global__ void k_init_state(SigState *st, int batch, int32_t *gsum, int64_t gsum_elems, int keep_in_nan = 0)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < gsum_elems; i += (int64_t)gridDim.x * blockDim.x)
        gsum[i] = 0;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int in_nan = st[b].in_nan;
    sig_state_reset(st + b);
    if (keep_in_nan) st[b].in_nan = in_nan;   // the NaN-input repeat needs to know which signals hold one (k_nan_level0)
}

// ---- device-visible validity and the device-side repair (itd_set_valid_flags, itd_set_device_repair) ----
// One thread per signal, behind the last launch of a decomposition: the fused levels' verdict merged into the signal's state (what
// kf_verdict does on the host when the summary is read), then valid[b] = 1 if the rows in the caller's buffer are final; need[b] = 1
// if the optimistic forms fell short for this signal (fused levels refused, fused level 0 out of reach, resident form met a
// non-finite value) and a level-by-level run would repair it.  A NaN in the caller's signal is neither (the host repeats such a
// call the way the reference runs it): valid = 0, need = 0.
__global__ void k_verdict(SigState *__restrict__ state, int batch, const KfSig *__restrict__ kf, int L0, int nan_follow,
                          int32_t *__restrict__ valid, int32_t *__restrict__ need, const unsigned long long *__restrict__ pipe_gave_up = nullptr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    SigState &st = state[b];
    // (a pipelined batch whose gate gave up — k_kf_gate — may have run a sample pass in front of its knot side: every signal is refused)
    if (kf && pipe_gave_up && *pipe_gave_up) st.kf_fail = kKfFailWait;
    else if (kf) {
        const KfSig &ks = kf[b];
        if (!ks.active) { if (ks.fail) st.kf_fail = ks.fail; }
        else {
            int fail = ks.fail;
            const int lend = ks.lend;
            if (lend < 0) fail |= kKfFailCapacity;
            // the last pending baseline feeds only the stop test (ITD.py:400-404): its exact count must take the same side of 2
            if (!fail && (ks.natural ? ks.m_exact >= 2 : ks.m_exact < 2)) fail |= kKfFailVerify;
            // (the levels' knot counts — what the stop rules were drawn from — the sample pass has compared with its verified flag words'.)
            // A level that was decomposed had at least 2 knots (ITD.py:404)
            for (int j = L0; !fail && j <= lend; ++j) if (j > L0 && ks.mlev[j] < 2) fail |= kKfFailVerify;
            if (fail) st.kf_fail = fail;
            else if (ks.cont) {         // capped fused levels that did not stop: the level launches behind them wrote the rest of the state
                for (int j = L0 + 1; j <= lend; ++j) st.m[j] = ks.mlev[j];
            } else {
                for (int j = L0 + 1; j <= lend; ++j) st.m[j] = ks.mlev[j];
                st.m[lend + 1] = ks.m_exact;
                st.fin_stopped = ks.natural;
                st.fin_stop_level = ks.natural ? lend + 1 : -1;
            }
        }
    }
    const int nan_in = st.in_nan != 0;
    const int short_fall = !nan_in && (st.kf_fail != 0 || st.l0_fail != 0 || st.res_fail != 0);
    if (need) need[b] = short_fall;
    if (valid) valid[b] = (!short_fall && !(nan_in && nan_follow)) ? 1 : 0;
}

// in front of the repair's launches: the set of states they work on — a signal that needs the repair starts from the initial
// state, every other one carries a copy of its final state and the skip flag (the repair's launches return at once for it); the
// set's group sums cleared
__global__ void k_repair_init(const SigState *__restrict__ from, SigState *__restrict__ to, const int32_t *__restrict__ need, int batch,
                              int32_t *gsum, int64_t gsum_elems)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < gsum_elems; i += (int64_t)gridDim.x * blockDim.x)
        gsum[i] = 0;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    if (need[b]) {
        // (bits 0 .. 2: which optimistic form fell short; from bit 3 on the fused levels' own failure bits, so that the host's back-offs
        //  — smaller ranges only for a list that outgrew its workgroup — see the reason as they do without the device-side repair)
        const int why = (from[b].kf_fail ? 1 : 0) | (from[b].l0_fail ? 2 : 0) | (from[b].res_fail ? 4 : 0) | ((from[b].kf_fail & 31) << 3);
        sig_state_reset(to + b);
        to[b].skip = -why;
    } else { to[b] = from[b]; to[b].skip = 1; }
}

// behind the repair: what it repaired is final now (a level-by-level run cannot fall short; a NaN it met in the signal stays the host's)
__global__ void k_verdict_repaired(const SigState *__restrict__ state, int batch, const int32_t *__restrict__ need, int nan_follow,
                                   int32_t *__restrict__ valid)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch || !need[b]) return;
    valid[b] = (state[b].in_nan && nan_follow) ? 0 : 1;
}

// itd_debug_int_ratio_check: int_ratio(a, b) against the compiler's full division for every pair 0 <= a <= b <= max_den (b >= 1) and
// for pseudo-random pairs up to 2^31 - 1: the number of pairs whose bit patterns differ
__global__ void k_int_ratio_check(int max_den, unsigned long long *bad)
{
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (long long b = 1 + tid; b <= max_den; b += nthreads)
        for (int a = 0; a <= (int)b; ++a) {
            volatile double x = (double)a, y = (double)(int)b;          // (volatile: the reference stays the emitted division)
            mine += dbits(int_ratio(a, (int)b)) != dbits(x / y);
        }
    unsigned long long h = 0x9e3779b97f4a7c15ull * (unsigned long long)(tid + 1);
    for (int k = 0; k < 4096; ++k) {
        h ^= h << 13; h ^= h >> 7; h ^= h << 17;
        const int b = (int)((h >> 1) & 0x7fffffff) | 1, a = (int)((h >> 33) & 0x7fffffff) % b;
        volatile double x = (double)a, y = (double)b;
        mine += dbits(int_ratio(a, b)) != dbits(x / y);
    }
    if (mine) atomicAdd(bad, mine);
}

// totals[2b + 1] = signal b holds a NaN (k_compact): OR them into one flag
__global__ void k_or_nan_flags(const int32_t *__restrict__ totals, int batch, int32_t *__restrict__ flag)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && totals[2 * b + 1]) atomicOr(flag, 1);
}

// up to 32 counts for a polling host: self-validating words (itd_kernels.hpp: small_put)
__global__ void k_copy_words(const int32_t *__restrict__ src, unsigned long long *__restrict__ words, int cnt, int32_t seq)
{
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) small_put(words, i, (uint32_t)src[i], (uint32_t)seq);
}

__global__ void k_widen_idx(const int32_t *__restrict__ src, int64_t *__restrict__ dst, int64_t cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) dst[i] = src[i];
}
}  // namespace

#ifdef ITD_DEBUG_GAP
// experiment build only: an idle gap of ITD_DEBUG_GAP microseconds in front of the sample pass (does the memory system use it?)
namespace { __global__ void k_debug_gap(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8); } }
#endif

struct itd_engine {
    int device = 0;
    int64_t max_n = 0;
    int32_t max_batch = 0;
    int64_t max_tiles = 0;
    hipStream_t own_stream = nullptr;
    // workspace
    int32_t *d_lists = nullptr;    // [tiles][T]  per-tile knot lists, written only for the API helpers (k_compact)
    int32_t *d_counts = nullptr;   // [2][batch][tiles]  knots per tile, double buffered by level parity
    TileRec *d_recs = nullptr;     // [2][batch][tiles]  head/tail knot records, double buffered by level parity
    int64_t tiles_half = 0;        // elements per counts/recs buffer
    int32_t *d_gsum = nullptr;     // [2][3][batch][groups*pitch]: per-64-tile knot totals, rotating by level % 3; two sets (below)
    int64_t gsum_third = 0;        // elements per buffer
    int32_t *d_kidx = nullptr;     // [max_n + 2]  ordered knot indices for the API helpers (single signal)
    int32_t *d_total = nullptr;    // [1] knot total written by k_compact
    double *d_pp = nullptr;        // [batch][3][pp_pitch] rotating baselines (slot = level % 3)
    int64_t pp_pitch = 0;          // elements between consecutive slots: max_n + kSlotPad (breaks the power-of-two distance)
    SigState *d_state = nullptr;   // [2][batch]
    SigState *h_state = nullptr;   // pinned
    char *h_kf = nullptr;          // pinned: the heads of the fused levels' KfSig (kKfSigHead bytes per signal), on demand
    // Per-signal states and group sums exist twice.  A decomposition works on the set the previous one did not use, and its last
    // launch (k_finalize) re-initialises the other set for the call after it: no initialising launch in front of a decomposition,
    // and the summary of the last call stays readable.  dirty_*: the leading part of a set that may not be in its initial state
    // (signals / group-sum elements per buffer); a call that finds its set dirty initialises it with a launch of its own.
    int cur_set = 0;
    int32_t dirty_sig[2] = {0, 0};
    int64_t dirty_gs[2] = {0, 0};
    // workspace of the single-level helpers (itd_detect_*, itd_baseline_extract_*): one signal, apart from the
    // decomposition's, so a helper call never disturbs a decomposition that is still in flight or not yet summarised
    int32_t *d_hcounts = nullptr;  // [2][tiles]
    TileRec *d_hrecs = nullptr;    // [2][tiles]
    int32_t *d_hgsum = nullptr;    // [3][groups*pitch]
    int64_t hgsum_third = 0;
    SigState *d_hstate = nullptr;  // [1]
    int32_t chunk = 0;             // signals per launch sequence of a batched decomposition (0 = automatic, see enqueue_decompose)
    int32_t batch_streams = 2;     // chunks of a batch rotate over this many streams (itd_set_batch_streams): 1 .. kMaxBatchStreams
    hipStream_t aux_stream[3] = {nullptr, nullptr, nullptr};   // the others besides the caller's, created on demand
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // the batch pipeline (itd_set_batch_pipeline; enqueue_decompose): the fused levels' knot side of chunk k + 1 on aux_stream[0] beside the
    // memory-bound launches of the caller's stream
    int32_t batch_pipeline = 0;                    // off by default: measured 12.8-13.0 ms against the rotating chunks' 11.9-12.3 on 512 x 2^20 (profiles/r06/experiments)
    std::vector<hipEvent_t> ev_pipe;               // two per chunk: the chunk's level launches are done (caller's stream) / its knot side is (aux_stream[0])
    unsigned long long *d_kf_started = nullptr;    // KfWs::started: knot-side workgroups that have started, over the engine's life
    unsigned long long kf_started_target = 0;      // ... as many as have been launched
    long long pipe_gate_timeout = 5000000;         // ticks of the 100 MHz clock the pipeline's waits last at most (50 ms: longer than a knot side
                                                   // whose own halo waits are given up, ITD_KC_TIMEOUT)

    int32_t resident_mode = ITD_RESIDENT_AUTO;   // short signals as one workgroup each, one launch (itd_set_resident_mode)
    int32_t resident_off_left = 0;  // automatic mode: decompositions still to run level by level after a resident call met a non-finite value
    int32_t resident_repeats = 0;   // how often itd_get_summary had to repeat a resident call level by level
    bool resident_attr[12] = {};
    bool nak_small_attr = false;    // hipFuncSetAttribute done for k_nak_small<true>
    bool meitd_attr[2] = {};        // ... for k_meitd_small<false / true>
    // a few scalars per call come back to the host in MEITD's operators (counts, six sums): 256 bytes of pinned host memory that the
    // GPU writes directly (mapped, coherent) — no copy behind the launch, just the stream's synchronisation (a pageable destination
    // cost ~15 us per call: 110 calls per MEITD run)
    void *h_small = nullptr, *d_small = nullptr;
    int32_t small_seq = 0;          // the number of the call whose scalars are awaited: every word of a result carries it in its high half (small_put)
    int32_t resident_window = 0;    // segments per pass over a level's ranks (itd_set_resident_window; 0 = automatic)   // hipFuncSetAttribute done per kernel instance
    // the fused sparse levels (itd_knotfirst.hpp): workspace (allocated at first use), mode, first fused level
    void *d_kf = nullptr; size_t kf_bytes = 0;
    std::vector<void *> kf_retired;  // earlier, smaller workspaces: a captured graph may still hold their pointers — kept until the engine is destroyed
    KfWs kf{};                       // pointers into d_kf, for signal 0
    int32_t fuse_mode = ITD_FUSE_AUTO, fuse_level = 0, fuse_off_left = 0, fuse_repeats = 0;   // fuse_level 0: automatic (kf_first_level)
    int32_t fuse_off_span = 16;                      // calls that run level by level after the next whole-call refusal: doubles with every refusal that
                                                     // follows a back-off directly (a workload the fused form cannot deliver — periodic, collapsing input —
                                                     // pays one wasted attempt in 17, then 33, ... 1025 calls), back to 16 after a delivered call
    bool fuse_probe = false;                         // the call being summarised was the first fused attempt after a back-off
    // capped fused levels: a workload whose fused form fails at the same level every time keeps the fused form for the levels in front of it
    int32_t fuse_cap = 0;                            // itd_set_fuse_cap: the first level NOT fused (0 = whatever the engine has learned)
    int32_t fuse_cap_auto = 0;                       // learned from a refusal's KfSig::fail_lev (0 = none)
    int32_t fuse_cap_calls = 0;                      // delivered calls under the learned cap since the last probe
    int32_t fuse_cap_span = 16;                      // ... after this many the next call tries all levels again (a probe); doubles, up to 1024,
                                                     // every time a probe is refused at the learned level again
    bool last_pipelined = false;                     // the last call was a pipelined batch (its gates' give-up word is read with its summary)
    bool pipe_word_unread = false;                   // a pipelined batch has been enqueued since the give-up word was last read
    unsigned long long *h_pipe = nullptr;            // pinned: that word on the host
    int last_kf_cap = 0, last_kf_cap_form = 0;       // the cap of the last call (as enqueued: itd_get_last_fuse_cap; 0 = all levels fused)
    bool kf_force_tickets = false;                   // a halo wait was given up on this engine (kKfFailWait): workgroup ids are tickets from then on
    bool fuse_level2_off = false;                    // automatic first fused level: a level-2 list has outgrown its workgroup, level 3 from then on
    bool fuse_no_memory = false;                     // the fused levels' workspace could not be allocated: level by level from then on
    // device-visible validity / device-side repair (itd_set_valid_flags, itd_set_device_repair)
    int32_t *valid_dev = nullptr;                    // the caller's [batch] words, written behind every decomposition; NULL = none
    bool device_repair = false;
    int32_t *d_need = nullptr;                       // [max_batch] which signals the repair's launches work on
    int32_t *d_valid_own = nullptr;                  // [max_batch] (the repair needs the words even if the caller gave none)
    bool last_device_repair = false;                 // the last call carried its repair: the summary has nothing to repeat
    int64_t device_repairs = 0;                      // signals the device-side repair has re-run (counted when a summary is read)
    int32_t fuse_range = 0;                          // tiles per knot-side workgroup: 0 = automatic, or 16 / 32 / 64 (itd_set_fuse_range)
    int32_t kf_shrink = 0;                           // automatic: how often a list has outgrown a workgroup (each time halves the range, down to 16 tiles)
    int64_t kf_resident_wgs = 0;                     // knot-side workgroups the device holds at once (occupancy query at creation of the workspace)
    int32_t fuse_group = 1;                          // chunks of a batch that share one knot side of the fused levels (1 since the knot side is one launch:
                                                     // sharing it over 2 / 4 chunks measured 25.1 / 26.1 ms against 23.9 on 1024 x 2^20)
    int64_t fuse_min_samples = (int64_t)2 << 20;   // automatic mode: samples per launch sequence from which the fused form pays
    int64_t fuse_signal_repairs = 0;   // signals itd_get_summary has re-run on their own (a few of a batch refused the fused form)
    bool last_kf = false;
    int last_kf_level = 0;         // the first fused level of that call
    int last_kf_form = 0;          // ... as enqueued (itd_get_last_fuse_level): stays when a summary has drawn the verdict (last_kf = false then)
    // fault injection into the fused levels' workspace (itd_debug_kf_fault; tests only): kind < 0 = none
    int32_t fault_kind = -1, fault_level = 0, fault_where = 0, fault_slot = 0, fault_delta = 0;
    int32_t fault_sig = 0;         // the signal of the batch the fault lands in (itd_debug_kf_fault_signal)
    int32_t spline_solver = ITD_SPLINE_AUTO;   // FITPACK flavour: serial bit-level sweep or the parallel moment form (itd_set_spline_solver)
    int32_t l0_mode = ITD_LEVEL0_AUTO;   // how level 0 finds its knots (itd_set_level0_mode)
    int32_t l0_records_left = 0;   // automatic mode: decompositions still to run record-driven after a fused launch fell short
    int64_t ws_bytes = 0;
    // host-convenience staging (grow only)
    void *d_cub = nullptr; size_t cub_bytes = 0;          // cubic variant: per-signal jobs + K, bf, b (3 arrays of idx+2 doubles each);
                                                          // also staging of the NaN-input helper path and the instantaneous step
    int32_t *d_cub_e = nullptr; size_t cub_e_bytes = 0;   // cubic variant: the caller's knots narrowed to int32 (host form)
    int32_t *d_flag = nullptr;                            // [1] device-side argument check
    void *d_dw = nullptr; size_t dw_bytes = 0;            // batched knot detection (DetectWs): cubic batch, detect batch, streams
    void *d_bw = nullptr; size_t bw_bytes = 0;            // batched single-level tier-1 extraction (BatchWs)
    void *d_sp = nullptr; size_t sp_bytes = 0;            // spline flavour (batched): lists, counts, records, group sums, states,
                                                          // ordered knot lists, totals, fit arrays, metadata
    double *d_sp2 = nullptr; size_t sp2_bytes = 0;        // 2-D consumers: three planes of scratch
    char *d_wpe = nullptr; size_t wpe_bytes = 0;          // weighted permutation entropy: the segments' sums
    void *d_io_x = nullptr; size_t io_x_bytes = 0;
    void *d_iq_avg = nullptr; size_t iq_avg_bytes = 0;   // the I/Q form of the cubic operator: the components' mean series
    double *d_io_rows = nullptr; size_t io_rows_bytes = 0;
    double *d_io_bases = nullptr; size_t io_bases_bytes = 0;
    void *h_pin[2] = {nullptr, nullptr};   // host-form calls: pinned bounce buffers of the pipelined device -> host copy (copy_to_host)
    bool host_keep_bases = false;      // itd_set_host_keep_baselines: host-form calls leave their baselines in d_io_bases
    int64_t kept_n = 0; int32_t kept_nb = -1;   // what itd_get_last_baselines_host can still deliver (-1: nothing)
    // last run
    bool ran = false;
    int32_t last_batch = 0, last_m = 0;
    int64_t last_n = 0;
    hipStream_t last_stream = nullptr;
    const void *last_x = nullptr;      // the last decomposition's arguments: itd_get_summary repeats it record-driven when the
    bool last_x_f32 = false;           // fused level-0 launch fell short of a tile's halo knots
    int64_t last_x_stride = 0;
    double *last_rows = nullptr, *last_bases = nullptr;
    bool last_fused = false;
    bool last_resident = false;        // the last run was the one-workgroup form (k_resident)
    bool last_nan_input = false;       // the last run was the NaN-input repeat (k_nan_level0): its results follow the reference
    int32_t nan_input_mode = ITD_NAN_INPUT_FOLLOW;   // itd_set_nan_input_mode
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;   // event pairs: [2k] start, [2k+1] stop
    std::vector<int> ev_tag;      // what pair k brackets (ITD_TIME_*)
    std::vector<int> ev_from, ev_to;   // the pair's two events (normally 2k, 2k+1; a span borrows the events of the launches at its ends)
    int span_first = -1, span_last = -1;   // of the decomposition being enqueued: its first / last instrumented launch
    int n_timed = 0;              // pairs recorded since timing was (re)enabled
    bool timing_overflow = false;
    int timing_mode = 0;          // 0: every launch class; 1: the level-0 launch only (itd_get_step_periods)
    int timing_stride = 1;        // instrument every stride-th decomposition only (event records cost ~5 us each)
    int timing_seq = 0;
    bool timing_now = false;
    char err[512] = {0};
};

namespace {

// PYITD_POISON=1 (a debugging switch of the environment, read once): every workspace the library allocates is filled with 0xFF bytes
// (NaNs / -1) before its first use, so that a kernel that reads memory nobody wrote fails on every run instead of once in ten
// thousand — what tools/stream_fuzz.py and the suite are run under in the evidence session.  Off: allocations cost what they cost.
inline bool poison_on()
{
    static const bool on = [] { const char *v = getenv("PYITD_POISON"); return v && *v && *v != '0'; }();
    return on;
}
inline void poison(void *p, size_t bytes)
{
    if (p && bytes && poison_on()) { (void)hipMemset(p, 0xFF, bytes); (void)hipDeviceSynchronize(); }   // (a fill on the null stream is not ordered with the engines' non-blocking streams)
}

int fail_hip(itd_engine *e, hipError_t rc, const char *what)
{
    if (e) snprintf(e->err, sizeof(e->err), "%s: %s (%d)", what, hipGetErrorString(rc), (int)rc);
    return ITD_ERR_HIP;
}

#define HIP_TRY(e, call)                                         \
    do {                                                         \
        hipError_t rc__ = (call);                                \
        if (rc__ != hipSuccess) return fail_hip((e), rc__, #call); \
    } while (0)

struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline int64_t tiles_of(int64_t n) { return (n + T - 1) / T; }

// hipEvent pairs on the launch stream around selected launches (bench instrumentation, off by default)
// a pair of events for a launch that records them itself (hipExtLaunchKernel)
int time_slot(itd_engine *e, int tag)
{
    if (!e->timing || !e->timing_now) return -1;
    if (e->timing_mode == 1 && tag != ITD_TIME_EXTRACT_L0) return -1;   // step periods: only the level-0 launch carries events
    if (2 * (size_t)e->n_timed + 1 >= e->ev.size()) { e->timing_overflow = true; return -1; }
    const int k = e->n_timed++;
    e->ev_tag[(size_t)k] = tag;
    e->ev_from[(size_t)k] = 2 * k; e->ev_to[(size_t)k] = 2 * k + 1;
    if (tag != ITD_TIME_DECOMPOSE && tag != ITD_TIME_KF_KNOTS) {      // a launch of its own: the ends of the decomposition's span
        if (e->span_first < 0) e->span_first = k;
        e->span_last = k;
    }
    return k;
}

// signals per launch sequence of a batched decomposition.  All levels of a chunk run before the next chunk starts, so
// the baseline a level writes (8 B per sample per signal) is still in the 256 MiB Infinity Cache when the next level reads it
// — the state the single 2^24-sample signal is in.  Automatic: about 2^24 samples per chunk.
constexpr int32_t kMaxGridY = 65535;   // HIP's limit on gridDim.y
int chunk_of(const itd_engine *e, int64_t n, int32_t batch)
{
    if (e->chunk > 0) return std::min<int32_t>(std::min<int32_t>(e->chunk, kMaxGridY), batch);
    // one stream: 2^24 samples per chunk; two or more (the default): half of that per chunk — two chunks in flight, measured best
    // on 512 x 2^20 with the fused levels' knot side as one launch (chunks of 8 signals over 2 streams: 12.3 ms against 12.7 with 12
    // and 13.0 with 16 over one stream; round 2, level by level: 10-12 signals over 2 streams, profiles/r02/session2_batch_streams.txt)
    // a batch of up to 2^24 samples in all is ONE sequence (64 signals of 2^18 samples: 393 us as one sequence against 466 as two chunks
    // of 32 over two streams; 256 x 2^16: 416 against 436 — profiles/r05/bench_default_form.json, many_mid_size_signals)
    if ((int64_t)batch * n <= ((int64_t)1 << 24) && batch <= kMaxGridY) return batch;
    const int64_t per = e->batch_streams > 1 ? ((int64_t)1 << 23) : ((int64_t)1 << 24);
    const int64_t c = std::max<int64_t>(1, per / n);
    return (int)std::min<int64_t>(std::min<int64_t>(c, kMaxGridY), batch);   // a chunk's signals are the launches' grid.y
}

// The workspace of the fused sparse levels (itd_knotfirst.hpp), allocated at the first call that takes that path.  Per signal and
// knot-side workgroup (kKcTiles tiles): a slab of table entries (32 B per knot and level: kKcSlab of them) and one 256-byte
// boundary record per level; per level and tile the knots' flag words and the tile's first table index; per tile the near-tie flag words.
constexpr int kKfLevels = ITD_MAX_ITERATION + 3;
// tiles per knot-side workgroup: its LDS holds kKcCapH candidates at the hand-over level — the first fused level's knots of its tiles
// (typically 0.036 / 0.012 n at levels 2 / 3: 18 / 6 per tile, 1180 / 400 per 64-tile range) plus the sticky ones — and kKcCap from
// the level after it; a call that a list outgrew halves the ranges of the calls after it
inline int kf_tiles_per_wg(const itd_engine *e, int first_fused_level)
{
    (void)first_fused_level;
    if (e->fuse_range) return e->fuse_range;
    return std::max(kKcTiles / 4, kKcTiles >> e->kf_shrink);
}
// The first fused level.  Automatic: level 2 where a launch sequence covers at least 2^22 samples — one level launch less (-63 us
// at 2^24) for a sample pass that writes one more row (+22) and a knot side that starts on a 2.6 x longer list (+14): one signal of
// 2^22 / 2^23 / 2^24 samples 153 / 225 / 377 us against 160 / 243 / 404 from level 3, but 102 against 98 at 2^20
// (profiles/r05/fuse_level_probe.txt) — unless a level-2 list has outgrown its workgroup on this engine (denser knots than the
// hand-over layout's 1720 per 64 tiles): level 3 then, before the ranges are halved.
inline int kf_first_level(const itd_engine *e, int64_t n, int32_t batch)
{
    if (e->fuse_level) return e->fuse_level;
    const int64_t seq = (int64_t)std::min<int32_t>(chunk_of(e, n, batch), batch) * n;
    return (seq >= ((int64_t)1 << 22) && !e->fuse_level2_off) ? 2 : 3;
}
// What the engine's NEXT calls do after fused levels refused with the failure bits `bits` (host-side repeat and device-side repair
// alike).  A list that outgrew its workgroup: the automatic first fused level goes from 2 to 3, after that the ranges are halved
// (down to 16 tiles) — the calls stay fused: returns true.  A halo wait given up with nothing else wrong (the grid was not resident
// at once after all: other work on the device): tickets from now on.  Anything else (verification, non-finite knot data): false —
// the caller lets the next calls run level by level for a while.
bool kf_back_off(itd_engine *e, int bits, int level)
{
    if (bits & kKfFailCapacity) {
        if (!e->fuse_level && level == 2 && !e->fuse_level2_off) { e->fuse_level2_off = true; return true; }
        if (!e->fuse_range && kf_tiles_per_wg(e, level) > kKcTiles / 4) { ++e->kf_shrink; return true; }
        return false;
    }
    if (bits == kKfFailWait && !e->kf_force_tickets) { e->kf_force_tickets = true; return true; }
    return false;
}
// the engine's next calls run level by level: 16 of them after a first refusal, twice as many after every refusal of the probing
// call that follows a back-off (up to 1024)
void kf_levels_off(itd_engine *e)
{
    if (e->fuse_probe) e->fuse_off_span = std::min(e->fuse_off_span * 2, 1024);
    e->fuse_off_left = e->fuse_off_span;
    e->fuse_probe = false;
}
int ensure_kf_ws(itd_engine *e, int tpw, bool may_allocate)
{
    const size_t wgs = (size_t)(e->max_tiles + tpw - 1) / tpw;
    if (e->d_kf && (size_t)e->kf.wgs_max >= wgs) return ITD_OK;
    if (!may_allocate) return ITD_ERR_NOMEM;
    if (e->d_kf) {
        // smaller ranges (a list outgrew its workgroup, itd_set_fuse_range, a lower first fused level): more workgroups per signal need
        // a larger workspace.  The old one is NOT freed: a hipGraph captured on this engine has its pointers baked into the fused
        // levels' launches and may be replayed at any time (its launches stay self-consistent: geometry and pointers travel
        // together as kernel arguments) — it is retired until itd_engine_destroy; its bytes stay counted
        e->kf_retired.push_back(e->d_kf);
        e->d_kf = nullptr;
    }
    const size_t B = (size_t)e->max_batch;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_sig = al(B * sizeof(KfSig)), b_pool = al(B * wgs * kKcSlab * sizeof(KfEntry));
    const size_t b_first = al(B * kKfLevels * (size_t)e->max_tiles * 4), b_tf = al(B * kKfLevels * (size_t)e->max_tiles * 64);
    const size_t b_tie = al(B * (size_t)e->max_tiles * 64), b_rec = al(B * kKfLevels * wgs * kKcRecGran * 8);
    const size_t total = b_sig + b_pool + b_first + b_tf + b_tie + b_rec;
    const hipError_t rc = hipMalloc(&e->d_kf, total);
    if (rc != hipSuccess) { e->d_kf = nullptr; fail_hip(e, rc, "hipMalloc(fused levels' workspace)"); return rc == hipErrorOutOfMemory ? ITD_ERR_NOMEM : ITD_ERR_HIP; }
    poison(e->d_kf, total);
    e->kf_bytes = total;
    e->ws_bytes += (int64_t)total;
    char *p = (char *)e->d_kf;
    KfWs &w = e->kf;
    w.sig = (KfSig *)p; p += b_sig;
    w.pool = (KfEntry *)p; p += b_pool;
    w.first = (int32_t *)p; p += b_first;
    w.tflags = (unsigned long long *)p; p += b_tf;
    w.nearw = (unsigned long long *)p; p += b_tie;
    w.rec = (unsigned long long *)p;
    // the signals' generation counters start at 0 and no record carries a tag yet
    // (hipMemset runs on the null stream; the launches that use the workspace on non-blocking streams: the fills have to be over first)
    if (hipMemset(w.sig, 0, b_sig) != hipSuccess || hipMemset(w.rec, 0, b_rec) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return ITD_ERR_HIP;
    w.wgs_max = (int32_t)wgs; w.rec_levels = kKfLevels;
    {   // how many knot-side workgroups are resident at once: a grid within that takes its ids from blockIdx (itd_knotfirst.hpp)
        int per_cu = 0, cus = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_kf_knots<T>), kKcThreads, 0) != hipSuccess ||
            hipGetDeviceProperties(&prop, e->device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        else cus = prop.multiProcessorCount;
        // (Should the hardware admit fewer than the API says, the surplus workgroups
        //  start in blockIdx order as others finish — observed, not promised; a wait that can never end is given up after
        //  ITD_KC_TIMEOUT and the call repeated level by level)
        e->kf_resident_wgs = (int64_t)per_cu * cus;
    }
    return ITD_OK;
}

template <typename Tin>
int enqueue_decompose(itd_engine *e, const Tin *x, int64_t n, int32_t batch, int64_t x_stride, int32_t M,
                      double *rows, double *bases_user, hipStream_t st, bool fuse0, bool nan_input = false, bool kf = false,
                      const int32_t *repair_need = nullptr)
{
    // repair_need: the device-side repair of the call just enqueued (itd_set_device_repair): the same call again, level by level,
    // whose launches return at once for every signal that is not flagged (SigState::skip, set by k_repair_init)
    // nan_input: the caller's signal holds a NaN (found by the previous, rejected run of this call): level 0 the way the
    // reference runs it — k_nan_level0 writes the mutated signal (NaN -> +inf, ITD.py:50) into the third baseline slot, which
    // nothing touches before level 2, and the level-0 records; the record-driven level-0 extraction then reads that copy
    if (nan_input) fuse0 = false;
    // kf: levels L0 .. max_iteration + 1 run fused (itd_knotfirst.hpp): one launch per level only for levels 0 .. L0 - 1
    int L0 = kf_first_level(e, n, batch);
    if (!e->fuse_level && L0 > M) L0 = 2;               // (automatic: few levels asked for)
    kf = kf && fuse0 && L0 >= 2 && L0 <= M && n < ((int64_t)1 << 31) - 65536;
    // a call that is being captured into a graph must be complete in itself (the graph may be replayed any number of times) and
    // cannot allocate: a captured call on an engine whose fused workspace does not exist yet runs level by level
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap_status);
    const bool capturing = cap_status != hipStreamCaptureStatusNone;
    const int kf_tpw = kf_tiles_per_wg(e, L0);
    if (kf && capturing && ensure_kf_ws(e, kf_tpw, false) != ITD_OK) kf = false;
    if (kf) {
        const int rc = ensure_kf_ws(e, kf_tpw, true);
        if (rc == ITD_ERR_NOMEM && e->fuse_mode != ITD_FUSE_ONLY) {
            // no room for the fused levels' workspace (136 B x max_n / 8 + 136 B per tile and level, per signal): this engine stays
            // level by level — the result is the same
            (void)hipGetLastError();
            e->fuse_no_memory = true;
            kf = false;
        } else if (rc) return rc;
    }
    // capped fused levels: levels L0 .. cap - 1 fused, cap .. M + 1 one launch each behind a scan of the baseline the sample pass leaves
    int cap = 0;
    if (kf) {
        cap = e->fuse_cap ? e->fuse_cap : e->fuse_cap_auto;                          // (fuse_cap -1 = never: falls out below)
        if (cap > 0 && !e->fuse_cap && e->fuse_cap_calls >= e->fuse_cap_span) cap = 0;   // (a learned cap is probed now and then: workloads change)
        if (cap < L0 + 2 || cap > M + 1) cap = 0;                                    // (fewer than two fused levels are not worth a knot side; beyond the call's levels: no cap)
    }
    const int Mk = cap ? cap - 2 : M;               // the knot side's "max_iteration": its levels are L0 .. Mk + 1
    const int n_tiles = (int)tiles_of(n);
    const int n_groups = groups_of(n_tiles);
    const int64_t R = (int64_t)M + 2;
    const int64_t rows_stride = R * (n + ITD_ROW_PAD);
    const dim3 blk(kWave);

    // instrument every timing_stride-th decomposition only: a launch that carries events needs a completion signal of its own
    // (~2 us per launch, measured), the whole-decomposition span two marker records (~5 us each)
    e->timing_now = e->timing && (e->timing_seq++ % e->timing_stride == 0);
    // the whole-decomposition span: from the first instrumented launch's begin to the last one's end — their dispatches' own
    // timestamps, no marker packets in the stream (two markers cost ~10 us of the instrumented step; k_finalize, not instrumented,
    // lies outside the span of a level-by-level call)
    const int span_pair = time_slot(e, ITD_TIME_DECOMPOSE);
    e->span_first = e->span_last = -1;
    // the set of states / group sums this call works on: the one the previous call did not use (the NaN-input repeat: the same
    // again, its states carry the in_nan flags), initialised by that call's k_finalize unless the bookkeeping says otherwise
    const int set = nan_input ? e->cur_set : (e->cur_set ^ 1);
    SigState *const set_state = e->d_state + (size_t)set * e->max_batch;
    int32_t *const set_gsum = e->d_gsum + (size_t)set * 3 * e->gsum_third;
    SigState *const other_state = e->d_state + (size_t)(set ^ 1) * e->max_batch;
    int32_t *const other_gsum = e->d_gsum + (size_t)(set ^ 1) * 3 * e->gsum_third;
    const int64_t gs_extent = (int64_t)batch * n_groups * kGsumPitch;
    if (repair_need) {
        const int64_t ge = 3 * e->gsum_third;
        const int gb = (int)std::min<int64_t>(std::max<int64_t>((ge + 255) / 256, (batch + 255) / 256), 2048);
        k_repair_init<<<gb, 256, 0, st>>>(other_state, set_state, repair_need, batch, set_gsum, ge);
        e->dirty_sig[set] = std::max(e->dirty_sig[set], batch);
    } else if (nan_input || capturing || e->dirty_sig[set] > 0 || e->dirty_gs[set] > 0) {
        const int64_t ge = 3 * e->gsum_third;   // the buffers are small: clear all of them
        const int gb = (int)std::min<int64_t>(std::max<int64_t>((ge + 255) / 256, (batch + 255) / 256), 2048);
        k_init_state<<<gb, 256, 0, st>>>(set_state, batch, set_gsum, ge, nan_input ? 1 : 0);
        e->dirty_sig[set] = std::max(e->dirty_sig[set], batch);
    } else {
        e->dirty_sig[set] = batch;
    }
    e->dirty_gs[set] = gs_extent;
    e->cur_set = set;
    if (bases_user)  // the reference's timeout result keeps an all-zero last baselines row (ITD.py:385,424)
        HIP_TRY(e, hipMemset2DAsync(bases_user + (R - 1) * n, (size_t)rows_stride * sizeof(double), 0,
                                    (size_t)n * sizeof(double), (size_t)batch, st));
    const int chunk = chunk_of(e, n, batch);
    // chunks are independent (per-signal state, counts, records, group sums, slots): with two streams they alternate, so that one
    // chunk's launch boundaries and tails overlap the other's work (fork after the init, join before the caller's stream goes on)
    const int n_chunks = (batch + chunk - 1) / chunk;
    // streams in use: the caller's and S - 1 of the engine's; signals too long for two of them to share the Infinity Cache
    // (more than 3 * 2^22 samples each) keep to one stream
    // chunks that share one knot side of the fused levels (grid.y of its launches: at most 65535 signals)
    const int group = (kf && n_chunks > 1) ? std::max(1, std::min<int>(e->fuse_group, kMaxGridY / chunk)) : 1;
    const int n_seqs = (n_chunks + group - 1) / group;
    // The batch pipeline (itd_set_batch_pipeline): with fused levels the chunks do not rotate over the streams — the caller's stream runs
    // every chunk's level launches and knot side in order, aux_stream[0] the sample passes, each beside the NEXT chunk's knot side
    // (below).  Not while capturing: a graph replays by its dependencies, a gate would poll in vain.
    const bool pipelined = kf && !capturing && e->batch_pipeline && e->batch_streams > 1 && n_chunks > 1 && group == 1;
    const int S = pipelined ? 1 : ((e->chunk == 0 && (int64_t)chunk * n > ((int64_t)3 << 22)) ? 1 : std::min<int>(e->batch_streams, n_seqs));
    if (pipelined) {
        if (!e->aux_stream[0]) {
            HIP_TRY(e, hipStreamCreateWithFlags(&e->aux_stream[0], hipStreamNonBlocking));
            HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join[0], hipEventDisableTiming));
        }
        if (!e->d_kf_started) {
            HIP_TRY(e, hipMalloc(&e->d_kf_started, 16));
            HIP_TRY(e, hipMemset(e->d_kf_started, 0, 16));
            HIP_TRY(e, hipDeviceSynchronize());          // (a fill on the null stream is not ordered with the engine's non-blocking streams)
            e->kf_started_target = 0;
        }
        while (e->ev_pipe.size() < 1) {
            hipEvent_t ev = nullptr;
            HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->ev_pipe.push_back(ev);
        }
    }
    if (S > 1) {
        if (!e->ev_fork) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIP_TRY(e, hipEventRecord(e->ev_fork, st));
        for (int k = 0; k < S - 1; ++k) {
            if (!e->aux_stream[k]) {
                HIP_TRY(e, hipStreamCreateWithFlags(&e->aux_stream[k], hipStreamNonBlocking));
                HIP_TRY(e, hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming));
            }
            HIP_TRY(e, hipStreamWaitEvent(e->aux_stream[k], e->ev_fork, 0));
        }
    }
    // One chunk's launches, in three phases: 1 = the level launches + k_finalize, 2 = the fused levels' knot side, 4 = their sample
    // pass.  Without fused levels a chunk is phase 1 alone.  With them, `group` consecutive chunks share ONE knot side: its dozen
    // launches are bound by their boundaries (6-9 us each whatever the list lengths) and cannot hide behind another stream's
    // memory-bound launches (measured: a short launch that meets a long one from the other stream ends when the long one ends, on a
    // stream of the highest priority as well), so they are amortised over more signals instead — while the level launches keep the
    // chunk size at which a level's baseline is still in the Infinity Cache for the next level.
    auto run_chunk = [&](const int b0, const int nb, const hipStream_t cst, const int phase) -> int {
        // signals b0 .. b0+nb-1: grid.y, every per-signal pointer offset by b0
        auto gs = [&](int level) { return set_gsum + (int64_t)(level % 3) * e->gsum_third + (int64_t)b0 * n_groups * kGsumPitch; };
        auto cnt = [&](int level) { return e->d_counts + (int64_t)(level & 1) * e->tiles_half + (int64_t)b0 * n_tiles; };
        auto rec = [&](int level) { return e->d_recs + (int64_t)(level & 1) * e->tiles_half + (int64_t)b0 * n_tiles; };
        SigState *state = set_state + b0;
        const Tin *xc = x + (int64_t)b0 * x_stride;
        double *rows_c = rows + (int64_t)b0 * rows_stride;
        double *bases_c = bases_user ? bases_user + (int64_t)b0 * rows_stride : nullptr;
        double *pp_c = e->d_pp + (int64_t)b0 * 3 * e->pp_pitch;
        double *xm_c = pp_c + 2 * e->pp_pitch;   // NaN-input repeat: the mutated signal, one per signal at the slots' stride
        const int j_last = kf ? L0 - 1 : M + 1;
        unsigned long long *near_c = kf ? e->kf.nearw + (size_t)b0 * n_tiles * 8 : nullptr;
        // the level launches ja .. jb of this chunk (one k_extract each)
        auto run_levels = [&](const int ja, const int jb) -> int {
        for (int j = ja; j <= jb; ++j) {
            // extraction j+1: input = level-j signal, rotation -> rows[j], baseline -> bases[j]
            double *base_out;
            int64_t base_stride;
            const double *base_in = nullptr;
            int64_t base_in_stride = 0;
            if (bases_c) {
                base_out = bases_c + (int64_t)j * n;
                base_stride = rows_stride;
                if (j >= 1) { base_in = bases_c + (int64_t)(j - 1) * n; base_in_stride = rows_stride; }
            } else {
                base_out = pp_c + (int64_t)(j % 3) * e->pp_pitch;
                base_stride = 3 * e->pp_pitch;
                if (j >= 1) { base_in = pp_c + (int64_t)((j - 1) % 3) * e->pp_pitch; base_in_stride = 3 * e->pp_pitch; }
            }
            double *rot_out = rows_c + (int64_t)j * (n + ITD_ROW_PAD);
            const bool final_level = j == M + 1;        // (with fused levels: only behind capped ones)
            const int pair = time_slot(e, final_level ? ITD_TIME_EXTRACT_FINAL : (j == 0 ? ITD_TIME_EXTRACT_L0 : ITD_TIME_EXTRACT));
            // launched through hipExtLaunchKernel: when this step is instrumented the two events take the dispatch's own
            // begin / end timestamps (no marker packets in the stream: nothing is added to the timed region)
#define ITD_LAUNCH_EXTRACT(TIN, FIN, XIN, XSTRIDE, CAPK, FUSE) ITD_LAUNCH_EXTRACT_KT(TIN, FIN, XIN, XSTRIDE, CAPK, FUSE, kTilesPerWave, false)
#define ITD_LAUNCH_EXTRACT_KT(TIN, FIN, XIN, XSTRIDE, CAPK, FUSE, KTW, TIES)                                                \
    do {                                                                                                                   \
        const TIN *a_x = XIN; int64_t a_xs = XSTRIDE, a_n = n, a_rs = rows_stride, a_bs = base_stride;                     \
        int a_nt = n_tiles, a_b = nb, a_lvl = j, a_keep = 0;                                                               \
        const int32_t *a_ci = cnt(j), *a_gi = gs(j); int32_t *a_co = cnt(j + 1), *a_go = gs(j + 1), *a_gc = gs(j + 2);      \
        const TileRec *a_ri = rec(j); TileRec *a_ro = rec(j + 1); double *a_rot = rot_out, *a_bas = base_out;              \
        SigState *a_st = state; unsigned long long *a_tie = (TIES) ? near_c : nullptr;                                                 \
        void *args[] = {&a_x, &a_xs, &a_n, &a_nt, &a_b, &a_ci, &a_co, &a_ri, &a_ro, &a_gi, &a_go, &a_gc, &a_rot, &a_rs,    \
                        &a_bas, &a_bs, &a_st, &a_lvl, &a_keep, &a_tie};                                                    \
        HIP_TRY(e, hipExtLaunchKernel(reinterpret_cast<const void *>(&k_extract<TIN, T, FIN, CAPK, KTW, FUSE, TIES>),              \
                                      dim3((n_tiles + (KTW) - 1) / (KTW), nb),                                               \
                                      blk, args, 0, cst, pair >= 0 ? e->ev[2 * (size_t)pair] : nullptr,                    \
                                      pair >= 0 ? e->ev[2 * (size_t)pair + 1] : nullptr, 0));                              \
    } while (0)
            if (j == 0) {   // never the last level: M >= 0
                if (nan_input) ITD_LAUNCH_EXTRACT(double, false, xm_c, 3 * e->pp_pitch, kRankCap0, false);
                else if (fuse0) ITD_LAUNCH_EXTRACT_KT(Tin, false, xc, x_stride, kRankCap0, true, kFuse0TilesPerWave, false);
                else ITD_LAUNCH_EXTRACT(Tin, false, xc, x_stride, kRankCap0, false);
            } else {
                if (final_level) ITD_LAUNCH_EXTRACT(double, true, base_in, base_in_stride, kRankCap, false);
                // (the launch in front of the fused sparse levels also flags the tiles of its baseline that hold a near tie)
                else if (kf && j == j_last) ITD_LAUNCH_EXTRACT_KT(double, false, base_in, base_in_stride, kRankCap, false, kTilesPerWave, true);
                else ITD_LAUNCH_EXTRACT(double, false, base_in, base_in_stride, kRankCap, false);
            }
#undef ITD_LAUNCH_EXTRACT
#undef ITD_LAUNCH_EXTRACT_KT
        }
        return ITD_OK;
        };
        // stop test on the last pending baseline (ITD.py:400-404 takes priority over the timeout branch) and the residual row
        auto run_finalize = [&]() {
            // blocks per signal: a thread of the row fix-up moves 8 samples (four 16-byte accesses) before the grid is widened
            const int fb = (int)std::min<int64_t>(std::max<int64_t>((n + 8 * kFinalizeThreads - 1) / (8 * kFinalizeThreads), 1), 1024);
            int32_t *og = other_gsum + (int64_t)b0 * n_groups * kGsumPitch;
            const int jf = M + 2;      // the level whose input is pending
            if (bases_c)
                k_finalize<<<dim3(fb, nb), kFinalizeThreads, 0, cst>>>(rows_c, rows_stride, n, bases_c, rows_stride, n, 0,
                                                                       gs(jf), n_tiles, jf, state, other_state + b0, og, e->gsum_third);
            else
                k_finalize<<<dim3(fb, nb), kFinalizeThreads, 0, cst>>>(rows_c, rows_stride, n, pp_c, 3 * e->pp_pitch,
                                                                       e->pp_pitch, 3, gs(jf), n_tiles, jf, state, other_state + b0, og, e->gsum_third);
        };
        if (phase & 1) {
        if (nan_input) {
            k_nan_level0<Tin, T><<<dim3(n_tiles, nb), blk, 0, cst>>>(xc, x_stride, n, n_tiles, xm_c, 3 * e->pp_pitch, cnt(0), rec(0),
                                                                    gs(0), state);
        } else if (!fuse0) {
            const int pair = time_slot(e, ITD_TIME_SCAN0);
            const Tin *a_x = xc; int64_t a_xs = x_stride, a_n = n; int a_nt = n_tiles;
            int32_t *a_c = cnt(0), *a_g = gs(0); TileRec *a_r = rec(0); SigState *a_st = state; int a_lv = 0;
            void *args[] = {&a_x, &a_xs, &a_n, &a_nt, &a_c, &a_r, &a_g, &a_st, &a_lv};
            HIP_TRY(e, hipExtLaunchKernel(reinterpret_cast<const void *>(&k_scan0<Tin, T, kScanTilesPerWave>),
                                          dim3((n_tiles + kScanTilesPerWave - 1) / kScanTilesPerWave, nb), blk, args, 0, cst,
                                          pair >= 0 ? e->ev[2 * (size_t)pair] : nullptr,
                                          pair >= 0 ? e->ev[2 * (size_t)pair + 1] : nullptr, 0));
        }

        {
            const int rc = run_levels(0, j_last);
            if (rc) return rc;
        }
        // With fused sparse levels the knot side's launch does k_finalize's work for its first level (k_kf_knots: KfFin) — one launch less
        if (!kf) run_finalize();
        }
        if (kf && (phase & 6)) {
            // ---- levels L0 .. M + 1 fused: hand-over, the knot-side steps, ONE pass over the samples, the verdict ----
            KfWs w = e->kf;
            w.n_tiles = n_tiles; w.L0 = L0; w.nlev = Mk + 3 - L0;
            w.cap = cap;
            w.xnext = cap ? pp_c + (int64_t)((cap - 1) % 3) * e->pp_pitch : nullptr; w.xnext_stride = 3 * e->pp_pitch;
            w.tpw = kf_tpw; w.wgs = (n_tiles + kf_tpw - 1) / kf_tpw; w.nb = nb;
            // (ids from blockIdx only where the whole grid is resident at once — with S streams in flight each launch may count on
            //  its share of the device only)
            w.ticketed = ((int64_t)w.wgs * nb * S > e->kf_resident_wgs || e->kf_force_tickets) ? 1 : 0;
            w.dbg_kind = e->fault_kind; w.dbg_lev = e->fault_level; w.dbg_wg = e->fault_where; w.dbg_slot = e->fault_slot; w.dbg_delta = e->fault_delta;
            w.dbg_sig = e->fault_sig - b0;               // (relative to this launch's first signal; outside it: no workgroup matches)
            w.started = pipelined ? e->d_kf_started : nullptr;
            const size_t B0 = (size_t)b0;
            w.sig += B0; w.pool += B0 * (size_t)w.wgs_max * kKcSlab; w.rec += B0 * (size_t)w.rec_levels * w.wgs_max * kKcRecGran;
            w.first += B0 * (size_t)w.nlev * n_tiles; w.tflags += B0 * (size_t)w.nlev * n_tiles * 8; w.nearw += B0 * (size_t)n_tiles * 8;
            // (the per-signal strides of `first` / `tflags` follow this call's geometry: nlev levels x n_tiles tiles per signal)
            const double *xl = bases_c ? bases_c + (int64_t)(L0 - 1) * n : pp_c + (int64_t)((L0 - 1) % 3) * e->pp_pitch;
            const int64_t xl_stride = bases_c ? rows_stride : 3 * e->pp_pitch;
            const hipStream_t kst = cst;
            if (phase & 2) {
                // ONE launch: hand-over and every fused level (timed from its dispatch's own begin / end timestamps)
                const int p_kn = time_slot(e, ITD_TIME_KF_KNOTS);
                KfWs a_w = w; int64_t a_ls = xl_stride, a_n = n; const double *a_xl = xl; int a_m = Mk;
                const int32_t *a_c = cnt(L0); const TileRec *a_r = rec(L0); SigState *a_st = state;
                KfFin a_f;      // k_finalize's work for the first fused level (the stop test of its input, the other state set)
                a_f.rows = rows_c; a_f.rows_stride = rows_stride;
                a_f.bases = bases_c ? bases_c : pp_c; a_f.bases_stride = bases_c ? rows_stride : 3 * e->pp_pitch;
                a_f.bases_row_pitch = bases_c ? n : e->pp_pitch; a_f.bases_rotate = bases_c ? 0 : 3;
                a_f.gsum = gs(L0); a_f.other_state = other_state + b0; a_f.other_gsum = other_gsum + (int64_t)b0 * n_groups * kGsumPitch;
                a_f.other_third = e->gsum_third;
                void *args[] = {&a_w, &a_f, &a_xl, &a_ls, &a_n, &a_m, &a_c, &a_r, &a_st};
                HIP_TRY(e, hipExtLaunchKernel(reinterpret_cast<const void *>(&k_kf_knots<T>), dim3((unsigned)w.wgs * (unsigned)nb), dim3(kKcThreads), args, 0, kst,
                                              p_kn >= 0 ? e->ev[2 * (size_t)p_kn] : nullptr, p_kn >= 0 ? e->ev[2 * (size_t)p_kn + 1] : nullptr, 0));
                if (w.started) e->kf_started_target += (unsigned long long)w.wgs * (unsigned long long)nb;
            }
            if (phase & 4) {
#ifdef ITD_DEBUG_GAP
                k_debug_gap<<<1, 64, 0, cst>>>((long long)(ITD_DEBUG_GAP) * 100);
#endif
                if (e->fault_kind >= 0 && (e->fault_kind <= 5 || e->fault_kind == 8) && e->fault_level >= L0 && e->fault_level - L0 < w.nlev && e->fault_where >= 0 &&
                    e->fault_where < n_tiles && e->fault_sig >= b0 && e->fault_sig < b0 + nb)      // (tests only) one field of what the sample pass is about to read, perturbed
                    k_kf_fault<<<1, 64, 0, cst>>>(w, e->fault_sig - b0, e->fault_kind, e->fault_level - L0, e->fault_where, e->fault_slot & 0xffff, e->fault_delta);
                const int pair = time_slot(e, ITD_TIME_KF_APPLY);
                KfWs a_w = w; const double *a_xl = xl; int64_t a_xs = xl_stride, a_n = n, a_rs = rows_stride, a_bs = rows_stride;
                const TileRec *a_rec = rec(L0); double *a_rows = rows_c, *a_bases = bases_c;
                void *args[] = {&a_w, &a_xl, &a_xs, &a_n, &a_rec, &a_rows, &a_rs, &a_bases, &a_bs};
                const void *apply_fn = cap ? (bases_c ? reinterpret_cast<const void *>(&k_kf_apply<T, kKfCap, true, true>) : reinterpret_cast<const void *>(&k_kf_apply<T, kKfCap, false, true>))
                                           : (bases_c ? reinterpret_cast<const void *>(&k_kf_apply<T, kKfCap, true, false>) : reinterpret_cast<const void *>(&k_kf_apply<T, kKfCap, false, false>));
                HIP_TRY(e, hipExtLaunchKernel(apply_fn, dim3(n_tiles + kf_check_blocks(w.wgs), nb), dim3(kWave), args, 0, cst,
                                              pair >= 0 ? e->ev[2 * (size_t)pair] : nullptr, pair >= 0 ? e->ev[2 * (size_t)pair + 1] : nullptr, 0));
                if (cap) {
                    // ---- capped: levels cap .. M + 1 one launch each, from a scan of the baseline the sample pass has stored (the
                    //      input of level cap: its knots' records, counts and group sums, the level's end samples).  A signal that
                    //      stopped inside the fused levels carries SigState::skip: these launches return at once for it.  Should the
                    //      fused levels refuse, the whole call is repeated anyway: what runs here then is discarded.
                    const int64_t ge = (int64_t)nb * n_groups * kGsumPitch;
                    k_clear_gsum<<<(int)std::min<int64_t>((ge + 255) / 256, 1024), 256, 0, cst>>>(gs(cap), gs(cap + 1), ge);
                    const double *a_x = bases_c ? bases_c + (int64_t)(cap - 1) * n : pp_c + (int64_t)((cap - 1) % 3) * e->pp_pitch;
                    k_scan0<double, T, kScanTilesPerWave><<<dim3((n_tiles + kScanTilesPerWave - 1) / kScanTilesPerWave, nb), blk, 0, cst>>>(
                        a_x, bases_c ? rows_stride : 3 * e->pp_pitch, n, n_tiles, cnt(cap), rec(cap), gs(cap), state, cap);
                    const int rc = run_levels(cap, M + 1);
                    if (rc) return rc;
                    run_finalize();
                }
            }
        }
        return ITD_OK;
    };
    if (pipelined) {
        // caller's stream X:   A_0 K_0 | A_1 K_1 | A_2 K_2 | ...                      A_k: chunk k's level launches, K_k: its knot side
        // aux_stream[0]   Y:        g_1 P_0 | g_2 P_1 | ...  | P_last                P_k: its sample pass, g_k: a gate (k_kf_gate)
        // A knot side needs room when it STARTS: 65 KB of LDS and eight wave slots per workgroup, which a device full of one-wavefront
        // workgroups never has (profiles/r04/experiments/README.md) — but once its workgroups are resident it needs no bandwidth and
        // runs hidden beside a memory-bound launch.  So K_k follows A_k in stream order (started ~2 us after A_k's last wavefront has
        // gone), and the OTHER stream's next sample pass waits behind a gate — one wavefront that returns when every workgroup of
        // K_k has started (KfWs::started) — instead of filling the device the moment A_k drains.  P_{k-1} then runs beside K_k, its
        // tail beside A_{k+1}.  The gate also orders P_{k-1} behind K_{k-1}, which ended before A_k began.  No event crosses the
        // queues on the way (an event costs ~13 us of idle device, measured), except at the call's two ends.
        const hipStream_t q = e->aux_stream[0];
        if (!e->ev_fork) HIP_TRY(e, hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        HIP_TRY(e, hipEventRecord(e->ev_fork, st));
        HIP_TRY(e, hipStreamWaitEvent(q, e->ev_fork, 0));
        for (int k = 0; k < n_chunks; ++k) {
            const int b0 = k * chunk, nb = std::min(chunk, batch - b0);
            int rc = run_chunk(b0, nb, st, 3);
            if (rc) return rc;
            if (k >= 1) {
                k_kf_gate<<<1, kWave, 0, q>>>(e->d_kf_started, e->kf_started_target, e->pipe_gate_timeout);      // (a gate that gives up says so: itd_get_summary)
                rc = run_chunk(b0 - chunk, chunk, q, 4);
                if (rc) return rc;
            }
        }
        HIP_TRY(e, hipEventRecord(e->ev_pipe[0], st));
        HIP_TRY(e, hipStreamWaitEvent(q, e->ev_pipe[0], 0));
        const int bl = (n_chunks - 1) * chunk;
        const int rc = run_chunk(bl, batch - bl, q, 4);
        if (rc) return rc;
        HIP_TRY(e, hipEventRecord(e->ev_join[0], q));
        HIP_TRY(e, hipStreamWaitEvent(st, e->ev_join[0], 0));
    }
    int seq_no = 0;
    for (int s0 = pipelined ? batch : 0; s0 < batch; s0 += chunk * group, ++seq_no) {
        const int lane_s = seq_no % S;
        const hipStream_t cst = lane_s == 0 ? st : e->aux_stream[lane_s - 1];   // this launch sequence's stream
        const int s1 = std::min(batch, s0 + chunk * group);
        int rc = ITD_OK;
        if (group == 1) rc = run_chunk(s0, s1 - s0, cst, 7);
        else {
            for (int b0 = s0; b0 < s1 && !rc; b0 += chunk) rc = run_chunk(b0, std::min(chunk, s1 - b0), cst, 1);
            if (!rc) rc = run_chunk(s0, s1 - s0, cst, 2);
            for (int b0 = s0; b0 < s1 && !rc; b0 += chunk) rc = run_chunk(b0, std::min(chunk, s1 - b0), cst, 4);
        }
        if (rc) return rc;
    }
    for (int k = 0; k < S - 1; ++k) {
        HIP_TRY(e, hipEventRecord(e->ev_join[k], e->aux_stream[k]));
        HIP_TRY(e, hipStreamWaitEvent(st, e->ev_join[k], 0));
    }
    // what this call's k_finalize launches leave initialised in the other set
    if (!capturing && batch >= e->dirty_sig[set ^ 1] && gs_extent >= e->dirty_gs[set ^ 1]) {
        e->dirty_sig[set ^ 1] = 0;
        e->dirty_gs[set ^ 1] = 0;
    }
    if (span_pair >= 0) {
        if (e->span_first >= 0) { e->ev_from[(size_t)span_pair] = 2 * e->span_first; e->ev_to[(size_t)span_pair] = 2 * e->span_last + 1; }
        else e->ev_tag[(size_t)span_pair] = -1;           // nothing instrumented in this call
    }
    HIP_TRY(e, hipGetLastError());
    e->ran = true;
    e->last_batch = batch;
    e->last_m = M;
    e->last_n = n;
    e->last_stream = st;
    e->last_x = x;
    e->last_x_f32 = sizeof(Tin) == 4;
    e->last_x_stride = x_stride;
    e->last_rows = rows;
    e->last_bases = bases_user;
    e->last_fused = fuse0;
    e->last_resident = false;
    e->last_nan_input = nan_input;
    e->last_kf = kf;
    e->last_kf_level = L0;
    e->last_kf_cap = kf ? cap : 0;
    e->last_pipelined = pipelined;
    if (pipelined) e->pipe_word_unread = true;
    if (!repair_need) { e->last_kf_form = kf ? L0 : 0; e->last_kf_cap_form = kf ? cap : 0; }
    return ITD_OK;
}

// Short signals (n <= kResidentMax): the whole decomposition as ONE launch, one workgroup per signal, the signal resident
// in LDS (itd_resident.hpp).  Optimistic: the kernel handles finite data only and raises SigState::res_fail
// otherwise; itd_get_summary then repeats the call level by level.  The kernel initialises the states it works on itself
// and leaves the other set's states as k_finalize would (the group sums are not touched).
bool want_fused(itd_engine *e);
bool want_resident(itd_engine *e, int64_t n)
{
    if (n > kResidentMax || e->resident_mode == ITD_RESIDENT_OFF) return false;
    if (e->resident_mode == ITD_RESIDENT_ONLY) return true;
    // an engine that was told how to run its level 0 / its launches, or that is being timed launch by launch, means the
    // level-by-level form
    if (e->l0_mode != ITD_LEVEL0_AUTO || e->timing) return false;
    if (e->resident_off_left > 0) { --e->resident_off_left; return false; }
    return true;
}

template <typename Tin>
int enqueue_resident(itd_engine *e, const Tin *x, int64_t n, int32_t batch, int64_t x_stride, int32_t M, double *rows,
                     double *bases_user, hipStream_t st)
{
    const int64_t R = (int64_t)M + 2;
    const int64_t rows_stride = R * n;
    const int set = e->cur_set ^ 1;
    SigState *const set_state = e->d_state + (size_t)set * e->max_batch;
    SigState *const other_state = e->d_state + (size_t)(set ^ 1) * e->max_batch;
    if (bases_user)  // the reference's timeout result keeps an all-zero last baselines row (ITD.py:385,424)
        HIP_TRY(e, hipMemset2DAsync(bases_user + (R - 1) * n, (size_t)rows_stride * sizeof(double), 0,
                                    (size_t)n * sizeof(double), (size_t)batch, st));
    // the window of by-rank knot slots: automatic (0.4 knots per sample) or what the engine was told, cut to what fits the LDS
    int cw = e->resident_window > 0 ? e->resident_window : resident_auto_window((int)n);
    cw = std::min(cw, resident_pad((int)n));
    while (resident_lds_bytes((int)n, cw) > kResidentLdsMax) cw -= 64;
    const size_t lds = resident_lds_bytes((int)n, cw);
    // four samples per thread: 64 threads up to 256 samples ... 1024 threads up to 4096, eight per thread up to 8192 (the
    // per-thread sample loops are unrolled over registers; the wider workgroups hide the phases' latencies better)
    const int cls = n <= 256 ? 0 : n <= 512 ? 1 : n <= 1024 ? 2 : n <= 2048 ? 3 : n <= 4096 ? 4 : 5;
    const int inst = cls + (sizeof(Tin) == 4 ? 0 : 6);
    const int threads = cls == 5 ? 1024 : 64 << cls;
    const void *fn = cls == 0 ? reinterpret_cast<const void *>(&k_resident<Tin, 64, 4>)
                   : cls == 1 ? reinterpret_cast<const void *>(&k_resident<Tin, 128, 4>)
                   : cls == 2 ? reinterpret_cast<const void *>(&k_resident<Tin, 256, 4>)
                   : cls == 3 ? reinterpret_cast<const void *>(&k_resident<Tin, 512, 4>)
                   : cls == 4 ? reinterpret_cast<const void *>(&k_resident<Tin, 1024, 4>)
                              : reinterpret_cast<const void *>(&k_resident<Tin, 1024, 8>);
    if (!e->resident_attr[inst]) {   // more than 64 KB of dynamic LDS has to be asked for
        const hipError_t arc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResidentLdsMax);
        if (arc != hipSuccess) {     // a device / runtime that does not grant it: this engine runs level by level from now on
            (void)hipGetLastError();
            if (e->resident_mode == ITD_RESIDENT_ONLY) return fail_hip(e, arc, "hipFuncSetAttribute(k_resident, MaxDynamicSharedMemorySize)");
            e->resident_mode = ITD_RESIDENT_OFF;
            return enqueue_decompose<Tin>(e, x, n, batch, x_stride, M, rows, bases_user, st, want_fused(e));
        }
        e->resident_attr[inst] = true;
    }
    const Tin *a_x = x; int64_t a_xs = x_stride, a_rs = rows_stride, a_bs = rows_stride;
    int a_n = (int)n, a_m = M, a_cw = cw, a_nf = e->nan_input_mode == ITD_NAN_INPUT_FOLLOW ? 1 : 0;
    double *a_rows = rows, *a_bases = bases_user;
    SigState *a_st = set_state, *a_ot = other_state;
    void *args[] = {&a_x, &a_xs, &a_n, &a_m, &a_cw, &a_nf, &a_rows, &a_rs, &a_bases, &a_bs, &a_st, &a_ot};
    HIP_TRY(e, hipLaunchKernel(fn, dim3((unsigned)batch), dim3((unsigned)threads), args, lds, st));
    HIP_TRY(e, hipGetLastError());
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    e->dirty_sig[set] = std::max(e->dirty_sig[set], batch);
    if (cap == hipStreamCaptureStatusNone && batch >= e->dirty_sig[set ^ 1]) e->dirty_sig[set ^ 1] = 0;
    e->cur_set = set;
    e->ran = true;
    e->last_batch = batch;
    e->last_m = M;
    e->last_n = n;
    e->last_stream = st;
    e->last_x = x;
    e->last_x_f32 = sizeof(Tin) == 4;
    e->last_x_stride = x_stride;
    e->last_rows = rows;
    e->last_bases = bases_user;
    e->last_fused = false;
    e->last_resident = true;
    e->last_kf = false;
    e->last_kf_form = 0;
    e->last_nan_input = false;
    return ITD_OK;
}

// The sparse levels fused (itd_knotfirst.hpp)?  Automatic: when a launch sequence covers enough samples for its launches to be
// memory bound — the fused form trades traffic for launches (18 instead of 10 at 8 levels), and below ~2^22.5 samples per
// sequence every launch is bound by its ~6.5 us boundary: measured one signal of 2^16 / 2^20 / 2^22 / 2^23 samples 64 / 88 / 180 /
// 305 us level by level against 102 / 116 / 191 / 277 us fused (tools/fuse_threshold.py) —, the fused level 0 (it flags the input's
// exact ties), no per-launch timing; not while a recent call had to be repeated level by level (smooth or quantised input
// keeps failing the verification: workloads tend to be homogeneous).
bool want_kf(itd_engine *e, int64_t n, int32_t batch, int32_t M, bool fuse0)
{
    if (e->fuse_mode == ITD_FUSE_OFF || !fuse0 || M < 2 || (e->fuse_level && e->fuse_level > M)) return false;
    if (e->fuse_mode == ITD_FUSE_ONLY) return true;
    if (n < 65536 || e->l0_mode != ITD_LEVEL0_AUTO || e->fuse_no_memory) return false;
    if ((int64_t)std::min<int32_t>(chunk_of(e, n, batch), batch) * n < e->fuse_min_samples) return false;
    if (e->fuse_off_left > 0) { if (--e->fuse_off_left == 0) e->fuse_probe = true; return false; }
    return true;
}

template <typename Tin>
int enqueue_any(itd_engine *e, const Tin *x, int64_t n, int32_t batch, int64_t x_stride, int32_t M, double *rows,
                double *bases_user, hipStream_t st)
{
    int rc;
    if (want_resident(e, n)) rc = enqueue_resident<Tin>(e, x, n, batch, x_stride, M, rows, bases_user, st);
    else {
        const bool f0 = want_fused(e);
        rc = enqueue_decompose<Tin>(e, x, n, batch, x_stride, M, rows, bases_user, st, f0, false, want_kf(e, n, batch, M, f0));
    }
    e->last_device_repair = false;
    if (rc || (!e->valid_dev && !e->device_repair)) return rc;
    // ---- behind the call's last launch: the verdict on the device (and, if asked for, the repair) ----
    const int vb = (batch + 63) / 64;
    SigState *state_a = e->d_state + (size_t)e->cur_set * e->max_batch;
    int32_t *valid = e->valid_dev ? e->valid_dev : e->d_valid_own;
    const int follow = e->nan_input_mode == ITD_NAN_INPUT_FOLLOW ? 1 : 0;
    k_verdict<<<vb, 64, 0, st>>>(state_a, batch, e->last_kf ? e->kf.sig : nullptr, e->last_kf_level, follow, valid, e->device_repair ? e->d_need : nullptr,
                                 (e->last_kf && e->last_pipelined) ? e->d_kf_started + 1 : nullptr);
    if (e->device_repair) {
        // the same call level by level (record-driven level 0: any knot spacing), guarded per signal by d_need: rows_dev is final
        // when the stream has drained, with no host synchronisation in between
        rc = enqueue_decompose<Tin>(e, x, n, batch, x_stride, M, rows, bases_user, st, false, false, false, e->d_need);
        if (rc) return rc;
        k_verdict_repaired<<<vb, 64, 0, st>>>(e->d_state + (size_t)e->cur_set * e->max_batch, batch, e->d_need, follow, valid);
        e->last_device_repair = true;
    }
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

// how the next decomposition's level 0 finds its knots: fused (one pass over the signal) unless the engine was told
// otherwise or a recent fused launch fell short (smooth input: the following decompositions go record-driven directly)
bool want_fused(itd_engine *e)
{
    if (e->l0_mode == ITD_LEVEL0_RECORDS) return false;
    if (e->l0_mode == ITD_LEVEL0_FUSED) return true;
    if (e->l0_records_left > 0) { --e->l0_records_left; return false; }
    return true;
}

int check_args(itd_engine *e, const void *x, int64_t n, int32_t batch, int64_t x_stride, int32_t M, const void *rows)
{
    if (!e || !x || !rows) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n || n >= (int64_t)INT32_MAX) return ITD_ERR_INVALID_ARG;  // N < 3: ITD.py:42-43 is garbage
    if (batch < 1 || batch > e->max_batch) return ITD_ERR_INVALID_ARG;   // any size: batches run in chunks of <= 65535 signals
    if (batch > 1 && x_stride < n) return ITD_ERR_INVALID_ARG;
    if (M < 0 || M > ITD_MAX_ITERATION) return ITD_ERR_INVALID_ARG;  // row M+1 must fit in 22 rows (ITD.py:384,421)
    return ITD_OK;
}

template <typename Tp>
int grow(itd_engine *e, Tp **p, size_t *have, size_t want)
{
    if (*have >= want) return ITD_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    hipError_t rc = hipMalloc((void **)p, want);
    if (rc != hipSuccess) { fail_hip(e, rc, "hipMalloc(io)"); return ITD_ERR_NOMEM; }
    poison(*p, want);
    *have = want;
    return ITD_OK;
}

// Device -> pageable host memory for the host-form calls (numpy in, numpy out).  hipMemcpy into pageable memory stages through
// the runtime's own pinned buffer and one host thread moves the bytes on (~12 GB/s measured: the 1.2 GB of rows of a 2^24-sample
// decomposition took 100 ms for 0.56 ms of compute).  Here: DMA into one of two pinned bounce buffers while a few host threads copy
// the other one into the caller's array (and take its first-touch page faults in parallel).
constexpr size_t kPinBytes = (size_t)16 << 20;
constexpr int kCopyThreads = 4;
int copy_to_host(itd_engine *e, void *dst_host, const void *src_dev, size_t bytes, hipStream_t st)
{
    if (bytes < 4 * kPinBytes) {   // small: the plain path
        HIP_TRY(e, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        return ITD_OK;
    }
    for (int k = 0; k < 2; ++k)
        if (!e->h_pin[k]) {
            const hipError_t rc = hipHostMalloc(&e->h_pin[k], kPinBytes);
            if (rc != hipSuccess) {   // no pinned memory to be had: the plain path
                e->h_pin[k] = nullptr;
                HIP_TRY(e, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
                HIP_TRY(e, hipStreamSynchronize(st));
                return ITD_OK;
            }
        }
    const size_t n_chunks = (bytes + kPinBytes - 1) / kPinBytes;
    std::atomic<size_t> ready{0};                       // chunks whose DMA has completed
    std::unique_ptr<std::atomic<int>[]> done(new (std::nothrow) std::atomic<int>[n_chunks]);   // host threads done with chunk k
    if (!done) {
        HIP_TRY(e, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        return ITD_OK;
    }
    for (size_t k = 0; k < n_chunks; ++k) done[k].store(0, std::memory_order_relaxed);
    std::atomic<bool> abort_copy{false};
    auto worker = [&](int tix) {
        for (size_t k = 0; k < n_chunks; ++k) {
            while (ready.load(std::memory_order_acquire) <= k) {
                if (abort_copy.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t len = std::min(kPinBytes, bytes - k * kPinBytes);
            const size_t per = ((len + kCopyThreads - 1) / kCopyThreads + 63) & ~(size_t)63;
            const size_t lo = std::min(len, per * (size_t)tix), hi = std::min(len, lo + per);
            if (hi > lo) memcpy((char *)dst_host + k * kPinBytes + lo, (const char *)e->h_pin[k & 1] + lo, hi - lo);
            done[k].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    try {   // nothing may be thrown across the C ABI: without its host threads the copy takes the plain path
        pool.reserve(kCopyThreads);
        for (int t = 0; t < kCopyThreads; ++t) pool.emplace_back(worker, t);
    } catch (...) {
        abort_copy.store(true);
        for (auto &t : pool) t.join();
        HIP_TRY(e, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
        return ITD_OK;
    }
    hipError_t rc = hipSuccess;
    for (size_t k = 0; k < n_chunks && rc == hipSuccess; ++k) {
        if (k >= 2)   // the bounce buffer is free again once every host thread has copied chunk k-2 out of it
            while (done[k - 2].load(std::memory_order_acquire) < kCopyThreads) std::this_thread::yield();
        const size_t len = std::min(kPinBytes, bytes - k * kPinBytes);
        rc = hipMemcpyAsync(e->h_pin[k & 1], (const char *)src_dev + k * kPinBytes, len, hipMemcpyDeviceToHost, st);
        if (rc == hipSuccess) rc = hipStreamSynchronize(st);
        if (rc == hipSuccess) ready.store(k + 1, std::memory_order_release);
    }
    if (rc != hipSuccess) abort_copy.store(true);
    for (auto &t : pool) t.join();
    if (rc != hipSuccess) return fail_hip(e, rc, "copy_to_host");
    return ITD_OK;
}

}  // namespace

extern "C" {

int itd_abi_version(void) { return ITD_ABI_VERSION; }

const char *itd_status_string(int s)
{
    switch (s) {
        case ITD_OK: return "ok";
        case ITD_ERR_INVALID_ARG: return "invalid argument";
        case ITD_ERR_NO_DEVICE: return "no HIP device";
        case ITD_ERR_HIP: return "HIP runtime error";
        case ITD_ERR_NOMEM: return "out of memory";
        case ITD_ERR_NOT_RUN: return "no decomposition has been run";
        case ITD_ERR_NONFINITE: return "NaN in the input (rejected by this operator / mode)";
        default: return "unknown status";
    }
}

const char *itd_last_error(const itd_engine *e) { return e ? e->err : "null engine"; }

int itd_engine_create(itd_engine **out, int device_id, int64_t max_n, int32_t max_batch)
{
    if (!out || max_n < 3 || max_batch < 1 || max_n >= (int64_t)INT32_MAX) return ITD_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ITD_ERR_NO_DEVICE;
    itd_engine *e = new (std::nothrow) itd_engine();
    if (!e) return ITD_ERR_NOMEM;
    e->device = device_id;
    e->max_n = max_n;
    e->max_batch = max_batch;
    e->max_tiles = tiles_of(max_n);
    // (sweeps only, tools/pipeline_sweep.py: the batch pipeline's two constants)
    if (const char *v = getenv("PYITD_PIPE_GATE_US")) e->pipe_gate_timeout = 100ll * atoll(v);
    DevGuard g(device_id);
    const size_t B = (size_t)max_batch;
    const int max_groups = groups_of((int)e->max_tiles);
    e->tiles_half = (int64_t)B * e->max_tiles;
    e->gsum_third = (int64_t)B * max_groups * kGsumPitch;
    hipError_t rc = hipSuccess;
    auto alloc = [&](void **p, size_t bytes) { if (rc == hipSuccess) { rc = hipMalloc(p, bytes); if (rc == hipSuccess) { e->ws_bytes += (int64_t)bytes; poison(*p, bytes); } } };
    alloc((void **)&e->d_lists, (size_t)e->max_tiles * T * sizeof(int32_t));   // API helpers only (one signal)
    alloc((void **)&e->d_counts, 2 * (size_t)e->tiles_half * sizeof(int32_t));
    alloc((void **)&e->d_recs, 2 * (size_t)e->tiles_half * sizeof(TileRec));
    alloc((void **)&e->d_gsum, 2 * 3 * (size_t)e->gsum_third * sizeof(int32_t));
    alloc((void **)&e->d_kidx, (size_t)(max_n + 2) * sizeof(int32_t));
    alloc((void **)&e->d_total, 64);
    e->pp_pitch = max_n + kSlotPad;
    alloc((void **)&e->d_pp, B * 3 * (size_t)e->pp_pitch * sizeof(double));
    alloc((void **)&e->d_state, 2 * B * sizeof(SigState));
    e->dirty_sig[0] = e->dirty_sig[1] = max_batch;       // nothing is initialised yet
    e->dirty_gs[0] = e->dirty_gs[1] = e->gsum_third;
    e->hgsum_third = (int64_t)max_groups * kGsumPitch;
    alloc((void **)&e->d_hcounts, 2 * (size_t)e->max_tiles * sizeof(int32_t));
    alloc((void **)&e->d_hrecs, 2 * (size_t)e->max_tiles * sizeof(TileRec));
    alloc((void **)&e->d_hgsum, 3 * (size_t)e->hgsum_third * sizeof(int32_t));
    alloc((void **)&e->d_hstate, sizeof(SigState));
    alloc((void **)&e->d_flag, 64);
    alloc((void **)&e->d_need, sizeof(int32_t) * (size_t)max_batch);
    alloc((void **)&e->d_valid_own, sizeof(int32_t) * (size_t)max_batch);
    const size_t st_b = B * sizeof(SigState);
    if (rc == hipSuccess) rc = hipHostMalloc((void **)&e->h_state, st_b);
    if (rc == hipSuccess) rc = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
    if (rc != hipSuccess) {
        const bool oom = (rc == hipErrorOutOfMemory);
        itd_engine_destroy(e);
        return oom ? ITD_ERR_NOMEM : ITD_ERR_HIP;
    }
    *out = e;
    return ITD_OK;
}

void itd_engine_destroy(itd_engine *e)
{
    if (!e) return;
    DevGuard g(e->device);
    if (e->own_stream) (void)hipStreamSynchronize(e->own_stream);
    (void)hipFree(e->d_lists); (void)hipFree(e->d_counts); (void)hipFree(e->d_recs); (void)hipFree(e->d_total);
    (void)hipFree(e->d_kidx); (void)hipFree(e->d_pp); (void)hipFree(e->d_state); (void)hipFree(e->d_gsum);
    (void)hipFree(e->d_hcounts); (void)hipFree(e->d_hrecs); (void)hipFree(e->d_hgsum); (void)hipFree(e->d_hstate);
    (void)hipFree(e->d_io_x); (void)hipFree(e->d_io_rows); (void)hipFree(e->d_io_bases); (void)hipFree(e->d_iq_avg);
    (void)hipFree(e->d_cub); (void)hipFree(e->d_cub_e); (void)hipFree(e->d_dw); (void)hipFree(e->d_bw); (void)hipFree(e->d_kf); for (void *q : e->kf_retired) (void)hipFree(q); (void)hipFree(e->d_flag); (void)hipFree(e->d_need); (void)hipFree(e->d_valid_own);
    (void)hipFree(e->d_sp); (void)hipFree(e->d_sp2); (void)hipFree(e->d_wpe);
    if (e->h_state) (void)hipHostFree(e->h_state);
    if (e->h_kf) (void)hipHostFree(e->h_kf);
    if (e->h_pipe) (void)hipHostFree(e->h_pipe);
    for (int k = 0; k < 2; ++k) if (e->h_pin[k]) (void)hipHostFree(e->h_pin[k]);
    if (e->h_small) (void)hipHostFree(e->h_small);
    for (auto ev : e->ev) if (ev) (void)hipEventDestroy(ev);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    for (int k = 0; k < 3; ++k) {
        if (e->aux_stream[k]) { (void)hipStreamSynchronize(e->aux_stream[k]); (void)hipStreamDestroy(e->aux_stream[k]); }
        if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
    }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    for (auto ev : e->ev_pipe) if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(e->d_kf_started);
    delete e;
}

int64_t itd_engine_workspace_bytes(const itd_engine *e) { return e ? e->ws_bytes : 0; }

// ---- plain device-memory helpers: a host binding that owns no GPU allocator of its own (numpy callers, the C client) ----
int itd_dev_alloc(int device_id, int64_t bytes, void **out)
{
    if (!out || bytes <= 0) return ITD_ERR_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return ITD_ERR_NO_DEVICE;
    DevGuard g(device_id);
    const hipError_t rc = hipMalloc(out, (size_t)bytes);
    if (rc == hipErrorOutOfMemory) return ITD_ERR_NOMEM;
    return rc == hipSuccess ? ITD_OK : ITD_ERR_HIP;
}

int itd_dev_free(int device_id, void *p)
{
    if (!p) return ITD_OK;
    DevGuard g(device_id);
    return hipFree(p) == hipSuccess ? ITD_OK : ITD_ERR_HIP;
}

int itd_dev_copy(int device_id, void *dst, const void *src, int64_t bytes, int32_t to_device)
{
    if (!dst || !src || bytes < 0) return ITD_ERR_INVALID_ARG;
    DevGuard g(device_id);
    // Ordered against EVERYTHING on the device, whatever stream it was enqueued on: the engines' own streams are non-blocking
    // (no implicit ordering with the null stream this copy uses), and a caller of this plain helper expects "after what I
    // launched, before what I launch next"
    if (hipDeviceSynchronize() != hipSuccess) return ITD_ERR_HIP;
    return hipMemcpy(dst, src, (size_t)bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost) == hipSuccess
               ? ITD_OK : ITD_ERR_HIP;
}
int itd_engine_device(const itd_engine *e) { return e ? e->device : -1; }

// ---- sharding a batch of independent signals over the GPUs of a node, for hosts without torch.distributed ----
int itd_shard_range(int64_t batch, int32_t world, int32_t rank, int64_t *lo, int64_t *hi)
{
    if (batch < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return ITD_ERR_INVALID_ARG;
    const int64_t q = batch / world, r = batch % world;
    *lo = (int64_t)rank * q + std::min<int64_t>(rank, r);
    *hi = *lo + q + (rank < r ? 1 : 0);
    return ITD_OK;
}

namespace {
// RCCL's point-to-point entry points, resolved at the first scatter (the library does not link against librccl: a single-GPU
// host never needs it)
struct RcclApi {
    int (*send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    bool tried = false, ok = false;
};
RcclApi g_rccl;
bool rccl_load()
{
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return false;
    g_rccl.send = reinterpret_cast<decltype(g_rccl.send)>(dlsym(h, "ncclSend"));
    g_rccl.recv = reinterpret_cast<decltype(g_rccl.recv)>(dlsym(h, "ncclRecv"));
    g_rccl.group_start = reinterpret_cast<decltype(g_rccl.group_start)>(dlsym(h, "ncclGroupStart"));
    g_rccl.group_end = reinterpret_cast<decltype(g_rccl.group_end)>(dlsym(h, "ncclGroupEnd"));
    g_rccl.ok = g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end;
    return g_rccl.ok;
}
}  // namespace

int itd_shard_scatter(const void *x_root_dev, void *x_local_dev, int64_t n, int64_t batch, int32_t elem_bytes, int32_t world,
                      int32_t rank, int32_t root, void *nccl_comm, void *stream)
{
    int64_t lo = 0, hi = 0;
    if (n < 1 || (elem_bytes != 4 && elem_bytes != 8) || root < 0 || root >= world || itd_shard_range(batch, world, rank, &lo, &hi)) return ITD_ERR_INVALID_ARG;
    if ((rank == root && !x_root_dev) || (hi > lo && !x_local_dev)) return ITD_ERR_INVALID_ARG;
    const hipStream_t st = (hipStream_t)stream;
    const size_t row = (size_t)n * (size_t)elem_bytes;
    if (rank == root && hi > lo && x_local_dev != static_cast<const char *>(x_root_dev) + (size_t)lo * row)      // the root's own shard: a local copy
        if (hipMemcpyAsync(x_local_dev, static_cast<const char *>(x_root_dev) + (size_t)lo * row, (size_t)(hi - lo) * row, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return ITD_ERR_HIP;
    if (world == 1) return ITD_OK;
    if (!nccl_comm) return ITD_ERR_INVALID_ARG;
    if (!rccl_load()) return ITD_ERR_NO_DEVICE;            // no RCCL library on this host
    // one group of point-to-point transfers: every link of the root busy at once (xGMI is point to point), bytes as ncclChar (= 0)
    if (g_rccl.group_start()) return ITD_ERR_HIP;
    int rc = 0;
    if (rank == root) {
        for (int r = 0; r < world && !rc; ++r) {
            int64_t l = 0, h = 0;
            (void)itd_shard_range(batch, world, r, &l, &h);
            if (r != root && h > l) rc = g_rccl.send(static_cast<const char *>(x_root_dev) + (size_t)l * row, (size_t)(h - l) * row, 0, r, nccl_comm, st);
        }
    } else if (hi > lo) {
        rc = g_rccl.recv(x_local_dev, (size_t)(hi - lo) * row, 0, root, nccl_comm, st);
    }
    const int rc_end = g_rccl.group_end();
    return (rc || rc_end) ? ITD_ERR_HIP : ITD_OK;
}

int itd_decompose_f32(itd_engine *e, const float *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                      int32_t max_iteration, double *rows_dev, double *baselines_dev, void *stream)
{
    int rc = check_args(e, x_dev, n, batch, x_stride, max_iteration, rows_dev);
    if (rc) return rc;
    DevGuard g(e->device);
    return enqueue_any<float>(e, x_dev, n, batch, x_stride, max_iteration, rows_dev, baselines_dev,
                              stream ? (hipStream_t)stream : e->own_stream);
}

int itd_decompose_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                      int32_t max_iteration, double *rows_dev, double *baselines_dev, void *stream)
{
    int rc = check_args(e, x_dev, n, batch, x_stride, max_iteration, rows_dev);
    if (rc) return rc;
    DevGuard g(e->device);
    return enqueue_any<double>(e, x_dev, n, batch, x_stride, max_iteration, rows_dev, baselines_dev,
                               stream ? (hipStream_t)stream : e->own_stream);
}

// The signals of the last call whose fused levels reported a failure (h_state[b].kf_fail), each run again on its own: level by level,
// record-driven level 0, into its own rows (and baselines) of the caller's buffers; its state replaces h_state[b].  The engine's
// record of the last call (what a later itd_get_summary, itd_get_timing, ... refer to) is put back afterwards.
int repair_signals(itd_engine *e, int B)
{
    const auto keep_batch = e->last_batch; const auto keep_m = e->last_m; const auto keep_n = e->last_n; const auto keep_stream = e->last_stream;
    const void *keep_x = e->last_x; const bool keep_f32 = e->last_x_f32; const auto keep_xs = e->last_x_stride;
    double *keep_rows = e->last_rows, *keep_bases = e->last_bases;
    const bool keep_fused = e->last_fused; const int keep_kfl = e->last_kf_level, keep_form = e->last_kf_form;
    const bool keep_timing = e->timing;
    const int main_set = e->cur_set;
    const int64_t main_gs = e->dirty_gs[main_set];
    e->timing = false;                       // (the repairs are not part of any timed launch class)
    const int64_t rs = (int64_t)(keep_m + 2) * keep_n;
    int rc = ITD_OK;
    for (int b = 0; b < B && rc == ITD_OK; ++b) {
        if (!e->h_state[b].kf_fail) continue;
        double *rows_b = keep_rows + (int64_t)b * rs, *bases_b = keep_bases ? keep_bases + (int64_t)b * rs : nullptr;
        rc = keep_f32 ? enqueue_decompose<float>(e, (const float *)keep_x + (int64_t)b * keep_xs, keep_n, 1, keep_xs, keep_m, rows_b, bases_b, keep_stream, false)
                      : enqueue_decompose<double>(e, (const double *)keep_x + (int64_t)b * keep_xs, keep_n, 1, keep_xs, keep_m, rows_b, bases_b, keep_stream, false);
        if (rc) break;
        // (stream ordered: the copy leaves before the next repair's last launch puts this state set back into its initial state)
        if (hipMemcpyAsync(&e->h_state[b], e->d_state + (size_t)e->cur_set * e->max_batch, sizeof(SigState), hipMemcpyDeviceToHost, keep_stream) != hipSuccess) { rc = ITD_ERR_HIP; break; }
        ++e->fuse_signal_repairs;
    }
    if (hipStreamSynchronize(keep_stream) != hipSuccess && rc == ITD_OK) rc = ITD_ERR_HIP;
    e->timing = keep_timing;
    // The call's state set on the device becomes what the host now knows (fused verdicts and repaired signals merged), and the
    // engine's current set again: a later itd_get_summary of this call reads it as it stands (last_kf = false: nothing left to
    // draw or repair); the other set was last used by a one-signal repair.
    if (rc == ITD_OK) {
        for (int b = 0; b < B; ++b) e->h_state[b].kf_fail = 0;
        if (hipMemcpyAsync(e->d_state + (size_t)main_set * e->max_batch, e->h_state, sizeof(SigState) * (size_t)B, hipMemcpyHostToDevice, keep_stream) != hipSuccess ||
            hipStreamSynchronize(keep_stream) != hipSuccess) rc = ITD_ERR_HIP;
    }
    e->cur_set = main_set;
    e->dirty_sig[main_set] = std::max(e->dirty_sig[main_set], keep_batch);
    e->dirty_sig[main_set ^ 1] = std::max(e->dirty_sig[main_set ^ 1], 1);
    e->dirty_gs[main_set] = std::max(e->dirty_gs[main_set], main_gs);
    e->dirty_gs[main_set ^ 1] = std::max<int64_t>(e->dirty_gs[main_set ^ 1], (int64_t)groups_of((int)tiles_of(keep_n)) * kGsumPitch);
    e->last_batch = keep_batch; e->last_m = keep_m; e->last_n = keep_n; e->last_stream = keep_stream; e->last_x = keep_x; e->last_x_f32 = keep_f32;
    e->last_x_stride = keep_xs; e->last_rows = keep_rows; e->last_bases = keep_bases; e->last_fused = keep_fused; e->last_kf = false;
    e->last_kf_level = keep_kfl; e->last_kf_form = keep_form; e->last_resident = false; e->last_nan_input = false;
    return rc;
}

// The verdict of the fused sparse levels, drawn from the heads of their KfSig into the signals' states as the host sees them
// (no launch of its own behind the sample pass): the knot counts of the fused levels, the stop; or kf_fail — the sample pass found a
// knot the knot side had missed, a list outgrew its workspace, non-finite knot data, too many exact ties.
void kf_verdict(itd_engine *e, int B)
{
    const int L0 = e->last_kf_level;
    for (int b = 0; b < B; ++b) {
        KfSig ks;
        memcpy(&ks, e->h_kf + (size_t)b * kKfSigHead, kKfSigHead);
        SigState &st = e->h_state[b];
        if (!ks.active) {
            if (ks.fail) st.kf_fail = ks.fail;
            continue;
        }
        int fail = ks.fail;
        const int lend = ks.lend;
        if (lend < 0) fail |= kKfFailCapacity;      // the steps never reached a stop rule (cannot happen: they run to max_iteration + 1)
        // the last pending baseline feeds only the stop test (ITD.py:400-404): its exact count must take the same side of 2
        if (!fail && (ks.natural ? ks.m_exact >= 2 : ks.m_exact < 2)) fail |= kKfFailVerify;
        // (the levels' knot counts — what the stop rules were drawn from — the sample pass has compared with its verified flag words'.)
        // A level that was decomposed had at least 2 knots (ITD.py:404)
        for (int j = L0; !fail && j <= lend; ++j) if (j > L0 && ks.mlev[j] < 2) fail |= kKfFailVerify;
        if (fail) { st.kf_fail = fail; continue; }
        for (int j = L0 + 1; j <= lend; ++j) st.m[j] = ks.mlev[j];
        if (ks.cont) continue;          // capped fused levels that did not stop: the level launches behind them wrote the rest of the state
        st.m[lend + 1] = ks.m_exact;
        st.fin_stopped = ks.natural;
        st.fin_stop_level = ks.natural ? lend + 1 : -1;
    }
}

int itd_get_summary(itd_engine *e, int32_t *n_rows, int32_t *n_baselines, int32_t *stop_reason,
                    int64_t *knot_counts, int32_t *nan_levels)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    if (!e->ran) return ITD_ERR_NOT_RUN;
    DevGuard g(e->device);
    const int B = e->last_batch;
    HIP_TRY(e, hipMemcpyAsync(e->h_state, e->d_state + (size_t)e->cur_set * e->max_batch, sizeof(SigState) * (size_t)B, hipMemcpyDeviceToHost, e->last_stream));
    if (e->last_kf) {
        if (!e->h_kf) HIP_TRY(e, hipHostMalloc((void **)&e->h_kf, kKfSigHead * (size_t)e->max_batch));
        HIP_TRY(e, hipMemcpy2DAsync(e->h_kf, kKfSigHead, e->kf.sig, sizeof(KfSig), kKfSigHead, (size_t)B, hipMemcpyDeviceToHost, e->last_stream));
    }
    const bool piped = e->pipe_word_unread;
    if (piped) {
        if (!e->h_pipe) HIP_TRY(e, hipHostMalloc((void **)&e->h_pipe, 8));
        HIP_TRY(e, hipMemcpyAsync(e->h_pipe, e->d_kf_started + 1, 8, hipMemcpyDeviceToHost, e->last_stream));
    }
    HIP_TRY(e, hipStreamSynchronize(e->last_stream));
    if (e->last_kf) kf_verdict(e, B);
    e->pipe_word_unread = false;
    if (piped && *e->h_pipe) {
        // a gate of the pipelined batch gave up (k_kf_gate: the knot side it waited for did not start within its time-out — another
        // process's work on the device): a sample pass may have run in front of its own knot side.  Every signal is refused (the whole
        // call is repeated below) and this engine's batches rotate over the streams from now on.
        // (behind a device-side repair the call has been re-run level by level already: k_verdict read the same word)
        if (e->last_kf && e->last_pipelined) for (int b = 0; b < B; ++b) e->h_state[b].kf_fail |= kKfFailWait;
        HIP_TRY(e, hipMemsetAsync(e->d_kf_started + 1, 0, 8, e->last_stream));
        e->batch_pipeline = 0;
    }
    if (e->last_device_repair) {
        // the call carried its own repair (itd_set_device_repair): nothing to repeat here; count what it re-ran and let the engine's
        // next calls start the way that would have delivered (workloads tend to be homogeneous) — as the host-side repeats do
        int fixed = 0, why = 0;
        for (int b = 0; b < B; ++b) if (e->h_state[b].skip < 0) { ++fixed; why |= -e->h_state[b].skip; }
        e->device_repairs += fixed;
        e->last_device_repair = false;      // (a second summary of the same call counts nothing)
        if (fixed && (B < 8 || fixed * 8 > B)) {
            if ((why & 1) && !kf_back_off(e, (why >> 3) & 31, e->last_kf_level)) kf_levels_off(e);
            if (why & 2) e->l0_records_left = 16;
            if (why & 4) e->resident_off_left = 16;
        }
    }
    auto any_nan_input = [&]() {
        for (int b = 0; b < B; ++b) if (e->h_state[b].in_nan) return true;
        return false;
    };
    auto repeat = [&](bool f0, bool nan_in) {   // the same call again, level by level
        const int rc = e->last_x_f32
            ? enqueue_decompose<float>(e, (const float *)e->last_x, e->last_n, B, e->last_x_stride, e->last_m, e->last_rows, e->last_bases, e->last_stream, f0, nan_in)
            : enqueue_decompose<double>(e, (const double *)e->last_x, e->last_n, B, e->last_x_stride, e->last_m, e->last_rows, e->last_bases, e->last_stream, f0, nan_in);
        if (rc) return rc;
        HIP_TRY(e, hipMemcpyAsync(e->h_state, e->d_state + (size_t)e->cur_set * e->max_batch, sizeof(SigState) * (size_t)B, hipMemcpyDeviceToHost, e->last_stream));
        HIP_TRY(e, hipStreamSynchronize(e->last_stream));
        return (int)ITD_OK;
    };
    if (e->last_resident) {
        // the one-workgroup form handles finite data only: a NaN / infinity in the input or in a baseline (a leading or
        // trailing plateau, ITD.py:115-116) raised res_fail.  Repeat the call level by level — those kernels carry the
        // reference's NaN rules — and let the engine's next decompositions start that way: workloads tend to be homogeneous
        bool redo = false;
        for (int b = 0; b < B; ++b) redo = redo || e->h_state[b].res_fail;
        if (redo) {
            if (e->resident_mode == ITD_RESIDENT_ONLY) {
                snprintf(e->err, sizeof(e->err), "resident form: a non-finite sample or baseline (ITD_RESIDENT_ONLY forbids the level-by-level repeat)");
                return ITD_ERR_HIP;
            }
            ++e->resident_repeats;
            e->resident_off_left = 16;
            const int rc = repeat(want_fused(e), false);
            if (rc) return rc;
        }
    }
    if (!e->last_nan_input && e->nan_input_mode == ITD_NAN_INPUT_FOLLOW && any_nan_input()) {
        // A signal of the call holds a NaN.  The reference runs such input through detect_peaks' NaN branch and overwrites the
        // NaNs with +inf (ITD.py:46-51, 64-68); the launches so far evaluated plain rules.  Repeat the call with the level 0
        // that follows the reference (k_nan_level0); SigState::in_nan stays set and tells it which signals are concerned.
        const int rc = repeat(false, true);
        if (rc) return rc;
    }
    if (e->last_kf) {
        // the fused sparse levels deliver the reference's result or report that they cannot (SigState::kf_fail: the sample pass
        // found a knot the knot side had missed, a list / table outgrew its workspace, non-finite knot data, too many exact ties).
        // A few signals of a batch: each of them is run again on its own, level by level (record-driven level 0: any knot
        // spacing), into its rows — the rest of the batch keeps its fused result.  Many, or a single signal: the whole call is
        // repeated level by level and the engine's next decompositions start that way.
        int nfail = 0;
        for (int b = 0; b < B; ++b) nfail += e->h_state[b].kf_fail != 0;
        if (!nfail) {      // delivered: the back-off starts over
            e->fuse_off_span = 16; e->fuse_probe = false;
            if (!e->fuse_cap && e->fuse_cap_auto) {
                if (e->last_kf_cap) ++e->fuse_cap_calls;
                else if (e->last_m + 1 >= e->fuse_cap_auto) { e->fuse_cap_auto = 0; e->fuse_cap_calls = 0; e->fuse_cap_span = 16; }   // the probe without the cap was delivered: the workload has changed
            }
        }
        if (nfail) {
            if (e->fuse_mode == ITD_FUSE_ONLY) {
                int code = 0;
                for (int b = 0; b < B; ++b) code |= e->h_state[b].kf_fail;
                snprintf(e->err, sizeof(e->err), "fused sparse levels: not the reference's result (fail bits 0x%x: 1 verification, 2 capacity, 4 non-finite, 8 ties, 16 halo wait); ITD_FUSE_ONLY forbids the level-by-level repeat", code);
                return ITD_ERR_HIP;
            }
            // a list that outgrew its workgroup (dense knots): the calls after this one hand over a level later or run with half the
            // tiles per workgroup (kf_back_off)
            int bits = 0;
            for (int b = 0; b < B; ++b) bits |= e->h_state[b].kf_fail;
            const bool can_shrink = kf_back_off(e, bits, e->last_kf_level);
            if (B >= 8 && nfail * 8 <= B) {
                const int rc = repair_signals(e, B);
                if (rc) return rc;
            } else {
                ++e->fuse_repeats;
                // Where did it fail?  A workload whose fused form fails at the same level every time (periodic input whose baseline collapses
                // there: BASELINE configs[4]'s substitute clip at level 8) keeps the fused form for the levels in front of it: the engine's
                // next calls cap their fused levels at the lowest level anything failed at (KfSig::fail_lev) and run the rest level by level
                bool capped_next = false;
                if (!e->fuse_cap && !(bits & (kKfFailCapacity | kKfFailWait))) {
                    int fl = 99;
                    for (int b = 0; b < B; ++b)
                        if (e->h_state[b].kf_fail) fl = std::min<int>(fl, reinterpret_cast<const KfSig *>(e->h_kf + (size_t)b * kKfSigHead)->fail_lev);
                    const int L0 = e->last_kf_level;
                    if (fl >= L0 + 2 && fl <= e->last_m + 1 && (e->last_kf_cap == 0 || fl < e->last_kf_cap)) {
                        // (a refused PROBE — the cap was known, this call tried without it —: the next probe comes later)
                        e->fuse_cap_span = (e->fuse_cap_auto && !e->last_kf_cap) ? std::min(e->fuse_cap_span * 2, 1024) : 16;
                        e->fuse_cap_auto = fl; e->fuse_cap_calls = 0; capped_next = true;
                    } else { e->fuse_cap_auto = 0; e->fuse_cap_span = 16; }
                }
                if (!can_shrink && !capped_next) kf_levels_off(e);
                const int rc = repeat(want_fused(e), false);
                if (rc) return rc;
            }
        }
    }
    if (e->last_fused) {
        // the fused level-0 launch reaches kReach windows beyond a tile for its halo knots; a signal smoother than that
        // (knots more than ~4000 samples apart at level 0) raised l0_fail: repeat the call record-driven (k_scan0 + records),
        // and let the next decompositions of this engine start record-driven — workloads tend to be homogeneous
        bool fell_short = false;
        for (int b = 0; b < B; ++b) fell_short = fell_short || (e->h_state[b].l0_fail && !e->h_state[b].in_nan);
        if (fell_short) {
            if (e->l0_mode == ITD_LEVEL0_FUSED) {
                snprintf(e->err, sizeof(e->err), "fused level 0: a tile's halo knots lie beyond its reach (ITD_LEVEL0_FUSED forbids the record-driven repeat)");
                return ITD_ERR_HIP;
            }
            e->l0_records_left = 16;
            const int rc = e->last_x_f32
                ? enqueue_decompose<float>(e, (const float *)e->last_x, e->last_n, B, e->last_x_stride, e->last_m, e->last_rows, e->last_bases, e->last_stream, false)
                : enqueue_decompose<double>(e, (const double *)e->last_x, e->last_n, B, e->last_x_stride, e->last_m, e->last_rows, e->last_bases, e->last_stream, false);
            if (rc) return rc;
            HIP_TRY(e, hipMemcpyAsync(e->h_state, e->d_state + (size_t)e->cur_set * e->max_batch, sizeof(SigState) * (size_t)B, hipMemcpyDeviceToHost, e->last_stream));
            HIP_TRY(e, hipStreamSynchronize(e->last_stream));
        }
    }
    for (int b = 0; b < B; ++b) {
        const SigState &s = e->h_state[b];
        int rows, nb, why;
        if (s.fin_stopped) {           // ITD.py:404-416, counter = stop_level-1
            const int c = s.fin_stop_level - 1;
            rows = c + 1;
            nb = c;                    // baselines[0:counter-1] after the increment
            why = ITD_STOP_NATURAL;
        } else {                       // ITD.py:418-426, counter = max_iteration+1
            rows = e->last_m + 2;
            nb = e->last_m + 2;        // baselines[0:counter] (last row zero)
            why = ITD_STOP_TIMEOUT;
        }
        if (n_rows) n_rows[b] = rows;
        if (n_baselines) n_baselines[b] = nb;
        if (stop_reason) stop_reason[b] = why;
        if (knot_counts)
            for (int j = 0; j <= ITD_MAX_ROWS; ++j) knot_counts[(size_t)b * (ITD_MAX_ROWS + 1) + j] = s.m[j];
        if (nan_levels) nan_levels[b] = (s.in_nan && !e->last_nan_input) ? -2 : -1;
    }
    return ITD_OK;
}

int itd_set_level0_mode(itd_engine *e, int32_t mode)
{
    if (!e || mode < ITD_LEVEL0_AUTO || mode > ITD_LEVEL0_FUSED) return ITD_ERR_INVALID_ARG;
    e->l0_mode = mode;
    e->l0_records_left = 0;
    return ITD_OK;
}

int itd_set_nan_input_mode(itd_engine *e, int32_t mode)
{
    if (!e || (mode != ITD_NAN_INPUT_FOLLOW && mode != ITD_NAN_INPUT_REJECT)) return ITD_ERR_INVALID_ARG;
    e->nan_input_mode = mode;
    return ITD_OK;
}

int itd_set_host_keep_baselines(itd_engine *e, int32_t enable)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    e->host_keep_bases = enable != 0;
    if (!enable) e->kept_nb = -1;
    return ITD_OK;
}

int itd_get_last_baselines_host(itd_engine *e, double *baselines_host, int64_t n, int32_t n_baselines)
{
    if (!e || (!baselines_host && n_baselines > 0)) return ITD_ERR_INVALID_ARG;
    if (e->kept_nb < 0) return ITD_ERR_NOT_RUN;
    if (n != e->kept_n || n_baselines != e->kept_nb) return ITD_ERR_INVALID_ARG;
    if (n_baselines == 0) return ITD_OK;
    DevGuard g(e->device);
    return copy_to_host(e, baselines_host, e->d_io_bases, (size_t)n_baselines * (size_t)n * sizeof(double), e->own_stream);
}

int itd_set_resident_mode(itd_engine *e, int32_t mode)
{
    if (!e || mode < ITD_RESIDENT_AUTO || mode > ITD_RESIDENT_ONLY) return ITD_ERR_INVALID_ARG;
    e->resident_mode = mode;
    e->resident_off_left = 0;
    return ITD_OK;
}

int itd_get_resident_repeats(const itd_engine *e) { return e ? e->resident_repeats : -1; }

int itd_set_valid_flags(itd_engine *e, int32_t *valid_dev)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    e->valid_dev = valid_dev;
    return ITD_OK;
}

int itd_set_device_repair(itd_engine *e, int32_t on)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    e->device_repair = on != 0;
    return ITD_OK;
}

int64_t itd_get_device_repairs(const itd_engine *e) { return e ? e->device_repairs : -1; }

int itd_set_fuse_range(itd_engine *e, int32_t tiles)
{
    if (!e || (tiles != 0 && tiles != 16 && tiles != 32 && tiles != 64)) return ITD_ERR_INVALID_ARG;
    e->fuse_range = tiles;
    e->kf_shrink = 0;
    return ITD_OK;
}

int itd_set_fuse_mode(itd_engine *e, int32_t mode)
{
    if (!e || mode < ITD_FUSE_AUTO || mode > ITD_FUSE_ONLY) return ITD_ERR_INVALID_ARG;
    e->fuse_mode = mode;
    e->fuse_off_left = 0;
    e->fuse_off_span = 16;
    e->fuse_probe = false;
    return ITD_OK;
}

int itd_set_fuse_level(itd_engine *e, int32_t first_fused_level)
{
    // (level 1's launch completes the signal's own knot count, and a level-1 list would not fit the workspace: 2 at least)
    if (!e || (first_fused_level != 0 && (first_fused_level < 2 || first_fused_level > ITD_MAX_ITERATION))) return ITD_ERR_INVALID_ARG;
    e->fuse_level = first_fused_level;
    e->fuse_level2_off = false;
    return ITD_OK;
}

int itd_set_fuse_group(itd_engine *e, int32_t chunks)
{
    if (!e || chunks < 1 || chunks > 1024) return ITD_ERR_INVALID_ARG;
    e->fuse_group = chunks;
    return ITD_OK;
}

int itd_set_fuse_min_samples(itd_engine *e, int64_t samples)
{
    if (!e || samples < 0) return ITD_ERR_INVALID_ARG;
    e->fuse_min_samples = samples;
    return ITD_OK;
}

int itd_debug_int_ratio_check(int device, int32_t max_den, int64_t *mismatches)
{
    if (!mismatches || max_den < 1) return ITD_ERR_INVALID_ARG;
    DevGuard g(device);
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(d); return ITD_ERR_HIP; }
    k_int_ratio_check<<<1024, 256>>>(max_den, d);
    const bool ok = hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    *mismatches = (int64_t)h;
    return ok ? ITD_OK : ITD_ERR_HIP;
}

int itd_debug_kf_fault(itd_engine *e, int32_t kind, int32_t level, int32_t where, int32_t slot, int32_t delta)
{
    if (!e || kind > 8 || (kind >= 0 && (level < 2 || level > ITD_MAX_ITERATION + 1 || where < 0 || slot < 0))) return ITD_ERR_INVALID_ARG;
    if ((kind == 6 || kind == 7) && slot > 4) return ITD_ERR_INVALID_ARG;
    e->fault_kind = kind < 0 ? -1 : kind;
    e->fault_level = level; e->fault_where = where; e->fault_slot = slot; e->fault_delta = delta;
    return ITD_OK;
}

int itd_debug_kf_fault_signal(itd_engine *e, int32_t signal)
{
    if (!e || signal < 0) return ITD_ERR_INVALID_ARG;
    e->fault_sig = signal;
    return ITD_OK;
}

int itd_set_fuse_cap(itd_engine *e, int32_t first_level_not_fused)
{
    if (!e || first_level_not_fused < -1 || first_level_not_fused > ITD_MAX_ITERATION + 1 || (first_level_not_fused > 0 && first_level_not_fused < 4)) return ITD_ERR_INVALID_ARG;
    e->fuse_cap = first_level_not_fused;
    e->fuse_cap_auto = 0;
    e->fuse_cap_calls = 0;
    e->fuse_cap_span = 16;
    return ITD_OK;
}
int itd_get_last_fuse_cap(const itd_engine *e) { return !e ? -1 : (e->ran && e->last_kf_form ? e->last_kf_cap_form : 0); }

int itd_get_fuse_repeats(const itd_engine *e) { return e ? e->fuse_repeats : -1; }
int itd_get_last_fuse_level(const itd_engine *e) { return !e ? -1 : (e->ran && e->last_kf_form ? e->last_kf_form : 0); }
int64_t itd_get_fuse_signal_repairs(const itd_engine *e) { return e ? e->fuse_signal_repairs : -1; }

int itd_set_resident_window(itd_engine *e, int32_t segments)
{
    if (!e || segments < 0 || (segments > 0 && segments < 8)) return ITD_ERR_INVALID_ARG;
    e->resident_window = segments;
    return ITD_OK;
}

#if ITD_PROF
// diagnostic build only (tools/level0_prof.py): where the fused level-0 launch's wavefronts leave their phase times
extern "C" int itd_debug_prof_buffer(void *dev_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(dev_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_buf), &p, sizeof(p)) == hipSuccess ? ITD_OK : ITD_ERR_HIP;
}
// (tools/knots_prof.py): the knot side's workgroups' phase marks
extern "C" int itd_debug_knots_prof_buffer(void *dev_buf)
{
    unsigned long long *p = static_cast<unsigned long long *>(dev_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_kc_prof), &p, sizeof(p)) == hipSuccess ? ITD_OK : ITD_ERR_HIP;
}
#endif

int itd_set_batch_streams(itd_engine *e, int32_t streams)
{
    if (!e || streams < 1 || streams > 4) return ITD_ERR_INVALID_ARG;
    e->batch_streams = streams;
    return ITD_OK;
}

int itd_set_batch_pipeline(itd_engine *e, int32_t on)
{
    if (!e || on < 0 || on > 1) return ITD_ERR_INVALID_ARG;
    e->batch_pipeline = on;
    return ITD_OK;
}

int itd_set_batch_chunk(itd_engine *e, int32_t signals_per_chunk)
{
    if (!e || signals_per_chunk < 0) return ITD_ERR_INVALID_ARG;
    e->chunk = signals_per_chunk;
    return ITD_OK;
}

}  // extern "C"

namespace {
template <typename Tin>
int decompose_host(itd_engine *e, const Tin *x_host, int64_t n, int32_t M, double *rows_host, double *bases_host,
                   int32_t *n_rows, int32_t *n_baselines, int32_t *stop_reason, int64_t *knot_counts)
{
    if (!e || !x_host || !rows_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n || M < 0 || M > ITD_MAX_ITERATION) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    const size_t R = (size_t)M + 2;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(Tin));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, R * (size_t)n * sizeof(double));
    if (rc) return rc;
    const bool dev_bases = bases_host || e->host_keep_bases;
    e->kept_nb = -1;
    if (dev_bases) {
        rc = grow(e, &e->d_io_bases, &e->io_bases_bytes, R * (size_t)n * sizeof(double));
        if (rc) return rc;
    }
    hipStream_t st = e->own_stream;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, (size_t)n * sizeof(Tin), hipMemcpyHostToDevice, st));
    rc = enqueue_any<Tin>(e, (const Tin *)e->d_io_x, n, 1, n, M, e->d_io_rows, dev_bases ? e->d_io_bases : nullptr, st);
    if (rc) return rc;
    int32_t nr = 0, nb = 0, why = 0, nanlv = -1;
    int64_t kc[ITD_MAX_ROWS + 1];
    rc = itd_get_summary(e, &nr, &nb, &why, kc, &nanlv);
    if (rc) return rc;
    rc = copy_to_host(e, rows_host, e->d_io_rows, (size_t)nr * n * sizeof(double), st);
    if (rc) return rc;
    if (bases_host) {
        rc = copy_to_host(e, bases_host, e->d_io_bases, (size_t)nb * n * sizeof(double), st);
        if (rc) return rc;
    }
    if (n_rows) *n_rows = nr;
    if (n_baselines) *n_baselines = nb;
    if (stop_reason) *stop_reason = why;
    if (knot_counts) memcpy(knot_counts, kc, sizeof(kc));
    if (dev_bases && nanlv == -1) { e->kept_n = n; e->kept_nb = nb; }
    return nanlv != -1 ? ITD_ERR_NONFINITE : ITD_OK;
}

// level-0 knot scan of one device signal into the workspace of batch slot 0; optional ordered list in d_kidx
// nan_xm: the scanned signal holds a NaN (found by a first scan, SigState::in_nan kept): the knot set the reference's NaN
// branch gives (k_nan_level0), and — if non-null — the mutated float64 copy of the signal
template <typename Tin>
int scan_level0(itd_engine *e, const Tin *x, int64_t n, int mode, bool compact, hipStream_t st, int64_t tail_value = -1,
                bool nan_follow = false, double *nan_xm = nullptr)
{
    const int n_tiles = (int)tiles_of(n);
    const dim3 grid_t(n_tiles, 1), blk(kWave);
    // the helpers' own state, counts, records and group sums: a decomposition's workspace is never touched
    k_init_state<<<(unsigned)std::min<int64_t>((3 * e->hgsum_third + 255) / 256 + 1, 2048), 256, 0, st>>>(e->d_hstate, 1, e->d_hgsum,
                                                                                                   3 * e->hgsum_third, nan_follow ? 1 : 0);
    if constexpr (std::is_same<Tin, double>::value) {
        if (!nan_follow && compact && (mode == (int)kCpp || mode == (int)kZeroCross)) {       // the short pair (itd_detect_fast.hpp)
            unsigned long long *fw = reinterpret_cast<unsigned long long *>(e->d_lists);
            if (mode == (int)kCpp) k_detect_fast<(int)kCpp><<<grid_t, blk, 0, st>>>(x, n, n, n_tiles, e->d_hcounts, fw, e->d_hgsum, e->d_hstate);
            else k_detect_fast<(int)kZeroCross><<<grid_t, blk, 0, st>>>(x, n, n, n_tiles, e->d_hcounts, fw, e->d_hgsum, e->d_hstate);
            k_compact_fast<<<grid_t, blk, 0, st>>>(fw, e->d_hcounts, e->d_hgsum, n_tiles, n, e->d_kidx, e->max_n + 2, e->d_total, e->d_hstate,
                                                   tail_value, e->d_hcounts + e->max_tiles, 1);
            HIP_TRY(e, hipGetLastError());
            return ITD_OK;
        }
    }
    if (nan_follow)
        k_nan_level0<Tin, T><<<grid_t, blk, 0, st>>>(x, n, n, n_tiles, nan_xm, n, e->d_hcounts, e->d_hrecs, e->d_hgsum, e->d_hstate,
                                                     mode, compact ? e->d_lists : nullptr);
    else
    k_detect<Tin, T><<<grid_t, blk, 0, st>>>(x, n, n, n_tiles, mode, compact ? e->d_lists : nullptr, e->d_hcounts,
                                              e->d_hrecs, e->d_hgsum, e->d_hstate);
    if (compact)
        // (the second half of the helpers' count buffer receives every tile's knot base: the instantaneous-frequency step uses it)
        k_compact<T><<<grid_t, blk, 0, st>>>(e->d_lists, e->d_hcounts, e->d_hgsum, n_tiles, n, e->d_kidx, e->max_n + 2,
                                              e->d_total, e->d_hstate, tail_value, e->d_hcounts + e->max_tiles);
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

// the knot total of the last helper scan; d_total[1] = the scanned signal held a NaN (the reference's detect_peaks would
// take its NaN branch and write +inf into the caller's array: rejected, like NaN input of a decomposition)
int fetch_total(itd_engine *e, hipStream_t st, int64_t *m_host, bool nan_is_error = true)
{
    int32_t m32[2] = {0, 0};
    HIP_TRY(e, hipMemcpyAsync(m32, e->d_total, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    *m_host = m32[0];
    return (m32[1] && nan_is_error) ? ITD_ERR_NONFINITE : ITD_OK;
}

// the three knot sets of the reference's own detect functions follow its NaN branch when the signal holds a NaN
inline bool nan_follows(const itd_engine *e, int mode)
{
    return e->nan_input_mode == ITD_NAN_INPUT_FOLLOW && (mode == (int)kKnots || mode == (int)kValleys || mode == (int)kPeaks);
}

template <typename Tin>
int extract_dev(itd_engine *e, const Tin *x, int64_t n, double *rot, double *base, int32_t *knots, int64_t *m_host,
                hipStream_t st, bool want_sync, bool *took_nan_path = nullptr)
{
    if (!e || !x || !rot || !base) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    const int n_tiles = (int)tiles_of(n);
    const bool want_list = m_host || knots || want_sync;
    int rc = scan_level0<Tin>(e, x, n, (int)kKnots, want_list, st);   // the ordered list must be taken before
    if (rc) return rc;                                                 // k_extract rewrites the per-tile lists
    k_extract<Tin, T, false, kRankCap0, kTilesPerWave><<<dim3((n_tiles + kTilesPerWave - 1) / kTilesPerWave), kWave, 0, st>>>(x, n, n, n_tiles, 1, e->d_hcounts,
                                                      e->d_hcounts + e->max_tiles, e->d_hrecs, e->d_hrecs + e->max_tiles,
                                                      e->d_hgsum, e->d_hgsum + e->hgsum_third, e->d_hgsum + 2 * e->hgsum_third,
                                                      rot, n, base, n, e->d_hstate, 0, 1);
    HIP_TRY(e, hipGetLastError());
    if (want_list) {
        int64_t m = 0;
        rc = fetch_total(e, st, &m, !nan_follows(e, (int)kKnots));
        if (rc) return rc;
        int32_t has_nan = 0;
        HIP_TRY(e, hipMemcpy(&has_nan, e->d_total + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (took_nan_path) *took_nan_path = has_nan != 0;
        if (has_nan) {
            // NaN in the signal: again, the way the reference runs it (ITD.py:87-88 over :46-51, 64-68): the mutated float64 copy
            // goes into a staging buffer, the record-driven extraction reads that
            rc = grow(e, &e->d_cub, &e->cub_bytes, (size_t)n * sizeof(double));
            if (rc) return rc;
            double *xm = (double *)e->d_cub;
            rc = scan_level0<Tin>(e, x, n, (int)kKnots, want_list, st, -1, true, xm);
            if (rc) return rc;
            k_extract<double, T, false, kRankCap0, kTilesPerWave><<<dim3((n_tiles + kTilesPerWave - 1) / kTilesPerWave), kWave, 0, st>>>(
                xm, n, n, n_tiles, 1, e->d_hcounts, e->d_hcounts + e->max_tiles, e->d_hrecs, e->d_hrecs + e->max_tiles, e->d_hgsum,
                e->d_hgsum + e->hgsum_third, e->d_hgsum + 2 * e->hgsum_third, rot, n, base, n, e->d_hstate, 0, 1);
            HIP_TRY(e, hipGetLastError());
            rc = fetch_total(e, st, &m, false);
            if (rc) return rc;
        }
        if (m_host) *m_host = m;
        if (knots && m > 0) {
            HIP_TRY(e, hipMemcpyAsync(knots, e->d_kidx + 1, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToDevice, st));
            HIP_TRY(e, hipStreamSynchronize(st));
        }
    }
    return ITD_OK;
}

template <typename Tin>
int detect_dev(itd_engine *e, const Tin *x, int64_t n, int32_t mode, int32_t *idx, int64_t *count, hipStream_t st)
{
    if (!e || !x || !count) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n || mode < 0 || mode > 4) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    int rc = scan_level0<Tin>(e, x, n, mode, true, st);
    if (rc) return rc;
    rc = fetch_total(e, st, count, !nan_follows(e, mode));
    if (rc) return rc;
    if (nan_follows(e, mode)) {
        int32_t has_nan = 0;
        HIP_TRY(e, hipMemcpy(&has_nan, e->d_total + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
        if (has_nan) {   // the reference's NaN branch (ITD.py:46-51, 64-68; numba_accelerated_itd.py:28-49)
            rc = scan_level0<Tin>(e, x, n, mode, true, st, -1, true, nullptr);
            if (rc) return rc;
            rc = fetch_total(e, st, count, false);
            if (rc) return rc;
        }
    }
    if (idx && *count > 0) {
        HIP_TRY(e, hipMemcpyAsync(idx, e->d_kidx + 1, sizeof(int32_t) * (size_t)*count, hipMemcpyDeviceToDevice, st));
        HIP_TRY(e, hipStreamSynchronize(st));
    }
    return ITD_OK;
}
}  // namespace

extern "C" {

int itd_decompose_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t max_iteration, double *rows_host,
                           double *baselines_host, int32_t *n_rows, int32_t *n_baselines, int32_t *stop_reason,
                           int64_t *knot_counts)
{
    return decompose_host<double>(e, x_host, n, max_iteration, rows_host, baselines_host, n_rows, n_baselines,
                                  stop_reason, knot_counts);
}

int itd_decompose_host_f32(itd_engine *e, const float *x_host, int64_t n, int32_t max_iteration, double *rows_host,
                           double *baselines_host, int32_t *n_rows, int32_t *n_baselines, int32_t *stop_reason,
                           int64_t *knot_counts)
{
    return decompose_host<float>(e, x_host, n, max_iteration, rows_host, baselines_host, n_rows, n_baselines,
                                 stop_reason, knot_counts);
}

int itd_baseline_extract_f64(itd_engine *e, const double *x_dev, int64_t n, double *rot_dev, double *base_dev,
                             int32_t *knots_dev, int64_t *m_host, void *stream)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    return extract_dev<double>(e, x_dev, n, rot_dev, base_dev, knots_dev, m_host,
                               stream ? (hipStream_t)stream : e->own_stream, false);
}

int itd_baseline_extract_f32(itd_engine *e, const float *x_dev, int64_t n, double *rot_dev, double *base_dev,
                             int32_t *knots_dev, int64_t *m_host, void *stream)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    return extract_dev<float>(e, x_dev, n, rot_dev, base_dev, knots_dev, m_host,
                              stream ? (hipStream_t)stream : e->own_stream, false);
}

int itd_baseline_extract_host_f64(itd_engine *e, const double *x_host, int64_t n, double *rot_host, double *base_host,
                                  int64_t *knots_host, int64_t *m_host, double *bk_host)
{
    if (!e || !x_host || !rot_host || !base_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    // rows staging: [rot | base | bk(n+2) | knots64(n)]
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)(4 * n + 4) * sizeof(double));
    if (rc) return rc;
    hipStream_t st = e->own_stream;
    double *d_rot = e->d_io_rows, *d_base = d_rot + n, *d_bk = d_base + n;
    int64_t *d_k64 = (int64_t *)(d_bk + n + 2);
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    int64_t m = 0;
    bool nan_path = false;
    rc = extract_dev<double>(e, (const double *)e->d_io_x, n, d_rot, d_base, nullptr, &m, st, true, &nan_path);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(rot_host, d_rot, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipMemcpyAsync(base_host, d_base, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (knots_host && m > 0) {
        k_widen_idx<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(e->d_kidx + 1, d_k64, m);
        HIP_TRY(e, hipMemcpyAsync(knots_host, d_k64, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    if (bk_host) {
        // (a signal that holds NaNs: the knot values of the mutated copy, as the reference computes them after detect_peaks' write)
        k_knot_values<double><<<(unsigned)((m + 2 + 255) / 256), 256, 0, st>>>(nan_path ? (const double *)e->d_cub : (const double *)e->d_io_x, n, e->d_kidx, (int)m, d_bk);
        HIP_TRY(e, hipMemcpyAsync(bk_host, d_bk, (size_t)(m + 2) * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(e, hipStreamSynchronize(st));
    if (m_host) *m_host = m;
    return ITD_OK;
}

int itd_detect_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t mode, int32_t *idx_dev, int64_t *count_host,
                   void *stream)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    return detect_dev<double>(e, x_dev, n, mode, idx_dev, count_host, stream ? (hipStream_t)stream : e->own_stream);
}

int itd_detect_f32(itd_engine *e, const float *x_dev, int64_t n, int32_t mode, int32_t *idx_dev, int64_t *count_host,
                   void *stream)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    return detect_dev<float>(e, x_dev, n, mode, idx_dev, count_host, stream ? (hipStream_t)stream : e->own_stream);
}

int itd_detect_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t mode, int64_t *idx_host,
                        int64_t *count_host)
{
    if (!e || !x_host || !count_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)n * sizeof(int64_t));
    if (rc) return rc;
    hipStream_t st = e->own_stream;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    int64_t m = 0;
    rc = detect_dev<double>(e, (const double *)e->d_io_x, n, mode, nullptr, &m, st);
    if (rc) return rc;
    if (idx_host && m > 0) {
        int64_t *d_k64 = (int64_t *)e->d_io_rows;
        k_widen_idx<<<(unsigned)((m + 255) / 256), 256, 0, st>>>(e->d_kidx + 1, d_k64, m);
        HIP_TRY(e, hipMemcpyAsync(idx_host, d_k64, (size_t)m * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
    }
    *count_host = m;
    return ITD_OK;
}

int itd_knot_values_host_f64(itd_engine *e, const double *x_host, int64_t n, const int64_t *extrema_host, int64_t m,
                             double *bk_host)
{
    if (!e || !x_host || !extrema_host || !bk_host) return ITD_ERR_INVALID_ARG;
    if (n < 2 || n > e->max_n || m < 0 || m + 2 > e->max_n + 2) return ITD_ERR_INVALID_ARG;
    for (int64_t k = 0; k < m + 2; ++k)
        if (extrema_host[k] < 0 || extrema_host[k] >= n) return ITD_ERR_INVALID_ARG;
    if (m == 0) return ITD_OK;
    DevGuard g(e->device);
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)(m + 2) * sizeof(double));
    if (rc) return rc;
    hipStream_t st = e->own_stream;
    std::vector<int32_t> e32((size_t)m + 2);
    for (int64_t k = 0; k < m + 2; ++k) e32[(size_t)k] = (int32_t)extrema_host[k];
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(e, hipMemcpyAsync(e->d_kidx, e32.data(), (size_t)(m + 2) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    k_knot_values<double><<<(unsigned)((m + 2 + 255) / 256), 256, 0, st>>>((const double *)e->d_io_x, n, e->d_kidx, (int)m, e->d_io_rows);
    // interior values only: bk[0] and bk[m+1] are the caller's (numba_accelerated_itd.py:171)
    HIP_TRY(e, hipMemcpyAsync(bk_host + 1, e->d_io_rows + 1, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

int itd_knot_values_f64(itd_engine *e, const double *x_dev, int64_t n, const int32_t *extrema_dev, int64_t m, double *bk_dev,
                        void *stream)
{
    if (!e || !x_dev || !extrema_dev || !bk_dev) return ITD_ERR_INVALID_ARG;
    if (n < 2 || n >= (int64_t)INT32_MAX || m < 0 || m > n) return ITD_ERR_INVALID_ARG;
    if (m == 0) return ITD_OK;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    k_knot_values<double><<<(unsigned)((m + 2 + 255) / 256), 256, 0, st>>>(x_dev, n, extrema_dev, (int)m, bk_dev, 1);
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The cubic-spline baseline variant (itd_cubic.hpp): itd_baseline_extract_fast(I, extrema_input, idx),
// itd_fourier_decomposition.py:49-122 = itd.cpp:156-239.  Single-level operator; synchronous like the other helpers.
// ---------------------------------------------------------------------------------------------
namespace {
// Batched knot detection into a workspace of its own (grow-only): per-tile lists, counts, records, group sums, per-signal
// states, the ordered knot lists kidx[b] = [lead slot, knots, tail] and totals[b] = {knot count, the signal holds a NaN}.
struct DetectWs {
    int32_t *lists, *counts, *gsum, *kidx, *totals, *tbase;
    TileRec *recs;
    SigState *state;
    int64_t kidx_stride;
    int n_tiles, n_groups;
};
int detect_workspace(itd_engine *e, int64_t n, int batch, DetectWs &w)
{
    w.n_tiles = (int)tiles_of(n);
    w.n_groups = groups_of(w.n_tiles);
    w.kidx_stride = n + 2;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t B = (size_t)batch, tiles = B * (size_t)w.n_tiles;
    const size_t b_lists = al(tiles * T * sizeof(int32_t)), b_counts = al(tiles * sizeof(int32_t)), b_recs = al(tiles * sizeof(TileRec));
    const size_t b_gsum = al(B * (size_t)w.n_groups * kGsumPitch * sizeof(int32_t)), b_state = al(B * sizeof(SigState));
    const size_t b_kidx = al(B * (size_t)w.kidx_stride * sizeof(int32_t)), b_tot = al(B * 2 * sizeof(int32_t));
    const int rc = grow(e, &e->d_dw, &e->dw_bytes, b_lists + 2 * b_counts + b_recs + b_gsum + b_state + b_kidx + b_tot);
    if (rc) return rc;
    char *p = (char *)e->d_dw;
    w.lists = (int32_t *)p; p += b_lists;
    w.counts = (int32_t *)p; p += b_counts;
    w.recs = (TileRec *)p; p += b_recs;
    w.gsum = (int32_t *)p; p += b_gsum;
    w.state = (SigState *)p; p += b_state;
    w.kidx = (int32_t *)p; p += b_kidx;
    w.totals = (int32_t *)p; p += b_tot;
    w.tbase = (int32_t *)p;          // the knots in front of every tile (k_compact)
    return ITD_OK;
}
// knots of `batch` signals (batch <= 65535: grid.y) by predicate `mode`, ordered, no host synchronisation.  kidx_out = NULL: into
// the workspace's lists with a leading slot (what the cubic kernels and the stream's selection read); else the caller's
// [batch][kidx_out_stride] array without one (itd_detect_batch_*)
int detect_enqueue(itd_engine *e, const double *x, int64_t x_stride, int64_t n, int batch, int mode, int64_t tail_value,
                   hipStream_t st, DetectWs &w, int32_t *kidx_out = nullptr, int64_t kidx_out_stride = 0, bool want_lists = true)
{
    int rc = detect_workspace(e, n, batch, w);
    if (rc) return rc;
    const dim3 grid_t(w.n_tiles, batch), blk(kWave);
    const int64_t ge = (int64_t)batch * w.n_groups * kGsumPitch;
    k_init_state<<<(unsigned)std::min<int64_t>(std::max<int64_t>((ge + 255) / 256, (batch + 255) / 256), 2048), 256, 0, st>>>(w.state, batch, w.gsum, ge);
    if (want_lists && (mode == (int)kCpp || mode == (int)kZeroCross)) {
        // the two predicates without NaN rules and without an extraction behind them: the short pair (itd_detect_fast.hpp); the
        // tiles' flag words live where the general pair keeps its per-tile position lists
        unsigned long long *fw = reinterpret_cast<unsigned long long *>(w.lists);
        if (mode == (int)kCpp) k_detect_fast<(int)kCpp><<<grid_t, blk, 0, st>>>(x, x_stride, n, w.n_tiles, w.counts, fw, w.gsum, w.state);
        else k_detect_fast<(int)kZeroCross><<<grid_t, blk, 0, st>>>(x, x_stride, n, w.n_tiles, w.counts, fw, w.gsum, w.state);
        k_compact_fast<<<grid_t, blk, 0, st>>>(fw, w.counts, w.gsum, w.n_tiles, n, kidx_out ? kidx_out : w.kidx,
                                               kidx_out ? kidx_out_stride : w.kidx_stride, w.totals, w.state, tail_value, w.tbase, kidx_out ? 0 : 1);
        HIP_TRY(e, hipGetLastError());
        return ITD_OK;
    }
    k_detect<double, T><<<grid_t, blk, 0, st>>>(x, x_stride, n, w.n_tiles, mode, want_lists ? w.lists : nullptr, w.counts, w.recs, w.gsum, w.state);
    if (want_lists)
        k_compact<T><<<grid_t, blk, 0, st>>>(w.lists, w.counts, w.gsum, w.n_tiles, n, kidx_out ? kidx_out : w.kidx,
                                              kidx_out ? kidx_out_stride : w.kidx_stride, w.totals, w.state, tail_value, w.tbase, kidx_out ? 0 : 1);
    else
        k_batch_totals<<<(batch + 3) / 4, 256, 0, st>>>(w.gsum, w.n_groups, batch, w.state, w.totals);
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

// The cubic operator over `batch` signals, asynchronous on st.  extrema = NULL: every signal's own knots (itd.cpp:159-169);
// else the caller's list(s) of idx + 1 entries (e_stride = 0: one list for every signal, itd.cpp:40-44).  *jobs_out: the
// per-signal jobs on the device (idx used, valid, status) for callers that synchronise afterwards.
int cubic_batch(itd_engine *e, const double *x, int64_t n, int batch, int64_t x_stride, const int32_t *extrema, int64_t e_stride,
                int64_t idx, double *baseline, int64_t b_stride, hipStream_t st, const CubicJob **jobs_out, int *n_jobs_out)
{
    const int64_t L = (extrema ? idx : n) + 2;                        // entries per knot array
    const size_t jobs_b = (((size_t)batch * sizeof(CubicJob)) + 255) & ~(size_t)255;
    int rc = grow(e, &e->d_cub, &e->cub_bytes, jobs_b + 3 * (size_t)batch * (size_t)L * sizeof(double));
    if (rc) return rc;
    CubicJob *jobs = (CubicJob *)e->d_cub;
    double *arr = (double *)((char *)e->d_cub + jobs_b);
    CubicArgs A;
    A.x = x; A.x_stride = x_stride; A.n = n;
    A.K = arr; A.bf = arr + (size_t)batch * L; A.b = arr + 2 * (size_t)batch * L; A.a_stride = L;
    A.jobs = jobs;
    int n_jobs;
    int64_t max_count;
    if (!extrema) {
        DetectWs w;
        rc = detect_enqueue(e, x, x_stride, n, batch, (int)kCpp, 0 /* e[idx] = 0: the file's static array at first call */, st, w);
        if (rc) return rc;
        n_jobs = batch;
        k_cubic_jobs<<<(batch + 255) / 256, 256, 0, st>>>(jobs, batch, 1, 0, w.totals);
        A.e = w.kidx; A.e_stride = w.kidx_stride; A.job_stride = 1;
        A.tbase = w.tbase; A.tb_stride = w.n_tiles;
        max_count = n;
    } else {
        n_jobs = e_stride ? batch : 1;
        k_cubic_jobs<<<(n_jobs + 255) / 256, 256, 0, st>>>(jobs, n_jobs, 0, idx, nullptr);
        k_cubic_validate<<<dim3((unsigned)((idx + 1 + 255) / 256), n_jobs), 256, 0, st>>>(extrema, e_stride, idx, n, jobs);
        A.e = extrema; A.e_stride = e_stride; A.job_stride = e_stride ? 1 : 0;
        max_count = idx - 1;
    }
    const unsigned nblk = (unsigned)std::max<int64_t>(1, (max_count + kScanBlockElems - 1) / kScanBlockElems);
    k_cubic_sweep<true><<<dim3(nblk, batch), kScanThreads, 0, st>>>(A);
    k_cubic_sweep<false><<<dim3(nblk, batch), kScanThreads, 0, st>>>(A);
    k_cubic_eval<T><<<dim3((unsigned)tiles_of(n), batch), kWave, 0, st>>>(A, 0, n, baseline, b_stride, 0);
    HIP_TRY(e, hipGetLastError());
    if (jobs_out) *jobs_out = jobs;
    if (n_jobs_out) *n_jobs_out = n_jobs;
    return ITD_OK;
}

// knots of x by one of the cubic variant's predicates into d_kidx+1 (d_kidx[0] = 0 in front of them), *idx_out = their
// count (find_extrema: including the leading 0 and the extrapolated tail, like the reference's return value)
int cubic_detect(itd_engine *e, const double *x, int64_t n, int mode, int64_t *count, hipStream_t st)
{
    int rc = scan_level0<double>(e, x, n, mode, true, st, mode == (int)kCpp ? 0 : -1);
    if (rc) return rc;
    if (mode == (int)kZeroCross) k_zero_cross_tail<<<1, 1, 0, st>>>(e->d_kidx, e->d_total);
    return fetch_total(e, st, count);
}

// one signal, synchronous: ONE host synchronisation, at the end (the job: knot count, validity)
int cubic_dev(itd_engine *e, const double *x, int64_t n, const int32_t *extrema, int64_t idx, double *baseline,
              int64_t *idx_out, hipStream_t st, const int32_t **knots_dev_out = nullptr)
{
    if (!e || !x || !baseline) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    if (extrema && (idx < 2 || idx > n - 1)) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    const CubicJob *jobs = nullptr;
    int rc = cubic_batch(e, x, n, 1, n, extrema, 0, idx, baseline, n, st, &jobs, nullptr);
    if (rc) return rc;
    CubicJob job;
    HIP_TRY(e, hipMemcpyAsync(&job, jobs, sizeof(job), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    if (job.status == 1) return ITD_ERR_INVALID_ARG;
    if (job.status == 2) return ITD_ERR_NONFINITE;
    if (idx_out) *idx_out = job.idx;
    if (knots_dev_out) {    // the detected knots (behind the workspace list's leading slot)
        DetectWs w;
        rc = detect_workspace(e, n, 1, w);   // no growth: same geometry as the call above
        if (rc) return rc;
        *knots_dev_out = w.kidx + 1;
    }
    return ITD_OK;
}
}  // namespace

extern "C" {

int itd_baseline_extract_cubic_f64(itd_engine *e, const double *x_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                   double *baseline_dev, int64_t *idx_host, void *stream)
{
    if (!e) return ITD_ERR_INVALID_ARG;
    return cubic_dev(e, x_dev, n, extrema_dev, idx, baseline_dev, idx_host, stream ? (hipStream_t)stream : e->own_stream);
}

int itd_baseline_extract_cubic_f32(itd_engine *e, const float *x_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                   double *baseline_dev, int64_t *idx_host, void *stream)
{
    if (!e || !x_dev) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    k_widen_f32<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x_dev, (double *)e->d_io_x, n);   // float64 arithmetic on the widened signal
    return cubic_dev(e, (const double *)e->d_io_x, n, extrema_dev, idx, baseline_dev, idx_host, st);
}

int itd_baseline_extract_cubic_host_f64(itd_engine *e, const double *x_host, int64_t n, const int64_t *extrema_host,
                                        int64_t idx, double *baseline_host, int64_t *idx_out, int64_t *extrema_out_host)
{
    if (!e || !x_host || !baseline_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)n * sizeof(double) * 2);   // baseline | widened knots
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    const int32_t *ek = nullptr;
    if (extrema_host) {
        if (idx < 2 || idx > n - 1) return ITD_ERR_INVALID_ARG;
        rc = grow(e, &e->d_cub_e, &e->cub_e_bytes, (size_t)(idx + 1) * sizeof(int32_t));
        if (rc) return rc;
        int64_t *d_e64 = (int64_t *)(e->d_io_rows + n);
        HIP_TRY(e, hipMemcpyAsync(d_e64, extrema_host, (size_t)(idx + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        // range check on the host copy first: narrowing to int32 must not wrap
        for (int64_t k = 0; k <= idx; ++k)
            if (extrema_host[k] < 0 || extrema_host[k] >= n) return ITD_ERR_INVALID_ARG;
        k_narrow_idx<<<(unsigned)((idx + 1 + 255) / 256), 256, 0, st>>>(d_e64, e->d_cub_e, idx + 1);
        ek = e->d_cub_e;
    }
    int64_t got = 0;
    const int32_t *knots_dev = nullptr;
    rc = cubic_dev(e, (const double *)e->d_io_x, n, ek, idx, e->d_io_rows, &got, st, extrema_host ? nullptr : &knots_dev);
    if (rc) return rc;
    if (idx_out) *idx_out = got;
    if (got >= 2) HIP_TRY(e, hipMemcpyAsync(baseline_host, e->d_io_rows, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (extrema_out_host && !extrema_host && got > 0) {
        int64_t *d_e64 = (int64_t *)(e->d_io_rows + n);
        k_widen_idx<<<(unsigned)((got + 255) / 256), 256, 0, st>>>(knots_dev, d_e64, got);
        HIP_TRY(e, hipMemcpyAsync(extrema_out_host, d_e64, (size_t)got * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

// The common-baseline form on complex (I/Q) data, itd.cpp:58-154 (itd_detect_fast.hpp: k_detect_fast_iq): knots where both
// components have an extremum, the natural-cubic operator of itd_baseline_extract_cubic_* on the components' mean.
int itd_baseline_extract_iq_f64(itd_engine *e, const double *iq_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                double *baseline_dev, int64_t *idx_host, void *stream)
{
    if (!e || !iq_dev || !baseline_dev) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n || (reinterpret_cast<uintptr_t>(iq_dev) & 15)) return ITD_ERR_INVALID_ARG;
    if (extrema_dev && (idx < 2 || idx > n - 1)) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    int rc = grow(e, &e->d_iq_avg, &e->iq_avg_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    double *avg = (double *)e->d_iq_avg;
    if (extrema_dev) {
        k_iq_mean<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(iq_dev, n, avg);
        return cubic_dev(e, avg, n, extrema_dev, idx, baseline_dev, idx_host, st);
    }
    const int n_tiles = (int)tiles_of(n);
    k_init_state<<<(unsigned)std::min<int64_t>((3 * e->hgsum_third + 255) / 256 + 1, 2048), 256, 0, st>>>(e->d_hstate, 1, e->d_hgsum, 3 * e->hgsum_third, 0);
    unsigned long long *fw = reinterpret_cast<unsigned long long *>(e->d_lists);
    k_detect_fast_iq<<<n_tiles, kWave, 0, st>>>(iq_dev, n, n_tiles, e->d_hcounts, fw, e->d_hgsum, e->d_hstate, avg);
    // the ordered list [lead slot, knots, e[idx] = 0 (itd.cpp's static array at first call)]
    k_compact_fast<<<dim3(n_tiles, 1), kWave, 0, st>>>(fw, e->d_hcounts, e->d_hgsum, n_tiles, n, e->d_kidx, e->max_n + 2, e->d_total, e->d_hstate, 0,
                                                       e->d_hcounts + e->max_tiles, 1);
    int64_t m = 0;
    rc = fetch_total(e, st, &m);
    if (rc) return rc;
    if (idx_host) *idx_host = m;
    if (m < 2) return ITD_OK;                      // itd.cpp:85-87: the caller's buffer is left alone
    return cubic_dev(e, avg, n, e->d_kidx + 1, m, baseline_dev, nullptr, st);
}

int itd_baseline_extract_iq_host_f64(itd_engine *e, const double *iq_host, int64_t n, const int64_t *extrema_host, int64_t idx,
                                     double *baseline_host, int64_t *idx_out, int64_t *extrema_out_host)
{
    if (!e || !iq_host || !baseline_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double) * 2);
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)n * sizeof(double) * 2);   // baseline | widened knots
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, iq_host, (size_t)n * sizeof(double) * 2, hipMemcpyHostToDevice, st));
    const int32_t *ek = nullptr;
    int64_t *d_e64 = (int64_t *)(e->d_io_rows + n);
    if (extrema_host) {
        if (idx < 2 || idx > n - 1) return ITD_ERR_INVALID_ARG;
        for (int64_t k = 0; k <= idx; ++k)
            if (extrema_host[k] < 0 || extrema_host[k] >= n) return ITD_ERR_INVALID_ARG;
        rc = grow(e, &e->d_cub_e, &e->cub_e_bytes, (size_t)(idx + 1) * sizeof(int32_t));
        if (rc) return rc;
        HIP_TRY(e, hipMemcpyAsync(d_e64, extrema_host, (size_t)(idx + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        k_narrow_idx<<<(unsigned)((idx + 1 + 255) / 256), 256, 0, st>>>(d_e64, e->d_cub_e, idx + 1);
        ek = e->d_cub_e;
    }
    int64_t got = extrema_host ? idx : 0;
    rc = itd_baseline_extract_iq_f64(e, (const double *)e->d_io_x, n, ek, idx, e->d_io_rows, extrema_host ? nullptr : &got, st);
    if (rc) return rc;
    if (idx_out) *idx_out = got;
    if (got >= 2) HIP_TRY(e, hipMemcpyAsync(baseline_host, e->d_io_rows, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (extrema_out_host && !extrema_host && got > 0) {
        k_widen_idx<<<(unsigned)((got + 255) / 256), 256, 0, st>>>(e->d_kidx + 1, d_e64, got);
        HIP_TRY(e, hipMemcpyAsync(extrema_out_host, d_e64, (size_t)got * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

int itd_find_extrema_host_f64(itd_engine *e, const double *s_host, int64_t n, int64_t *extrema_host, int64_t *idx_out)
{
    if (!e || !s_host || !extrema_host || !idx_out) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, (size_t)(n + 2) * sizeof(int64_t));
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, s_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    int64_t m = 0;
    rc = cubic_detect(e, (const double *)e->d_io_x, n, (int)kZeroCross, &m, st);
    if (rc) return rc;
    // d_kidx = [0, crossings (m of them), extrapolated tail]: idx = m + 2 entries, the rest of the caller's array is zero
    const int64_t idx = m + 2;
    if (idx > n) return ITD_ERR_INVALID_ARG;   // the reference's own array would overflow (every interior sample a crossing)
    int64_t *d_e64 = (int64_t *)e->d_io_rows;
    k_widen_idx<<<(unsigned)((idx + 255) / 256), 256, 0, st>>>(e->d_kidx, d_e64, idx);
    memset(extrema_host, 0, (size_t)n * sizeof(int64_t));
    HIP_TRY(e, hipMemcpyAsync(extrema_host, d_e64, (size_t)idx * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    *idx_out = idx;
    return ITD_OK;
}

// ---------------------------------------------------------------------------------------------
// The FITPACK flavour of the baseline, batched (itd_spline.hpp): itd_baseline_extract_modified,
// numba_accelerated_itd.py:182-211 (= siftED2D.ipynb cell 1); MEITD.py:303-338 with min_extrema = 0.
// ---------------------------------------------------------------------------------------------
}  // extern "C"
namespace {
struct SplineWs {
    int32_t *lists, *counts, *gsum, *kidx, *totals, *tbase;
    TileRec *recs;
    SigState *state;
    double *a, *c;
    SplineMeta *meta;
    int64_t kidx_stride, lda;
    int n_tiles, n_groups;
};
int spline_workspace(itd_engine *e, int64_t n, int batch, SplineWs &w)
{
    w.n_tiles = (int)tiles_of(n);
    w.n_groups = groups_of(w.n_tiles);
    w.kidx_stride = n + 2;
    w.lda = n + 3;                                   // m <= n + ... data sites: knots <= n - 2, m <= n; 1-based arrays
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t B = (size_t)batch, tiles = B * (size_t)w.n_tiles;
    const size_t b_lists = al(tiles * T * sizeof(int32_t)), b_counts = al(tiles * sizeof(int32_t)), b_recs = al(tiles * sizeof(TileRec));
    const size_t b_gsum = al(B * (size_t)w.n_groups * kGsumPitch * sizeof(int32_t)), b_state = al(B * sizeof(SigState));
    const size_t b_kidx = al(B * (size_t)w.kidx_stride * sizeof(int32_t)), b_tot = al(B * 2 * sizeof(int32_t));
    const size_t b_a = al(B * 4 * (size_t)w.lda * sizeof(double)), b_c = al(B * (size_t)w.lda * sizeof(double)), b_meta = al(B * sizeof(SplineMeta));
    const int rc = grow(e, &e->d_sp, &e->sp_bytes, b_lists + b_counts + b_recs + b_gsum + b_state + b_kidx + b_tot + b_a + b_c + b_meta);
    if (rc) return rc;
    char *p = (char *)e->d_sp;
    w.lists = (int32_t *)p; p += b_lists;
    w.counts = (int32_t *)p; p += b_counts;
    w.recs = (TileRec *)p; p += b_recs;
    w.gsum = (int32_t *)p; p += b_gsum;
    w.state = (SigState *)p; p += b_state;
    w.kidx = (int32_t *)p; p += b_kidx;
    w.totals = (int32_t *)p; p += b_tot;
    w.a = (double *)p; p += b_a;
    w.c = (double *)p; p += b_c;
    w.meta = (SplineMeta *)p;
    return ITD_OK;
}

// baseline (and optionally rotation) of `batch` contiguous-sample signals; all device pointers; asynchronous on st
int spline_enqueue(itd_engine *e, const double *x, int64_t n, int batch, int64_t x_stride, int min_extrema, double *base,
                   int64_t base_stride, double *rot, int64_t rot_stride, hipStream_t st, SplineWs &w)
{
    int rc = spline_workspace(e, n, batch, w);
    if (rc) return rc;
    const dim3 grid_t(w.n_tiles, batch), blk(kWave);
    const int64_t ge = (int64_t)batch * w.n_groups * kGsumPitch;
    k_init_state<<<(unsigned)std::min<int64_t>(std::max<int64_t>((ge + 255) / 256, (batch + 255) / 256), 2048), 256, 0, st>>>(w.state, batch, w.gsum, ge);
    k_detect<double, T><<<grid_t, blk, 0, st>>>(x, x_stride, n, w.n_tiles, (int)kKnots, w.lists, w.counts, w.recs, w.gsum, w.state);
    k_compact<T><<<grid_t, blk, 0, st>>>(w.lists, w.counts, w.gsum, w.n_tiles, n, w.kidx, w.kidx_stride, w.totals, w.state, -1);
    k_spline_fit<<<(batch + 63) / 64, 64, 0, st>>>(x, x_stride, n, batch, w.kidx, w.kidx_stride, w.totals, min_extrema, w.a, w.c, w.lda, w.meta);
    k_spline_eval<<<dim3((unsigned)((n + 255) / 256), batch), 256, 0, st>>>(x, x_stride, n, batch, w.kidx, w.kidx_stride, w.c, w.meta, base,
                                                                            base_stride, rot, rot_stride);
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

// The same operator, parallel in the knots (itd_nak.hpp): the interpolating not-a-knot spline from its second derivatives.
// Asynchronous on st; *totals_out = the device array of {knot count, NaN flag} per signal.
int nak_enqueue(itd_engine *e, const double *x, int64_t n, int batch, int64_t x_stride, int min_extrema, double *base,
                int64_t base_stride, double *rot, int64_t rot_stride, hipStream_t st, const int32_t **totals_out)
{
    DetectWs w;
    int rc = detect_enqueue(e, x, x_stride, n, batch, (int)kKnots, -1, st, w);     // kidx[b] = [0, knots, n-1]
    if (rc) return rc;
    const int64_t L = n + 2;
    const size_t jobs_b = (((size_t)batch * sizeof(CubicJob)) + 255) & ~(size_t)255;
    rc = grow(e, &e->d_cub, &e->cub_bytes, jobs_b + 6 * (size_t)batch * (size_t)L * sizeof(double));
    if (rc) return rc;
    CubicJob *jobs = (CubicJob *)e->d_cub;
    double *arr = (double *)((char *)e->d_cub + jobs_b);
    CubicArgs A;
    A.x = x; A.x_stride = x_stride; A.n = n;
    A.e = w.kidx; A.e_stride = w.kidx_stride;
    A.jobs = jobs; A.job_stride = 1;
    A.K = arr; A.bf = arr + (size_t)batch * L; A.b = arr + 2 * (size_t)batch * L; A.a_stride = L;
    double *cp = arr + 3 * (size_t)batch * L, *sub = arr + 4 * (size_t)batch * L, *rhs = arr + 5 * (size_t)batch * L;
    k_nak_jobs<<<(batch + 255) / 256, 256, 0, st>>>(jobs, batch, w.totals, min_extrema);
    k_nak_values<<<dim3((unsigned)((n + 255) / 256), batch), 256, 0, st>>>(A);
    k_nak_rows<<<dim3((unsigned)((n + 255) / 256), batch), 256, 0, st>>>(A, sub, rhs);
    const unsigned runs = (unsigned)((n + kNakRun - 1) / kNakRun);
    k_nak_forward<<<dim3((runs + 63) / 64, batch), 64, 0, st>>>(A, cp, sub, rhs);
    k_nak_backward<<<dim3((runs + 63) / 64, batch), 64, 0, st>>>(A, cp);
    k_cubic_eval<T, true><<<dim3((unsigned)tiles_of(n), batch), kWave, 0, st>>>(A, 0, n, base, base_stride, 1, rot, rot_stride);
    HIP_TRY(e, hipGetLastError());
    *totals_out = w.totals;
    return ITD_OK;
}

// the engine's 256 bytes of mapped pinned host memory (h_small / its device address d_small); false: not available (the callers
// then copy through their own buffers as before)
constexpr int kSmallWords = 32;           // 8-byte words of mapped host memory an operator's scalars come back in
bool small_results(itd_engine *e)
{
    if (e->h_small) return e->d_small != nullptr;
    if (hipHostMalloc(&e->h_small, kSmallWords * 8, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); e->h_small = nullptr; return false; }
    if (hipHostGetDevicePointer(&e->d_small, e->h_small, 0) != hipSuccess) { (void)hipGetLastError(); e->d_small = nullptr; return false; }
    memset(e->h_small, 0, kSmallWords * 8);
    return true;
}

// Wait for the launch that fills the first `nwords` mapped words: every word carries the call's number in its high half (small_put) —
// whatever order the stores reach the host in, a result is taken only when all of its words are this call's.  The host polls: a few
// microseconds behind the kernel's end instead of the ~25 us hipStreamSynchronize takes to come back from an interrupt on a busy host.
// Falls back to the synchronisation if the words do not turn up within poll_ms (a long kernel, a fault).
int32_t small_next(itd_engine *e) { e->small_seq = e->small_seq == INT32_MAX ? 1 : e->small_seq + 1; return e->small_seq; }
int small_wait(itd_engine *e, int32_t seq, int nwords, hipStream_t st, int poll_ms = 2)
{
    volatile unsigned long long *w = (volatile unsigned long long *)e->h_small;
    auto all_here = [&]() {
        for (int i = 0; i < nwords; ++i) if ((uint32_t)(w[i] >> 32) != (uint32_t)seq) return false;
        return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (int spin = 0; !all_here(); ++spin) {
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(synced ? 2000 : poll_ms)) {
            if (synced) { snprintf(e->err, sizeof(e->err), "the launch's result words did not arrive"); return ITD_ERR_HIP; }
            HIP_TRY(e, hipStreamSynchronize(st));
            synced = true;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return ITD_OK;
}
inline uint32_t small_get(const itd_engine *e, int i) { return (uint32_t)((const volatile unsigned long long *)e->h_small)[i]; }

// ONE signal of at most kNakSmallMax samples through the parallel-in-knots form: one launch, one workgroup (itd_nak.hpp: k_nak_small),
// one 16-byte copy back — knots, NaN flag, validity, and (want_bcount) the knot count of the produced baseline.  Synchronous.
int nak_small(itd_engine *e, const double *x, int64_t n, int min_extrema, double *base, double *rot, hipStream_t st,
              int32_t *knots_host, int32_t *baseline_knots_host)
{
    const int64_t L = n + 2;
    const size_t idx_b = (((size_t)L * sizeof(int32_t)) + 255) & ~(size_t)255, out_b = 256;
    int rc = grow(e, &e->d_cub, &e->cub_bytes, out_b + idx_b + 6 * (size_t)L * sizeof(double));
    if (rc) return rc;
    const bool mapped = small_results(e);
    int32_t *out = (int32_t *)e->d_cub;
    int32_t *idx = (int32_t *)((char *)e->d_cub + out_b);
    double *arr = (double *)((char *)e->d_cub + out_b + idx_b);
    const int32_t seq = mapped ? small_next(e) : 0;
    unsigned long long *words = mapped ? (unsigned long long *)e->d_small : nullptr;
    const size_t lds = 4 * (size_t)L * sizeof(double);
    if (lds <= kNakSmallLdsMax) {                          // the sweeps' arrays in LDS (more than 64 KB of dynamic LDS has to be asked for)
        if (!e->nak_small_attr) {
            HIP_TRY(e, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nak_small<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNakSmallLdsMax));
            e->nak_small_attr = true;
        }
        k_nak_small<true><<<1, kNakSmallThreads, lds, st>>>(x, (int)n, min_extrema, idx, arr, arr + L, arr + 2 * L, arr + 3 * L, arr + 4 * L, arr + 5 * L,
                                                             base, rot, baseline_knots_host ? 1 : 0, out, words, seq);
    } else
        k_nak_small<false><<<1, kNakSmallThreads, 0, st>>>(x, (int)n, min_extrema, idx, arr, arr + L, arr + 2 * L, arr + 3 * L, arr + 4 * L, arr + 5 * L,
                                                            base, rot, baseline_knots_host ? 1 : 0, out, words, seq);
    int32_t h[4] = {0, 0, 0, 0};
    HIP_TRY(e, hipGetLastError());
    if (mapped) {
        const int rc2 = small_wait(e, seq, 4, st);
        if (rc2) return rc2;
        for (int q = 0; q < 4; ++q) h[q] = (int32_t)small_get(e, q);
    } else {
        HIP_TRY(e, hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
    }
    if (knots_host) knots_host[0] = h[0];
    if (baseline_knots_host) baseline_knots_host[0] = h[3];
    return h[1] ? ITD_ERR_NONFINITE : ITD_OK;
}

// after spline_enqueue / nak_enqueue: synchronise, fetch the per-signal knot counts, report NaN input
int spline_finish(itd_engine *e, int batch, const int32_t *totals, int32_t *knots_host, hipStream_t st)
{
    std::vector<int32_t> tot((size_t)batch * 2);
    HIP_TRY(e, hipMemcpyAsync(tot.data(), totals, tot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    bool nan_in = false;
    for (int b = 0; b < batch; ++b) {
        if (knots_host) knots_host[b] = tot[2 * (size_t)b];
        nan_in = nan_in || tot[2 * (size_t)b + 1] != 0;
    }
    return nan_in ? ITD_ERR_NONFINITE : ITD_OK;
}
}  // namespace
extern "C" {

int itd_baseline_extract_spline_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                                    int32_t min_extrema, double *baseline_dev, int64_t baseline_stride, double *rot_dev,
                                    int64_t rot_stride, int32_t *knots_host, void *stream)
{
    if (!e || !x_dev || !baseline_dev) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n >= (int64_t)INT32_MAX - 8 || batch < 1 || batch > 65535 /* grid.y of its launches */ || min_extrema < 0) return ITD_ERR_INVALID_ARG;
    if (batch > 1 && (x_stride < n || baseline_stride < n || (rot_dev && rot_stride < n))) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    // few long signals: parallel in the knots; many short rows: one thread per signal, FITPACK's own sweep (bit-level)
    const bool par = e->spline_solver == ITD_SPLINE_PARALLEL || (e->spline_solver == ITD_SPLINE_AUTO && batch < 256 && n >= 1024);
    if (par && batch == 1 && n <= kNakSmallMax) return nak_small(e, x_dev, n, min_extrema, baseline_dev, rot_dev, st, knots_host, nullptr);
    const int32_t *totals = nullptr;
    SplineWs w;
    int rc;
    if (par) rc = nak_enqueue(e, x_dev, n, batch, x_stride, min_extrema, baseline_dev, baseline_stride, rot_dev, rot_stride, st, &totals);
    else {
        rc = spline_enqueue(e, x_dev, n, batch, x_stride, min_extrema, baseline_dev, baseline_stride, rot_dev, rot_stride, st, w);
        totals = w.totals;
    }
    if (rc) return rc;
    return spline_finish(e, batch, totals, knots_host, st);
}

int itd_set_spline_solver(itd_engine *e, int32_t solver)
{
    if (!e || solver < ITD_SPLINE_AUTO || solver > ITD_SPLINE_PARALLEL) return ITD_ERR_INVALID_ARG;
    e->spline_solver = solver;
    return ITD_OK;
}

int itd_baseline_extract_spline_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t min_extrema,
                                         double *baseline_host, double *rot_host, int32_t *knots_host)
{
    return itd_baseline_extract_spline_host2_f64(e, x_host, n, batch, min_extrema, baseline_host, rot_host, knots_host, nullptr);
}

// the same, plus (optionally) the knot count of every PRODUCED baseline — what MEITD's loops ask for right after an
// extraction (MEITD.py:362-363, :497-505: `count = matlab_detect_peaks(baseline).size + ...`): one more launch on the data
// already on the device instead of another upload / list download
int itd_baseline_extract_spline_host2_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t min_extrema,
                                          double *baseline_host, double *rot_host, int32_t *knots_host, int32_t *baseline_knots_host)
{
    if (!e || !x_host || !baseline_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || batch < 1) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    const size_t cnt = (size_t)n * (size_t)batch;
    int rc = grow(e, &e->d_sp2, &e->sp2_bytes, 3 * cnt * sizeof(double));
    if (rc) return rc;
    double *d_x = e->d_sp2, *d_b = d_x + cnt, *d_r = d_b + cnt;
    HIP_TRY(e, hipMemcpyAsync(d_x, x_host, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    rc = itd_baseline_extract_spline_f64(e, d_x, n, batch, n, min_extrema, d_b, n, rot_host ? d_r : nullptr, n, knots_host, st);
    if (rc) return rc;
    std::vector<int32_t> tot;
    if (baseline_knots_host) {
        if (batch > kMaxGridY) return ITD_ERR_INVALID_ARG;
        DetectWs w;
        rc = detect_enqueue(e, d_b, n, n, batch, (int)kKnots, -1, st, w, nullptr, 0, false);   // counts only
        if (rc) return rc;
        tot.resize(2 * (size_t)batch);
        HIP_TRY(e, hipMemcpyAsync(tot.data(), w.totals, tot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(e, hipMemcpyAsync(baseline_host, d_b, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    if (rot_host) HIP_TRY(e, hipMemcpyAsync(rot_host, d_r, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    if (baseline_knots_host) for (int b = 0; b < batch; ++b) baseline_knots_host[b] = tot[2 * (size_t)b];
    return ITD_OK;
}

// knot counts of host signals without any index list (matlab_detect_peaks(x).size + matlab_detect_peaks(-x).size for mode
// ITD_DETECT_KNOTS, MEITD.py:350, :376, :409): upload, one counting launch pair, 4 bytes per signal back
int itd_count_knots_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t mode, int32_t *counts_host)
{
    if (!e || !x_host || !counts_host || n < 3 || batch < 1 || batch > kMaxGridY || mode < 0 || mode > 4) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    const size_t cnt = (size_t)n * (size_t)batch;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, cnt * sizeof(double));
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, x_host, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    DetectWs w;
    rc = detect_enqueue(e, (const double *)e->d_io_x, n, n, batch, mode, -1, st, w, nullptr, 0, false);
    if (rc) return rc;
    std::vector<int32_t> tot(2 * (size_t)batch);
    HIP_TRY(e, hipMemcpyAsync(tot.data(), w.totals, tot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    bool nan_in = false;
    for (int b = 0; b < batch; ++b) { counts_host[b] = tot[2 * (size_t)b]; nan_in = nan_in || tot[2 * (size_t)b + 1]; }
    return nan_in ? ITD_ERR_NONFINITE : ITD_OK;     // counted under the plain rules: see itd_detect_* for detect_peaks' NaN branch
}

// ---- MEITD's operators on device-resident signals (MEITD.py:344-534 keeps one signal and its rotations / baselines in a loop:
//      nothing but a few scalars has to cross PCIe per pass) -----------------------------------------------------------------
int itd_count_knots_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride, int32_t mode,
                        int32_t *counts_host, void *stream)
{
    if (!e || !x_dev || !counts_host || n < 3 || batch < 1 || batch > kMaxGridY || mode < 0 || mode > 4) return ITD_ERR_INVALID_ARG;
    if (batch > 1 && x_stride < n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    DetectWs w;
    int rc = detect_enqueue(e, x_dev, x_stride, n, batch, mode, -1, st, w, nullptr, 0, false);
    if (rc) return rc;
    std::vector<int32_t> tot(2 * (size_t)batch);
    const bool mapped = 2 * batch <= kSmallWords && small_results(e);       // a few counts: the GPU copies them into the mapped words itself
    if (mapped) {
        const int32_t seq = small_next(e);
        k_copy_words<<<1, 64, 0, st>>>(w.totals, (unsigned long long *)e->d_small, 2 * batch, seq);
        HIP_TRY(e, hipGetLastError());
        rc = small_wait(e, seq, 2 * batch, st);
        if (rc) return rc;
        for (int i = 0; i < 2 * batch; ++i) tot[(size_t)i] = (int32_t)small_get(e, i);
    } else {
        HIP_TRY(e, hipMemcpyAsync(tot.data(), w.totals, tot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(e, hipStreamSynchronize(st));
    }
    bool nan_in = false;
    for (int b = 0; b < batch; ++b) { counts_host[b] = tot[2 * (size_t)b]; nan_in = nan_in || tot[2 * (size_t)b + 1]; }
    return nan_in ? ITD_ERR_NONFINITE : ITD_OK;
}

// the weighted sums and window counts of the six order-3 permutation patterns of x (itd_wpe.hpp; MEITD.py:79-128), and —
// optionally, in the same synchronisation — x's knot count: MEITD.py:346-351 and :373-378 ask for both of the same signal
int itd_wpe3_f64(itd_engine *e, const double *x_dev, int64_t n, double *bin_weights_host, int64_t *bin_windows_host,
                 int32_t *knots_host, void *stream)
{
    if (!e || !x_dev || !bin_weights_host || !bin_windows_host || n < 3) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    const int64_t nw = n - 2;
    const bool exact = nw <= kWpeExactWindows;
    const int64_t seg_len = exact ? nw : kWpeSeg;
    const int64_t nseg = (nw + seg_len - 1) / seg_len;
    if (nseg > INT32_MAX / 8) return ITD_ERR_INVALID_ARG;
    const size_t out_off = (size_t)nseg * (6 * (sizeof(double) + sizeof(long long)) + 2 * sizeof(int));    // the segments' sums, then the totals
    int rc = grow(e, &e->d_wpe, &e->wpe_bytes, out_off + 6 * (sizeof(double) + sizeof(long long)) + 2 * sizeof(int));
    if (rc) return rc;
    double *part_s = reinterpret_cast<double *>(e->d_wpe);
    long long *part_c = reinterpret_cast<long long *>(part_s + (size_t)nseg * 6);
    int *part_k = reinterpret_cast<int *>(part_c + (size_t)nseg * 6);
    double *out_s = reinterpret_cast<double *>(e->d_wpe + out_off);
    long long *out_c = reinterpret_cast<long long *>(out_s + 6);
    int *out_k = reinterpret_cast<int *>(out_c + 6);
    // (one segment — MEITD's signals —: the sums land in the engine's mapped host words, no copy behind the launch)
    const bool mapped = nseg == 1 && small_results(e);
    // (the knot count of x rides along: a window's middle sample is a knot or not — MEITD.py:346-351, :373-378 ask for both)
    const int32_t seq = mapped ? small_next(e) : 0;
    k_wpe3<<<(unsigned)nseg, kWpeThreads, 0, st>>>(x_dev, nw, seg_len, part_s, part_c, knots_host ? part_k : nullptr,
                                                   mapped ? (unsigned long long *)e->d_small : nullptr, seq);
    if (nseg > 1) k_wpe3_combine<<<1, 64, 0, st>>>(part_s, part_c, (int)nseg, out_s, out_c, knots_host ? part_k : nullptr, out_k);
    HIP_TRY(e, hipGetLastError());
    struct { double s[6]; long long c[6]; int k[2]; } res;
    res.k[0] = res.k[1] = 0;
    const size_t res_b = 6 * (sizeof(double) + sizeof(long long)) + (knots_host ? 2 * sizeof(int) : 0);
    if (nseg > 1) HIP_TRY(e, hipMemcpyAsync(&res, out_s, res_b, hipMemcpyDeviceToHost, st));
    else if (!mapped) HIP_TRY(e, hipMemcpyAsync(&res, part_s, res_b, hipMemcpyDeviceToHost, st));
    if (mapped) {
        const int rc2 = small_wait(e, seq, 20, st);
        if (rc2) return rc2;
        for (int b = 0; b < 6; ++b) {
            const unsigned long long bits = (unsigned long long)small_get(e, 2 * b) | ((unsigned long long)small_get(e, 2 * b + 1) << 32);
            memcpy(&res.s[b], &bits, sizeof(double));
            res.c[b] = (long long)small_get(e, 12 + b);
        }
        res.k[0] = (int)small_get(e, 18); res.k[1] = (int)small_get(e, 19);
    } else HIP_TRY(e, hipStreamSynchronize(st));
    for (int b = 0; b < 6; ++b) { bin_weights_host[b] = res.s[b]; bin_windows_host[b] = (int64_t)res.c[b]; }
    if (knots_host) *knots_host = res.k[0];
    return knots_host && res.k[1] ? ITD_ERR_NONFINITE : ITD_OK;
}

// MEITD's whole selection loop on one short device-resident signal as one launch (itd_meitd.hpp: k_meitd_small).  Synchronous.
int itd_meitd_small_f64(itd_engine *e, double *rows_dev, int64_t n, double wpemax, int32_t *result_host, void *probe_log_host,
                        int32_t log_cap, void *stream)
{
    if (!e || !rows_dev || !result_host || n < 3 || n > kNakSmallMax || log_cap < 0 || (log_cap > 0 && !probe_log_host)) return ITD_ERR_INVALID_ARG;
    // (only where the host-driven loop's extractions take the same operator: the parallel-in-knots form of one signal)
    if (!(e->spline_solver == ITD_SPLINE_PARALLEL || (e->spline_solver == ITD_SPLINE_AUTO && n >= 1024))) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    const int64_t L = n + 2;
    const size_t idx_b = (((size_t)L * sizeof(int32_t)) + 255) & ~(size_t)255, out_b = 256;
    const size_t arr_b = 6 * (size_t)L * sizeof(double);
    int rc = grow(e, &e->d_cub, &e->cub_bytes, out_b + idx_b + arr_b + kMeitdLogCap * sizeof(MeitdProbe));
    if (rc) return rc;
    int32_t *idx = (int32_t *)((char *)e->d_cub + out_b);
    double *arr = (double *)((char *)e->d_cub + out_b + idx_b);
    MeitdProbe *dlog = (MeitdProbe *)((char *)e->d_cub + out_b + idx_b + arr_b);
    MeitdOut *dout = (MeitdOut *)e->d_cub;                 // (the first 256 bytes of the workspace)
    const size_t nak_lds = 4 * (size_t)L * sizeof(double);
    const bool in_lds = nak_lds <= kNakSmallLdsMax;
    const size_t lds = in_lds && nak_lds > kMeitdWpeLds ? nak_lds : kMeitdWpeLds;   // the two operators take turns in the same bytes
    if (!e->meitd_attr[in_lds]) {
        const void *fn = in_lds ? reinterpret_cast<const void *>(&k_meitd_small<true>) : reinterpret_cast<const void *>(&k_meitd_small<false>);
        HIP_TRY(e, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(in_lds ? kNakSmallLdsMax : kMeitdWpeLds)));
        e->meitd_attr[in_lds] = true;
    }
    if (in_lds)
        k_meitd_small<true><<<1, kNakSmallThreads, lds, st>>>(rows_dev, (int)n, wpemax, idx, arr, arr + L, arr + 2 * L, arr + 3 * L, arr + 4 * L, arr + 5 * L, dlog, dout);
    else
        k_meitd_small<false><<<1, kNakSmallThreads, lds, st>>>(rows_dev, (int)n, wpemax, idx, arr, arr + L, arr + 2 * L, arr + 3 * L, arr + 4 * L, arr + 5 * L, dlog, dout);
    HIP_TRY(e, hipGetLastError());
    // (a whole loop: milliseconds — the header and the log are plain copies behind it)
    MeitdOut ho;
    HIP_TRY(e, hipMemcpyAsync(&ho, dout, sizeof(ho), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    memcpy(result_host, &ho, sizeof(MeitdOut));
    const int32_t got = result_host[4] < log_cap ? result_host[4] : log_cap;
    if (got > 0) HIP_TRY(e, hipMemcpy(probe_log_host, dlog, (size_t)(got < kMeitdLogCap ? got : kMeitdLogCap) * sizeof(MeitdProbe), hipMemcpyDeviceToHost));
    return ITD_OK;
}

// weighted_permutation_entropy's pass over the samples for any order 2 .. 5: sums_host / windows_host [order^order], indexed by the
// reference's hash value (a hash without windows is absent from its list)
int itd_wpe_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t order, double *sums_host, int64_t *windows_host, void *stream)
{
    if (!e || !x_dev || !sums_host || !windows_host || order < 2 || order > kWpeMaxOrder || n < order) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    const int64_t nw = n - order + 1;
    int nh = 1;
    for (int k = 0; k < order; ++k) nh *= order;
    const int64_t seg_len = nw <= kWpeExactWindows ? nw : kWpeSeg * 4;
    const int64_t nseg = (nw + seg_len - 1) / seg_len;
    if (nseg > 65535) return ITD_ERR_INVALID_ARG;
    const size_t wts_off = ((size_t)nw * sizeof(unsigned short) + 255) / 256 * 256;
    const size_t part_off = wts_off + (size_t)nw * sizeof(double);
    const size_t out_off = part_off + (size_t)nseg * nh * (sizeof(double) + sizeof(long long));
    int rc = grow(e, &e->d_wpe, &e->wpe_bytes, out_off + (size_t)nh * (sizeof(double) + sizeof(long long)));
    if (rc) return rc;
    unsigned short *hashes = reinterpret_cast<unsigned short *>(e->d_wpe);
    double *wts = reinterpret_cast<double *>(e->d_wpe + wts_off);
    double *part_s = reinterpret_cast<double *>(e->d_wpe + part_off);
    long long *part_c = reinterpret_cast<long long *>(part_s + (size_t)nseg * nh);
    double *out_s = reinterpret_cast<double *>(e->d_wpe + out_off);
    long long *out_c = reinterpret_cast<long long *>(out_s + nh);
    const unsigned gb = (unsigned)((nw + 255) / 256);
    switch (order) {
    case 2: k_wpe_eval<2><<<gb, 256, 0, st>>>(x_dev, nw, hashes, wts); break;
    case 3: k_wpe_eval<3><<<gb, 256, 0, st>>>(x_dev, nw, hashes, wts); break;
    case 4: k_wpe_eval<4><<<gb, 256, 0, st>>>(x_dev, nw, hashes, wts); break;
    default: k_wpe_eval<5><<<gb, 256, 0, st>>>(x_dev, nw, hashes, wts); break;
    }
    k_wpe_sum<<<dim3((unsigned)((nh + 255) / 256), (unsigned)nseg), 256, 0, st>>>(hashes, wts, nw, seg_len, nh, part_s, part_c);
    if (nseg > 1) k_wpe_combine<<<(unsigned)((nh + 255) / 256), 256, 0, st>>>(part_s, part_c, (int)nseg, nh, out_s, out_c);
    HIP_TRY(e, hipGetLastError());
    std::vector<long long> cnt((size_t)nh);
    HIP_TRY(e, hipMemcpyAsync(sums_host, nseg > 1 ? out_s : part_s, (size_t)nh * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipMemcpyAsync(cnt.data(), nseg > 1 ? out_c : part_c, (size_t)nh * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    for (int h = 0; h < nh; ++h) windows_host[h] = (int64_t)cnt[(size_t)h];
    return ITD_OK;
}

// itd_baseline_extract_spline_f64 plus the knot count of every PRODUCED baseline, one synchronisation for both
// (MEITD.py:362-363, :497-505)
int itd_baseline_extract_spline2_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                                     int32_t min_extrema, double *baseline_dev, int64_t baseline_stride, double *rot_dev,
                                     int64_t rot_stride, int32_t *knots_host, int32_t *baseline_knots_host, void *stream)
{
    if (!e || !x_dev || !baseline_dev) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n >= (int64_t)INT32_MAX - 8 || batch < 1 || batch > 65535 || min_extrema < 0) return ITD_ERR_INVALID_ARG;
    if (batch > 1 && (x_stride < n || baseline_stride < n || (rot_dev && rot_stride < n))) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    const bool par = e->spline_solver == ITD_SPLINE_PARALLEL || (e->spline_solver == ITD_SPLINE_AUTO && batch < 256 && n >= 1024);
    if (par && batch == 1 && n <= kNakSmallMax) return nak_small(e, x_dev, n, min_extrema, baseline_dev, rot_dev, st, knots_host, baseline_knots_host);
    const int32_t *totals = nullptr;
    SplineWs w;
    int rc;
    if (par) rc = nak_enqueue(e, x_dev, n, batch, x_stride, min_extrema, baseline_dev, baseline_stride, rot_dev, rot_stride, st, &totals);
    else {
        rc = spline_enqueue(e, x_dev, n, batch, x_stride, min_extrema, baseline_dev, baseline_stride, rot_dev, rot_stride, st, w);
        totals = w.totals;
    }
    if (rc) return rc;
    if (!baseline_knots_host) return spline_finish(e, batch, totals, knots_host, st);
    // (the extraction's totals are read before the counting launches reuse the detection workspace)
    std::vector<int32_t> tot((size_t)batch * 2), btot((size_t)batch * 2);
    HIP_TRY(e, hipMemcpyAsync(tot.data(), totals, tot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    DetectWs dw;
    rc = detect_enqueue(e, baseline_dev, baseline_stride, n, batch, (int)kKnots, -1, st, dw, nullptr, 0, false);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(btot.data(), dw.totals, btot.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    bool nan_in = false;
    for (int b = 0; b < batch; ++b) {
        if (knots_host) knots_host[b] = tot[2 * (size_t)b];
        baseline_knots_host[b] = btot[2 * (size_t)b];
        nan_in = nan_in || tot[2 * (size_t)b + 1] != 0;
    }
    return nan_in ? ITD_ERR_NONFINITE : ITD_OK;
}

int itd_subtract_f64(itd_engine *e, const double *a_dev, const double *b_dev, double *out_dev, int64_t count, void *stream)
{
    if (!e || !a_dev || !b_dev || !out_dev || count < 0) return ITD_ERR_INVALID_ARG;
    if (!count) return ITD_OK;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    k_subtract<<<(unsigned)((count + 255) / 256), 256, 0, st>>>(a_dev, b_dev, out_dev, count);
    HIP_TRY(e, hipGetLastError());
    return ITD_OK;
}

// copies ordered on the engine's stream (or `stream`): kind 0 device -> host, 1 host -> device, 2 device -> device,
// 3 zero bytes (src ignored); wait != 0: return when it is done
int itd_copy(itd_engine *e, void *dst, const void *src, int64_t bytes, int32_t kind, int32_t wait, void *stream)
{
    if (!e || !dst || (!src && kind != 3) || bytes < 0 || kind < 0 || kind > 3) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    if (bytes) {
        if (kind == 3) HIP_TRY(e, hipMemsetAsync(dst, 0, (size_t)bytes, st));
        else HIP_TRY(e, hipMemcpyAsync(dst, src, (size_t)bytes, kind == 0 ? hipMemcpyDeviceToHost : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice), st));
    }
    if (wait) HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

// crossways_itd_baseline_extract(data), siftED2D.ipynb cell 1, for `planes` images of rows x cols (device, contiguous):
// rows then columns of the rows' result, columns then rows of the columns' result, averaged.
int itd_crossways_f64(itd_engine *e, const double *img_dev, int32_t planes, int32_t rows, int32_t cols, int32_t min_extrema,
                      double *out_dev, void *stream)
{
    if (!e || !img_dev || !out_dev || planes < 1 || rows < 3 || cols < 3) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    const size_t cnt = (size_t)planes * rows * cols;
    int rc = grow(e, &e->d_sp2, &e->sp2_bytes, 3 * cnt * sizeof(double));
    if (rc) return rc;
    double *A = e->d_sp2, *Bq = A + cnt, *C = Bq + cnt;       // scratch planes
    SplineWs w;
    auto tr = [&](const double *in, int r, int c, double *out) {
        k_transpose<<<dim3((c + 31) / 32, (r + 31) / 32, planes), 256, 0, st>>>(in, r, c, out);
    };
    HIP_TRY(e, hipMemsetAsync(e->d_flag, 0, sizeof(int32_t), st));
    // every row of `sigs` x `len`, in chunks of at most 65535 signals (they are the launches' grid.y: 20 ensemble planes of a
    // 3840 x 2160 image are 76 800 rows per stage); the NaN-input flags of every chunk of every stage are OR-ed into d_flag
    auto ext = [&](const double *in, int64_t sigs, int len, double *out) {
        for (int64_t s0 = 0; s0 < sigs; s0 += kMaxGridY) {
            const int nb = (int)std::min<int64_t>(kMaxGridY, sigs - s0);
            const int rc2 = spline_enqueue(e, in + s0 * len, len, nb, len, min_extrema, out + s0 * len, len, nullptr, 0, st, w);
            if (rc2) return rc2;
            k_or_nan_flags<<<(nb + 255) / 256, 256, 0, st>>>(w.totals, nb, e->d_flag);
        }
        return (int)ITD_OK;
    };
    // lengthwise = rows(data); then its columns
    if ((rc = ext(img_dev, (int64_t)planes * rows, cols, A))) return rc;    // A = lengthwise (rows done)
    tr(A, rows, cols, Bq);                                                  // Bq = lengthwise^T  [cols][rows]
    if ((rc = ext(Bq, (int64_t)planes * cols, rows, A))) return rc;         // A = columns of lengthwise, transposed layout
    tr(A, cols, rows, C);                                                   // C = lengthwise, final [rows][cols]
    // crosswise = columns(data); then its rows
    tr(img_dev, rows, cols, A);                                             // A = data^T
    if ((rc = ext(A, (int64_t)planes * cols, rows, Bq))) return rc;         // Bq = columns of data (transposed layout)
    tr(Bq, cols, rows, A);                                                  // A = crosswise [rows][cols]
    if ((rc = ext(A, (int64_t)planes * rows, cols, Bq))) return rc;         // Bq = rows of crosswise
    k_mean2<<<(unsigned)((cnt + 255) / 256), 256, 0, st>>>(C, Bq, (int64_t)cnt, out_dev);
    HIP_TRY(e, hipGetLastError());
    int32_t nan_in = 0;     // a NaN in any row of any of the four stages
    HIP_TRY(e, hipMemcpyAsync(&nan_in, e->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    return nan_in ? ITD_ERR_NONFINITE : ITD_OK;
}

int itd_crossways_host_f64(itd_engine *e, const double *img_host, int32_t planes, int32_t rows, int32_t cols, int32_t min_extrema,
                           double *out_host)
{
    if (!e || !img_host || !out_host || planes < 1 || rows < 3 || cols < 3) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    const size_t cnt = (size_t)planes * rows * cols;
    int rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, 2 * cnt * sizeof(double));
    if (rc) return rc;
    double *d_in = e->d_io_rows, *d_out = d_in + cnt;
    HIP_TRY(e, hipMemcpyAsync(d_in, img_host, cnt * sizeof(double), hipMemcpyHostToDevice, st));
    rc = itd_crossways_f64(e, d_in, planes, rows, cols, min_extrema, d_out, st);
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(out_host, d_out, cnt * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

// ---------------------------------------------------------------------------------------------
// Instantaneous amplitude / phase / frequency of a proper rotation (itd_tfe.hpp; README.md:13-21, 41-55).
// ---------------------------------------------------------------------------------------------
int itd_instantaneous_f64(itd_engine *e, const double *rot_dev, int64_t n, double *amp_dev, double *phase_dev, double *freq_dev,
                          void *stream)
{
    if (!e || !rot_dev || (!amp_dev && !phase_dev && !freq_dev)) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
    int64_t m = 0;
    int rc = cubic_detect(e, rot_dev, n, (int)kZeroCross, &m, st);   // ordered zero crossings in d_kidx[1..m]
    if (rc) return rc;
    rc = grow(e, &e->d_cub, &e->cub_bytes, (size_t)(m + 2) * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long *amp_bits = (unsigned long long *)e->d_cub;
    HIP_TRY(e, hipMemsetAsync(amp_bits, 0, (size_t)(m + 1) * sizeof(unsigned long long), st));
    static_assert(kTfeTile == T, "k_compact's per-tile bases are per T samples");
    const unsigned blocks = (unsigned)tiles_of(n);
    const int32_t *tile_base = e->d_hcounts + e->max_tiles;   // crossings in front of every tile (k_compact)
    k_tfe_amplitude<<<blocks, 64, 0, st>>>(rot_dev, n, tile_base, amp_bits);
    k_tfe_phase<<<blocks, 64, 0, st>>>(rot_dev, n, tile_base, amp_bits, amp_dev, phase_dev, freq_dev);
    HIP_TRY(e, hipGetLastError());
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

int itd_instantaneous_host_f64(itd_engine *e, const double *rot_host, int64_t n, double *amp_host, double *phase_host,
                               double *freq_host)
{
    if (!e || !rot_host) return ITD_ERR_INVALID_ARG;
    if (n < 3 || n > e->max_n) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    hipStream_t st = e->own_stream;
    int rc = grow(e, &e->d_io_x, &e->io_x_bytes, (size_t)n * sizeof(double));
    if (rc) return rc;
    rc = grow(e, &e->d_io_rows, &e->io_rows_bytes, 3 * (size_t)n * sizeof(double));
    if (rc) return rc;
    HIP_TRY(e, hipMemcpyAsync(e->d_io_x, rot_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    double *d_a = e->d_io_rows, *d_p = d_a + n, *d_f = d_p + n;
    rc = itd_instantaneous_f64(e, (const double *)e->d_io_x, n, d_a, d_p, d_f, st);
    if (rc) return rc;
    if (amp_host) HIP_TRY(e, hipMemcpyAsync(amp_host, d_a, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (phase_host) HIP_TRY(e, hipMemcpyAsync(phase_host, d_p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    if (freq_host) HIP_TRY(e, hipMemcpyAsync(freq_host, d_f, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    return ITD_OK;
}

int itd_set_kernel_timing(itd_engine *e, int max_decompositions)
{
    if (!e || max_decompositions < 0 || max_decompositions > 4096) return ITD_ERR_INVALID_ARG;
    DevGuard g(e->device);
    e->timing = max_decompositions > 0;
    e->timing_seq = 0;
    e->n_timed = 0;
    e->timing_overflow = false;
    const size_t want = 2 * (size_t)max_decompositions * (ITD_MAX_ROWS + 3);
    while (e->ev.size() < want) {
        hipEvent_t ev = nullptr;
        HIP_TRY(e, hipEventCreate(&ev));
        e->ev.push_back(ev);
    }
    e->ev_tag.resize(e->ev.size() / 2, 0);
    e->ev_from.resize(e->ev.size() / 2, 0);
    e->ev_to.resize(e->ev.size() / 2, 0);
    return ITD_OK;
}

int itd_set_kernel_timing_stride(itd_engine *e, int stride)
{
    if (!e || stride < 1) return ITD_ERR_INVALID_ARG;
    e->timing_stride = stride;
    return ITD_OK;
}

int itd_set_kernel_timing_mode(itd_engine *e, int32_t mode)
{
    if (!e || mode < 0 || mode > 1) return ITD_ERR_INVALID_ARG;
    e->timing_mode = mode;
    return ITD_OK;
}

int itd_get_kernel_timing_samples(itd_engine *e, int32_t which, double *ms_out, int32_t cap, int32_t *count)
{
    if (!e || which < 0 || which > ITD_TIME_KF_KNOTS || cap < 0 || (cap > 0 && !ms_out)) return ITD_ERR_INVALID_ARG;
    if (!e->ran || !e->timing) return ITD_ERR_NOT_RUN;
    DevGuard g(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->last_stream));
    int cnt = 0;
    for (int k = 0; k < e->n_timed; ++k) {
        if (e->ev_tag[(size_t)k] != which) continue;
        if (cnt < cap) {
            float ms = 0.f;
            HIP_TRY(e, hipEventElapsedTime(&ms, e->ev[(size_t)e->ev_from[(size_t)k]], e->ev[(size_t)e->ev_to[(size_t)k]]));
            ms_out[cnt] = ms;
        }
        ++cnt;
    }
    if (count) *count = cnt;
    return ITD_OK;
}

int itd_get_step_periods(itd_engine *e, double *ms_out, int32_t cap, int32_t *count)
{
    if (!e || cap < 0 || (cap > 0 && !ms_out)) return ITD_ERR_INVALID_ARG;
    if (!e->ran || !e->timing) return ITD_ERR_NOT_RUN;
    DevGuard g(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->last_stream));
    int cnt = 0, prev = -1;
    for (int k = 0; k < e->n_timed; ++k) {
        if (e->ev_tag[(size_t)k] != ITD_TIME_EXTRACT_L0) continue;
        if (prev >= 0) {
            if (cnt < cap) {
                float ms = 0.f;
                HIP_TRY(e, hipEventElapsedTime(&ms, e->ev[(size_t)e->ev_from[(size_t)prev]], e->ev[(size_t)e->ev_from[(size_t)k]]));
                ms_out[cnt] = ms;
            }
            ++cnt;
        }
        prev = k;
    }
    if (count) *count = cnt;
    return ITD_OK;
}

int itd_get_kernel_timing(itd_engine *e, int32_t which, double *ms_total, int32_t *launches)
{
    if (!e || which < 0 || which > ITD_TIME_KF_KNOTS) return ITD_ERR_INVALID_ARG;
    if (!e->ran || !e->timing) return ITD_ERR_NOT_RUN;
    DevGuard g(e->device);
    HIP_TRY(e, hipStreamSynchronize(e->last_stream));
    double tot = 0.0;
    int cnt = 0;
    for (int k = 0; k < e->n_timed; ++k) {
        if (e->ev_tag[(size_t)k] != which) continue;
        float ms = 0.f;
        HIP_TRY(e, hipEventElapsedTime(&ms, e->ev[(size_t)e->ev_from[(size_t)k]], e->ev[(size_t)e->ev_to[(size_t)k]]));
        tot += ms;
        ++cnt;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = cnt;
    return ITD_OK;
}

}  // extern "C"

#include "itd_engine_batch.inc"
