/*
 * pyitd_hip.h — C ABI of the MI355X (gfx950) ITD engine, libpyitd_hip.so.
 *
 * This is the drop-in boundary for ONE path of falseywinchnet/PyITD:
 *     itd(x, max_iteration) -> rotations[rows, N] (+ baselines)
 * i.e. the reference's   ITD.itd            ITD.py:351-432  (runnable form: PyITD.ipynb cell 1;
 *                                           free-function form ITD_numba.py:100-136)
 *                        itd_baseline_extract  ITD.py:79-121
 *                        detect_peaks          ITD.py:33-76  (twin matlab_detect_peaks,
 *                                              numba_accelerated_itd.py:17-59)
 *                        baseline_knot_estimation  numba_accelerated_itd.py:167-178 (= ITD.py:100-110)
 * The reference is pure Python/numba with no FFI of its own; each entry point below names the
 * reference function a binding would replace.  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C types only; every function returns an itd_status (0 = ok) and never throws.
 *   - "_dev" pointers are device (HBM) pointers on the engine's GPU; "_host" pointers are host memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the engine's own stream).
 *   - all floating-point results are IEEE binary64 computed in the reference's association order
 *     (no FMA contraction), so they are bit-identical to the reference for finite data.
 *   - int32 knot indices on the device (N < 2^31); the host-facing entry points widen to int64
 *     like the reference's numpy.int64 arrays.
 */
#ifndef PYITD_HIP_H
#define PYITD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITD_ABI_VERSION 11

/* rotations/baselines hold at most 22 rows in the reference (ITD.py:384-385): max_iteration <= 20 */
#define ITD_MAX_ROWS 22
#define ITD_MAX_ITERATION 20

typedef enum itd_status {
    ITD_OK = 0,
    ITD_ERR_INVALID_ARG = 1,   /* NULL pointer, N < 3, batch < 1, max_iteration outside [0, 20], sizes beyond the engine's */
    ITD_ERR_NO_DEVICE = 2,     /* no HIP device / wrong device id */
    ITD_ERR_HIP = 3,           /* a HIP runtime call failed: see itd_last_error() */
    ITD_ERR_NOMEM = 4,         /* device or host allocation failed */
    ITD_ERR_NOT_RUN = 5,       /* results requested before a decomposition was enqueued */
    ITD_ERR_NONFINITE = 6      /* the input signal contains a NaN and the operator / the engine's mode rejects such input */
} itd_status;

/* stop reasons of the level loop */
#define ITD_STOP_NATURAL 0 /* "No more decompositions possible", ITD.py:404-416 */
#define ITD_STOP_TIMEOUT 1 /* "Out of time!",                    ITD.py:418-426 */

/* knot-detection modes for itd_detect_* */
#define ITD_DETECT_KNOTS 0  /* union used by the extraction, ITD.py:87-98 */
#define ITD_DETECT_VALLEYS 1 /* detect_peaks(x), ITD.py:33-76 (dx[i]>0 & dx[i-1]<=0) */
#define ITD_DETECT_PEAKS 2   /* detect_peaks(-x) = matlab_detect_peaks(x), numba_accelerated_itd.py:17-59 */
#define ITD_DETECT_CPP 3     /* itd.cpp:161-168: (x[i-1] < x[i] && x[i] >= x[i+1]) || (x[i-1] > x[i] && x[i] <= x[i+1]) */
#define ITD_DETECT_ZERO_CROSS 4 /* find_extrema's test, itd_fourier_decomposition.py:23-27: sign change x[i] -> x[i+1] */

typedef struct itd_engine itd_engine; /* opaque; not thread-safe: one engine per host thread/stream */
/* (Debugging: with PYITD_POISON=1 in the environment every workspace the library allocates is filled with 0xFF bytes before its
 *  first use — reading memory nobody wrote gives NaNs / -1 on every run instead of whatever the allocation happened to hold.) */

int itd_abi_version(void);
const char *itd_status_string(int status);
/* detail of the last ITD_ERR_HIP on this engine (static storage inside the engine) */
const char *itd_last_error(const itd_engine *e);

/* Create an engine on HIP device `device_id` able to decompose up to `max_batch` signals of up to
 * `max_n` samples per call.  Allocates the device workspace once (three rotating float64 baseline slots per signal =
 * 24 B per sample, 0.3 B per sample of per-tile records and counts, and one signal's worth of knot lists for the
 * single-level helpers).  No allocation happens
 * in the decompose calls (they are graph-capturable).  `max_batch` is bounded by memory only: a batch runs in chunks of
 * at most 65535 signals (itd_set_batch_chunk); the batched FITPACK-flavour call takes at most 65535 signals per call. */
int itd_engine_create(itd_engine **out, int device_id, int64_t max_n, int32_t max_batch);
void itd_engine_destroy(itd_engine *e);
int64_t itd_engine_workspace_bytes(const itd_engine *e);
int itd_engine_device(const itd_engine *e);

/* Plain device-memory helpers for bindings that own no GPU allocator (the numpy path of pyitd_amd.itd_batch, the C
 * client of tests/c_client): hipMalloc / hipFree / synchronous hipMemcpy on `device_id`, ordered against all work on the
 * device (itd_dev_copy synchronises the device first: the engines' own streams are non-blocking).  to_device: 1 = host -> device,
 * 0 = device -> host.  Callers that already hold device buffers (torch tensors, their own hipMalloc) never need these. */
int itd_dev_alloc(int device_id, int64_t bytes, void **out);
int itd_dev_free(int device_id, void *p);
int itd_dev_copy(int device_id, void *dst, const void *src, int64_t bytes, int32_t to_device);

/* ---- sharding a batch over the GPUs of a node (one process and one engine per GPU) ----
 * Signals are independent: rank r of `world` owns the contiguous range [lo, hi) of the batch (the first batch % world ranks get one
 * signal more) and decomposes it with its own engine; there is NO data-path collective.  itd_shard_range is that arithmetic
 * (pyitd_amd.distributed.shard_range).  itd_shard_scatter hands every rank its range when the batch lives on ONE rank — the
 * "trivial batch scatter" of the north star — as one group of ncclSend / ncclRecv over xGMI: `nccl_comm` is the caller's ncclComm_t
 * (as void *; RCCL is resolved with dlopen at the first call, the library does not link it), x_root_dev the root's [batch][n]
 * array (ignored elsewhere), x_local_dev this rank's [hi - lo][n] array, elem_bytes 4 or 8; enqueued on `stream`, no host
 * synchronisation.  world = 1 is a local copy and needs no communicator.  ITD_ERR_NO_DEVICE: no RCCL library on this host. */
int itd_shard_range(int64_t batch, int32_t world, int32_t rank, int64_t *lo, int64_t *hi);
int itd_shard_scatter(const void *x_root_dev, void *x_local_dev, int64_t n, int64_t batch, int32_t elem_bytes, int32_t world,
                      int32_t rank, int32_t root, void *nccl_comm, void *stream);

/* ---- full decomposition, device resident: replaces ITD.itd (ITD.py:351-432) -------------------
 * x_dev          [batch] signals, signal b starts at x_dev + b*x_stride (elements), n samples each
 * rows_dev       [batch][max_iteration+2][n] float64.  On return (after the stream has run) rows
 *                0 .. n_rows-1 of each signal are the reference's returned array: proper rotations, then
 *                the residual (natural stop: the previous baseline; timeout: rotation + baseline).
 * baselines_dev  optional [batch][max_iteration+2][n] float64: row j = baseline after extraction j+1
 *                (the reference's `baselines` buffer, ITD.py:385,429); NULL = keep only a ping-pong pair
 *                inside the engine (saves 8 B/sample/level of HBM capacity, same traffic).
 * Everything is enqueued on `stream` with no host synchronisation; call itd_get_summary afterwards — the rows are final once it
 * has returned (or, for stream-ordered consumers, under itd_set_valid_flags / itd_set_device_repair below).
 * Calls of one engine must be ordered with respect to each other (the same stream, or synchronised by the caller): they share
 * the engine's workspace, and a call's last launch leaves part of it initialised for the next call.  A call that is captured
 * into a graph initialises what it needs itself, so the graph can be replayed any number of times. */
int itd_decompose_f32(itd_engine *e, const float *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                      int32_t max_iteration, double *rows_dev, double *baselines_dev, void *stream);
int itd_decompose_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                      int32_t max_iteration, double *rows_dev, double *baselines_dev, void *stream);

/* Stream-ordered consumers (a kernel of the caller's enqueued behind the decomposition, a replayed hipGraph).
 * The engine runs optimistic forms first — the fused sparse levels, the fused level 0, the resident form of short signals — each
 * of which either delivers the reference's result or REPORTS that it cannot (tied / quantised / very smooth input, non-finite
 * values); by default itd_get_summary then repeats the call level by level, so rows_dev is final only once the summary has been
 * read.  Two settings make the result usable without that host round trip:
 *   itd_set_valid_flags(e, valid_dev)   valid_dev[batch] int32, device memory owned by the caller (NULL = off).  Behind the last
 *       launch of every later decomposition the engine writes valid_dev[b] = 1 if signal b's rows (and baselines) are final,
 *       0 if they are not (they will be once itd_get_summary has run).  One short launch.
 *   itd_set_device_repair(e, 1)          the engine also enqueues the level-by-level repeat itself, guarded on the device: its
 *       launches return at once for every signal whose first result stands and re-run the others, so that rows_dev is final —
 *       and valid_dev[b] = 1 — when the stream has drained, with no host synchronisation (graph-capturable).  A call whose
 *       optimistic forms all delivered pays the guarded launches' boundaries (a dozen empty launches, ~25 us).  The one case left
 *       to the host is a NaN in the caller's signal under ITD_NAN_INPUT_FOLLOW (valid_dev[b] = 0; itd_get_summary repeats such a
 *       call the way the reference runs it).  itd_get_summary stays valid after either setting and repeats nothing twice.
 * Reference behaviour: ITD.itd returns final arrays, ITD.py:404-432. */
int itd_set_valid_flags(itd_engine *e, int32_t *valid_dev);
int itd_set_device_repair(itd_engine *e, int32_t on);
int64_t itd_get_device_repairs(const itd_engine *e);   /* signals the device-side repair has re-run (counted when a summary is read) */

/* Synchronise with the last decomposition and fetch its per-signal summary (any pointer may be NULL):
 *   n_rows       [batch]      rows valid in rows_dev (= shape[0] of the reference's result)
 *   n_baselines  [batch]      rows valid in get_baselines() (ITD.py:414 / :424)
 *   stop_reason  [batch]      ITD_STOP_NATURAL / ITD_STOP_TIMEOUT
 *   knot_counts  [batch][ITD_MAX_ROWS+1]  knot_counts[j] = interior knots of the input of extraction j+1
 *                (j = 0: the signal itself; j >= 1: the number the reference prints at ITD.py:403);
 *                entries past the last evaluated level are -1
 *   nan_levels   [batch]      -1 = results follow the reference; -2 = the input signal itself contains a NaN and the engine
 *                was told to reject such input (itd_set_nan_input_mode: rows undefined)
 * NaNs that arise inside a decomposition are followed exactly: a baseline acquires NaNs when the signal starts with a
 * plateau (ITD.py:115-116 divides by x[e_1]-x[0] = 0); the reference's stop test then counts knots under detect_peaks'
 * NaN rules (ITD.py:46-51,64-68) and overwrites NaN with +inf in place (ITD.py:50).  Since ABI revision 2 the extraction
 * kernel applies these rules itself, tile by tile, in the same launch (no re-run, no extra pass over the signal). */
int itd_get_summary(itd_engine *e, int32_t *n_rows, int32_t *n_baselines, int32_t *stop_reason,
                    int64_t *knot_counts, int32_t *nan_levels);
/* NaN in the INPUT.  The reference runs it through detect_peaks' NaN branch at level 0 — NaN differences count as +inf, NaN
 * samples and their neighbours cannot be peaks — and overwrites the NaNs of the caller's array with +inf (ITD.py:46-51, 64-68;
 * there is no copy at ITD.py:41, 82, 389); everything after that sees the mutated array.
 *   ITD_NAN_INPUT_FOLLOW (default)  the decomposition found a NaN in a signal: itd_get_summary repeats the call with a level 0
 *                                   that follows those rules on a mutated float64 copy of the signal (inside the engine: x_dev is
 *                                   never written); rows / baselines / knot counts are the reference's, bit for bit.  As with the
 *                                   other repeats x_dev / rows_dev / baselines_dev must stay valid until itd_get_summary.
 *   ITD_NAN_INPUT_REJECT            ABI revision 2's behaviour: nan_levels = -2, host forms return ITD_ERR_NONFINITE.
 * The reference's own single-level functions follow the same mode: itd_detect_* in the modes KNOTS / VALLEYS (detect_peaks) /
 * PEAKS (matlab_detect_peaks: the NaN branch on the negated differences, numba_accelerated_itd.py:28-49) and
 * itd_baseline_extract_* repeat their scan under the NaN rules when they find a NaN (the device forms only when they
 * synchronise anyway, i.e. when m_host / count_host is given).  The cubic / spline / instantaneous operators, whose reference
 * code has no NaN branch, reject NaN input (ITD_ERR_NONFINITE). */
#define ITD_NAN_INPUT_FOLLOW 0
#define ITD_NAN_INPUT_REJECT 1
int itd_set_nan_input_mode(itd_engine *e, int32_t mode);
/* Level 0 of a decomposition (the caller's signal) finds its knots in one of two ways:
 *   fused    the level-0 extraction launch evaluates the knot predicate itself and takes each tile's neighbouring knots from
 *            the 128 samples either side of the tile (walking on through up to ~4000 samples where those hold too few): the
 *            signal is read once, there is no separate scan pass;
 *   records  a scan pass (k_scan0) leaves per-tile knot records, the extraction launch reads them (any knot spacing).
 * ITD_LEVEL0_AUTO (default): fused; if a tile's knots lie beyond the fused launch's reach (input smoother than ~4000
 * samples between extrema), itd_get_summary repeats the call record-driven before it returns — so, as before, x_dev /
 * rows_dev / baselines_dev must stay valid until itd_get_summary — and the engine's next 16 decompositions start
 * record-driven.  Results are identical in every mode. */
#define ITD_LEVEL0_AUTO 0
#define ITD_LEVEL0_RECORDS 1
#define ITD_LEVEL0_FUSED 2   /* never repeat: itd_get_summary fails with ITD_ERR_HIP if the reach was exceeded (benchmarks) */
int itd_set_level0_mode(itd_engine *e, int32_t mode);
/* The sparse levels fused ("knot first", pyitd_amd/csrc/itd_knotfirst.hpp; ABI revision 6, one knot-side launch since revision 7).
 * An extraction maps every sample through an affine function of itself inside its segment (ITD.py:114-117), so the next level's
 * knots sit at this level's knots (apart from places where near ties let rounding make or break a plateau: both samples of every
 * near tie of the first fused level's input stay candidates): from level `first_fused_level` on the level recursion runs on the
 * knot list alone — ONE launch, the lists resident in LDS — and the samples take ONE pass for all remaining levels: 8 B read + 8 B
 * per row written per sample instead of 24 B per sample and level.  The sample pass re-derives every level's knots from the values
 * it computes; where they differ from the knot side's (or a list outgrows its workgroup, or knot data go non-finite: coarsely
 * quantised, plateau-ridden or very smooth signals) itd_get_summary repeats the call level by level before it returns (x_dev /
 * rows_dev / baselines_dev must stay valid until then, as before; or on the device: itd_set_device_repair) and the engine's next 16
 * decompositions start level by level (32, 64, ... 1024 when the fused attempt that follows such a pause refuses again: a workload
 * the fused form cannot deliver pays one wasted attempt in ever more calls; a delivered call starts over at 16) — after a list
 * outgrew its workgroup they stay fused, handing over a level later or with half the tiles per workgroup (itd_set_fuse_range); when
 * only a few signals of a batch are concerned (at most one in eight) just those are run again, each on its own, and the engine stays
 * in the fused form.
 * Results are bit-identical in every mode: what the fused form cannot deliver it reports.  The fused levels' workspace is allocated
 * by the first call that takes this path; a call being captured into a graph cannot allocate: on an engine that has not fused yet it is
 * captured level by level (run one decomposition before the capture to get the fused form into the graph).  A graph that holds a fused
 * call stays replayable whatever the engine does afterwards: when later calls need a larger workspace (smaller ranges after a capacity
 * refusal, itd_set_fuse_range, itd_set_fuse_level) the one the graph refers to is kept until itd_engine_destroy, never freed.
 * ITD_FUSE_AUTO (default): calls whose launch sequences cover at least itd_set_fuse_min_samples samples (signals per chunk x n; default
 * 2 * 2^20: one signal of 2^20 / 2^21 / 2^22 / 2^23 samples takes 89 / 115 / 183 / 305 us level by level and 90 / 104 / 149 / 227 us
 * fused), signals of >= 65536 samples; ITD_FUSE_OFF: never; ITD_FUSE_ONLY: always, never repeat (itd_get_summary fails with
 * ITD_ERR_HIP instead: tests, benchmarks).  itd_set_fuse_level: the first fused level, 2 .. max_iteration, or 0 (default) = automatic:
 * level 2 where a launch sequence covers at least 2^22 samples (levels 0 and 1 as one launch each; 377 against 404 us at 2^24), else
 * level 3; a level-2 candidate list that outgrows its workgroup moves the engine's later calls to level 3 before the ranges are halved. */
#define ITD_FUSE_AUTO 0
#define ITD_FUSE_OFF 1
#define ITD_FUSE_ONLY 2
int itd_set_fuse_mode(itd_engine *e, int32_t mode);
int itd_set_fuse_level(itd_engine *e, int32_t first_fused_level);
/* Tiles (of 512 samples) a knot-side workgroup of the fused levels owns: 64, 32 or 16; 0 (default) = automatic: 64, halved for the
 * calls after one in which a workgroup's candidate list (1720 entries at the first fused level, 1024 from the next on: knots and
 * near-tie samples) outgrew its LDS — dense knots at the first fused level (white noise from level 2, alternating data).  Smaller
 * ranges hold denser lists and cost more workgroups. */
int itd_set_fuse_range(itd_engine *e, int32_t tiles);
int itd_set_fuse_min_samples(itd_engine *e, int64_t samples);
/* ABI revision 11: capped fused levels.  A workload whose fused form is refused at the same level every time — exactly periodic input
 * whose baseline collapses to a handful of knots at some level (BASELINE configs[4]'s substitute clip: 1049 knots, then 3, at level 8):
 * every sample a near tie there — keeps the fused form for the levels in front of that one: levels first_fused .. cap - 1 run fused
 * (the sample pass also stores the baseline behind level cap - 1), levels cap .. max_iteration + 1 one launch each from a scan of that
 * baseline.  first_level_not_fused = 0 (default): automatic — a whole-call refusal records the lowest level at which anything failed
 * (verification, non-finite knot data) and the engine's next calls are capped there instead of running level by level (after 16
 * delivered capped calls one call tries all levels again; a probe refused at the learned level doubles that span, up to 1024; a
 * delivered probe drops the cap); -1: never; 4 .. max_iteration + 1: always this cap (tests).  A cap leaves at least two fused levels or
 * is ignored.  Results are bit-identical either way. */
int itd_set_fuse_cap(itd_engine *e, int32_t first_level_not_fused);
/* the cap of the last decomposition as it was enqueued (0: none — every level from the first fused one on ran fused, or no fused levels) */
int itd_get_last_fuse_cap(const itd_engine *e);
/* batches: how many consecutive chunks (itd_set_batch_chunk) share ONE knot-side launch of the fused levels (default 1; sharing it
 * paid while the knot side was a dozen launches) */
int itd_set_fuse_group(itd_engine *e, int32_t chunks);
/* Tests only (ABI revision 9): arm ONE fault in what the fused levels' knot side hands to their sample pass, applied to signal 0 of
 * every following fused call of this engine until disarmed (kind < 0).  The sample pass verifies everything it takes from the knot
 * side (itd_knotfirst.hpp: V0 .. V3); a fault in a field it uses must end in a refusal (ITD_FUSE_ONLY: itd_get_summary fails;
 * otherwise the call is repeated level by level, itd_get_fuse_repeats counts it) — tests/test_gpu_fused.py holds that for thousands.
 *   kind 0 / 1 / 2  delta ulps added to X / B / S of table entry `slot` (mod the run's length) of tile `where`'s run at `level`;
 *   kind 3          delta added to that entry's knot position;     kind 4   delta added to the run's start index first[level][where];
 *   kind 5          bit (delta & 63) of flag word (slot & 7) of tile `where` at `level` flipped;
 *   kind 6 / 7      the value (delta ulps) / position (delta) of halo knot `slot` (0, 1: the two in front, 2 .. 4: the three behind) as
 *                   knot-side workgroup `where` receives it from its neighbours at `level`;
 *   kind 8          delta added to the knot side's count of `level`'s knots (what its stop rules read): the sample pass's check
 *                   wavefronts count the verified flag words themselves, the verdict compares.
 * `level` is the absolute level (first fused level .. max_iteration + 1). */
int itd_debug_kf_fault(itd_engine *e, int32_t kind, int32_t level, int32_t where, int32_t slot, int32_t delta);
/* Tests only (ABI revision 11): the armed fault lands in signal `signal` of the batch instead of signal 0 (kinds 6 / 7: in that signal's
 * knot-side workgroup `where`) — the fused levels of a batch run many signals per launch, and what is verified for signal 0 has to hold
 * for every other one. */
int itd_debug_kf_fault_signal(itd_engine *e, int32_t signal);
/* Tests only (ABI revision 9): the kernels form the knot spacings' ratio (k1 - k0) / (k2 - k0) of ITD.py:107 with the division's own
 * instruction sequence minus its range scaling and special-case fix-up, which do nothing for exact small integers (itd_kernels.hpp:
 * int_ratio).  This runs that sequence against the compiler's full float64 division on the device for EVERY pair 0 <= a <= b <= max_den
 * and for 2^30 pseudo-random pairs below 2^31: *mismatches = pairs whose results differ in any bit (must be 0). */
int itd_debug_int_ratio_check(int device, int32_t max_den, int64_t *mismatches);
/* how many whole calls of this engine itd_get_summary has had to repeat level by level because the fused levels reported a failure */
int itd_get_fuse_repeats(const itd_engine *e);
/* the first fused level of the last decomposition as it was enqueued (2, 3, ...), 0 if it ran one launch per level throughout
 * (too short, ITD_FUSE_OFF, a back-off) or in the resident form: what a benchmark's byte model needs (ABI revision 9) */
int itd_get_last_fuse_level(const itd_engine *e);
/* ... and how many single signals of batches it has re-run on their own (the rest of their batch kept the fused result) */
int64_t itd_get_fuse_signal_repairs(const itd_engine *e);
/* Short signals (n <= 8192 samples): the resident form — ONE launch, one workgroup per signal, the signal and its knot arrays
 * in LDS through all levels of the driver loop (ITD.py:384-432): the signal is read once and every result row written once
 * (4 + 8 rows bytes per sample; the level-by-level form is launch bound there: 10 dependent launches).  Baselines that go NaN
 * (a leading or trailing plateau, ITD.py:115-116) and NaNs in the caller's signal follow the reference's NaN rules inside the kernel;
 * only under ITD_NAN_INPUT_REJECT does a NaN input make itd_get_summary repeat the call level by level before it returns (x_dev /
 * rows_dev / baselines_dev must stay valid until then, as before; the engine's next 16 decompositions then start level by level).
 * Rows past n_rows are not written in this form.
 * ITD_RESIDENT_AUTO (default): resident for n <= 8192 unless the engine was given a level-0 mode or kernel timing;
 * ITD_RESIDENT_OFF: never; ITD_RESIDENT_ONLY: always for n <= 8192, never repeat (itd_get_summary fails with ITD_ERR_HIP
 * instead; tests, benchmarks).  Results are identical in every mode. */
#define ITD_RESIDENT_AUTO 0
#define ITD_RESIDENT_OFF 1
#define ITD_RESIDENT_ONLY 2
int itd_set_resident_mode(itd_engine *e, int32_t mode);
/* how many resident calls of this engine itd_get_summary has had to repeat level by level so far */
int itd_get_resident_repeats(const itd_engine *e);
/* The resident form holds a window of `segments` consecutive knot-to-knot segments of a level in LDS at a time; a level with more
 * knots takes several passes.  0 = automatic (0.4 knots per sample: one pass for ordinary signals), otherwise >= 8 (cut to what
 * fits the LDS).  Results do not depend on it (tests run tiny windows to exercise the passes). */
int itd_set_resident_window(itd_engine *e, int32_t segments);
/* Batched decompositions run as launch sequences over chunks of `signals_per_chunk` signals, all levels of a chunk before
 * the next chunk (0 = automatic: about 2^24 samples in flight — per chunk over one stream, 3/4 of that per chunk over two —, so a
 * level's baseline is still in the 256 MiB Infinity Cache when the next level reads it).  Results do not depend on the chunk size. */
int itd_set_batch_chunk(itd_engine *e, int32_t signals_per_chunk);
/* The chunks of a batch are independent: they rotate over `streams` streams (the caller's and streams - 1 of the engine's, forked
 * from / joined to the caller's stream by events), so that one chunk's launch boundaries and tails overlap another's work.
 * 1 .. 4; default 2 (with automatic chunks of about 1.2e7 samples: 32.5 ms against 35.9 ms over one stream for 1024 x 2^20). */
int itd_set_batch_streams(itd_engine *e, int32_t streams);
/* ABI revision 11.  on = 1: a batch whose chunks run the fused sparse levels (itd_set_fuse_mode) is PIPELINED (needs
 * itd_set_batch_streams >= 2): the caller's stream runs every chunk's level launches and its knot side in order, ONE stream of the
 * engine the chunks' passes over the samples, chunk k's beside chunk k + 1's knot side (latency-bound: ~50 us per chunk with HBM idle).
 * A 65 KB workgroup finds no room on a device that one-wavefront workgroups have filled, so each sample pass waits behind a gate (one
 * wavefront that returns when every workgroup of the next knot side has started) instead of behind an event.  on = 0 (default): the
 * chunks rotate over the streams, each chunk's launches in order on its stream.  Measured on 512 x 2^20 samples, 8 levels: 12.8-13.0 ms
 * pipelined against 11.9-12.3 ms rotating — a resident knot side costs the memory-bound launches a quarter of the wave slots, more than
 * hiding it returns (profiles/r06/experiments/README.md); kept for devices / shapes where that balance differs.  Results do not depend
 * on it; a call that is being captured into a hipGraph always takes the rotating form.  The gate also orders a sample pass behind its OWN
 * knot side; one that gives up (50 ms: the device busy with somebody else's work) records it, the call is refused as a whole and repeated
 * level by level — by itd_get_summary, or by the device-side repair (itd_set_device_repair) — and the engine's later batches rotate. */
int itd_set_batch_pipeline(itd_engine *e, int32_t on);

/* Per-level knot lists are not retained by a decomposition (each level's list is consumed by the next
 * launch); to inspect them run itd_detect_* on the input or on a stored baseline row.  The single-level operators
 * below work in a workspace of their own: calling them between itd_decompose_* and itd_get_summary (on any stream)
 * does not disturb the decomposition.  A signal that contains a NaN: see itd_set_nan_input_mode. */

/* ---- one-call host convenience (numpy in -> numpy out, what the reference's callers see) -------
 * Copies x to the GPU, decomposes, copies rows (and baselines if non-NULL) back.
 * rows_host / baselines_host: [max_iteration+2][n] float64, caller allocated. */
int itd_decompose_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t max_iteration,
                           double *rows_host, double *baselines_host, int32_t *n_rows, int32_t *n_baselines,
                           int32_t *stop_reason, int64_t *knot_counts /* [ITD_MAX_ROWS+1] */);
int itd_decompose_host_f32(itd_engine *e, const float *x_host, int64_t n, int32_t max_iteration,
                           double *rows_host, double *baselines_host, int32_t *n_rows, int32_t *n_baselines,
                           int32_t *stop_reason, int64_t *knot_counts);

/* The reference keeps the baselines of the last run on the instance (ITD.py:413-414,423-424) but most callers only look at
 * the returned rows: with itd_set_host_keep_baselines(e, 1) a host-form call with baselines_host == NULL still computes the
 * baselines buffer and leaves it on the device (half the PCIe traffic of the call); itd_get_last_baselines_host copies its
 * first n_baselines rows ([n_baselines][n] float64; n and n_baselines as the call reported them) out later — until the next
 * host-form call on this engine, after which it returns ITD_ERR_NOT_RUN. */
int itd_set_host_keep_baselines(itd_engine *e, int32_t enable);
int itd_get_last_baselines_host(itd_engine *e, double *baselines_host, int64_t n, int32_t n_baselines);

/* ---- single-level operators ---------------------------------------------------------------------
 * itd_baseline_extract (ITD.py:79-121): rotation/baseline of ONE extraction of a device signal.
 * knots_dev (optional, capacity n int32) receives the interior knot indices, *m_host their count. */
int itd_baseline_extract_f64(itd_engine *e, const double *x_dev, int64_t n, double *rot_dev, double *base_dev,
                             int32_t *knots_dev, int64_t *m_host, void *stream);
int itd_baseline_extract_f32(itd_engine *e, const float *x_dev, int64_t n, double *rot_dev, double *base_dev,
                             int32_t *knots_dev, int64_t *m_host, void *stream);
/* host convenience: rot/base [n] float64, knots_host (optional) [n] int64, bk_host (optional) [n+2] knot values */
int itd_baseline_extract_host_f64(itd_engine *e, const double *x_host, int64_t n, double *rot_host,
                                  double *base_host, int64_t *knots_host, int64_t *m_host, double *bk_host);

/* detect_peaks / matlab_detect_peaks / knot union (ITD.py:33-76, :87-98): ordered indices of a device
 * signal.  idx_dev capacity n int32; the count is returned through *count_host (synchronises). */
int itd_detect_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t mode, int32_t *idx_dev,
                   int64_t *count_host, void *stream);
int itd_detect_f32(itd_engine *e, const float *x_dev, int64_t n, int32_t mode, int32_t *idx_dev,
                   int64_t *count_host, void *stream);
int itd_detect_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t mode, int64_t *idx_host,
                        int64_t *count_host);

/* baseline_knot_estimation (numba_accelerated_itd.py:167-178 = ITD.py:106-110): fills bk[1..m] for the
 * caller's extended knot list e[0..m+1] (bk[0], bk[m+1] are left as given, like the reference).
 * All host pointers; x [n], extrema [m+2] int64, bk [m+2]. */
int itd_knot_values_host_f64(itd_engine *e, const double *x_host, int64_t n, const int64_t *extrema_host,
                             int64_t m, double *bk_host);
/* The same on device buffers, enqueued on `stream` (no synchronisation, no validation: the caller guarantees
 * 0 <= extrema_dev[k] < n): x_dev [n] float64, extrema_dev [m+2] int32 (what itd_detect_* delivers, with the end knots
 * 0 and n-1 around it), bk_dev [m+2] float64 of which bk_dev[1..m] are written. */
int itd_knot_values_f64(itd_engine *e, const double *x_dev, int64_t n, const int32_t *extrema_dev, int64_t m,
                        double *bk_dev, void *stream);

/* ---- cubic-spline baseline variant with externally supplied knots ---------------------------
 * itd_baseline_extract_fast(I, extrema_input, idx), itd_fourier_decomposition.py:49-122 — the Python twin of
 * itd_baseline_extract(data, baseline, length, &idx, compute_extrema), itd.cpp:156-239.  Restated literally in float64,
 * quirks included (knot values for k = 1..idx-2 only, K[idx-1] = 0, K[idx] = I[e[idx]], the sweep as written, natural ends,
 * the segment idx-2 evaluated linearly).  Float results agree with the reference to rounding (the reference's numpy form
 * uses libm pow for t**3, numba multiplies, the knot recurrences run here as scans): tests hold them to 1e-9 of the
 * signal's scale; knot indices are exact.
 *   extrema_dev   idx+1 knots (int32, device): e[0..idx-1] strictly increasing sample indices, e[idx] any sample index
 *                 (find_extrema leaves 0 there); 2 <= idx <= n-1.  Invalid lists return ITD_ERR_INVALID_ARG.
 *                 NULL = compute_extrema (itd.cpp:159-169): the knots are the samples with
 *                 (x[i-1] < x[i] && x[i] >= x[i+1]) || (x[i-1] > x[i] && x[i] <= x[i+1]), e[idx] = 0 (the file's static
 *                 array at first call); *idx_host receives their count; fewer than 2 leave baseline_dev untouched (:170).
 *   the float32 form widens the signal first: itd.cpp's all-float32 arithmetic has no compilable reference to pin. */
int itd_baseline_extract_cubic_f64(itd_engine *e, const double *x_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                   double *baseline_dev, int64_t *idx_host, void *stream);
int itd_baseline_extract_cubic_f32(itd_engine *e, const float *x_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                   double *baseline_dev, int64_t *idx_host, void *stream);
/* host form: extrema_host int64 [idx+1] or NULL (detect); extrema_out_host (optional, capacity n) receives detected knots */
int itd_baseline_extract_cubic_host_f64(itd_engine *e, const double *x_host, int64_t n, const int64_t *extrema_host,
                                        int64_t idx, double *baseline_host, int64_t *idx_out, int64_t *extrema_out_host);
/* ABI revision 11: the common-baseline form of the same operator on COMPLEX (I/Q) data, itd_baseline_extract_iq, itd.cpp:58-154 (a
 * non-compilable float32 fragment like the 1-D form below it: restated in float64 the way its Python twin restates the 1-D form; no
 * upstream test or Python twin pins the I/Q form: "recipe unpinned").  iq: n complex samples, interleaved (re, im), 16-byte aligned.
 *   knots      the samples at which BOTH components have an extremum under the file's 3-point predicate (itd.cpp:74-80), or the
 *              caller's list (extrema_dev / extrema_host with idx as in itd_baseline_extract_cubic_f64: "retain the extrema and reuse
 *              them ... along multiple channels", itd.cpp:40-44)
 *   baseline   ONE real baseline [n]: the natural-cubic operator above on the components' mean (I + Q) / 2 (itd.cpp:96-108)
 * Fewer than 2 knots leave the baseline buffer untouched (itd.cpp:85-87); *idx_host / *idx_out = the knot count.  Synchronous (one host
 * synchronisation for the knot count when the knots are detected, one at the end for the operator's validity). */
int itd_baseline_extract_iq_f64(itd_engine *e, const double *iq_dev, int64_t n, const int32_t *extrema_dev, int64_t idx,
                                double *baseline_dev, int64_t *idx_host, void *stream);
int itd_baseline_extract_iq_host_f64(itd_engine *e, const double *iq_host, int64_t n, const int64_t *extrema_host, int64_t idx,
                                     double *baseline_host, int64_t *idx_out, int64_t *extrema_out_host);
/* find_extrema(signal), itd_fourier_decomposition.py:17-31: [0, the sign changes s[i] -> s[i+1], one extrapolated index],
 * zero padded to n entries like the reference's numpy.zeros array; *idx_out = the reference's returned idx. */
int itd_find_extrema_host_f64(itd_engine *e, const double *s_host, int64_t n, int64_t *extrema_host, int64_t *idx_out);

/* ---- the FITPACK flavour of the baseline and its 2-D consumers -----------------------------
 * itd_baseline_extract_modified(x), numba_accelerated_itd.py:182-211 (= itd_baseline_extract, siftED2D.ipynb cell 1; MEITD.py:
 * 303-338 is the same operator without the early-out): knots = the tier-1 set, baseline knot values with odd-reflected ends
 * (:196-206), the INTERPOLATING cubic B-spline through them — custom_splrep -> scipy.interpolate.splrep(x, y, k=3), whose
 * default without weights is s = 0; SciPy's FITPACK curfit, restated in pyitd_amd/csrc/itd_fitpack.hpp and held bit for bit to
 * scipy's splrep (tests/test_fitpack_host.py) — evaluated at every sample like numba_splev (:89-164, its equi_spaced interval
 * formula included).  Batched: signal b at x_dev + b*x_stride, n samples each; one GPU thread per signal computes the
 * coefficients (the sweep over the knots is serial), one thread per sample evaluates.
 *   min_extrema   signals with fewer knots are returned unchanged (baseline = x): 10 for numba_accelerated_itd.py:188-190 and
 *                 siftED2D, 0 for MEITD's form; fewer than 2 knots are always returned unchanged (splrep needs m > k)
 *   rot_dev       optional: x - baseline (MEITD.py:335)
 *   knots_host    optional [batch]: the knot count of every signal
 * Returns ITD_ERR_NONFINITE if a signal holds a NaN.  Synchronous. */
int itd_baseline_extract_spline_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                                    int32_t min_extrema, double *baseline_dev, int64_t baseline_stride, double *rot_dev,
                                    int64_t rot_stride, int32_t *knots_host, void *stream);
int itd_baseline_extract_spline_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t min_extrema,
                                         double *baseline_host, double *rot_host, int32_t *knots_host);
/* How the spline's coefficients are obtained (ABI revision 6):
 *   ITD_SPLINE_SERIAL    FITPACK's own row-by-row sweep, one GPU thread per signal, bit-level against scipy's splrep: right for
 *                        thousands of short rows (the image sweeps always use it)
 *   ITD_SPLINE_PARALLEL  the same interpolating not-a-knot spline from its second derivatives, parallel in the knots
 *                        (pyitd_amd/csrc/itd_nak.hpp): equal to FITPACK's result to rounding (~1e-14 of the signal's scale; the
 *                        north star allows 1e-6), not bit for bit; right for ONE long signal (MEITD.py:344-549 calls the operator on
 *                        single 1-D signals in a loop)
 *   ITD_SPLINE_AUTO      (default) parallel for calls of fewer than 256 signals of at least 1024 samples, serial otherwise */
#define ITD_SPLINE_AUTO 0
#define ITD_SPLINE_SERIAL 1
#define ITD_SPLINE_PARALLEL 2
int itd_set_spline_solver(itd_engine *e, int32_t solver);
/* the host form with one more optional output (ABI revision 6): baseline_knots_host [batch] = the knot count (ITD_DETECT_KNOTS) of
 * every PRODUCED baseline — MEITD's loops ask for it right after an extraction (MEITD.py:362-363, :497-505); it is counted on the
 * device from the result that is already there (no second upload, no index list) */
int itd_baseline_extract_spline_host2_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t min_extrema,
                                          double *baseline_host, double *rot_host, int32_t *knots_host, int32_t *baseline_knots_host);
/* Knot counts of `batch` contiguous host signals under predicate `mode` (ITD_DETECT_*) without building or copying any index
 * list — `matlab_detect_peaks(x).size + matlab_detect_peaks(-x).size` (MEITD.py:350, :376, :409) is mode ITD_DETECT_KNOTS.
 * Plain rules; returns ITD_ERR_NONFINITE (counts filled in) if a signal holds a NaN.  batch <= 65535. */
int itd_count_knots_host_f64(itd_engine *e, const double *x_host, int64_t n, int32_t batch, int32_t mode, int32_t *counts_host);
/* ---- MEITD's operators on device-resident signals (ABI revision 8) --------------------------------
 * MEITD.py:344-534 keeps one signal, its rotation and its baseline in a loop of extractions, knot counts and entropy tests: with
 * these entries the arrays stay on the device for the whole loop and only scalars come back (pyitd_amd/meitd.py).
 *
 * itd_count_knots_f64: itd_count_knots_host_f64 for signals that are on the device already (signal b at x_dev + b*x_stride).
 *
 * itd_wpe3_f64: weighted_permutation_entropy(x, order=3), MEITD.py:79-128, as far as it touches the samples — for each of the six
 * permutation patterns of the windows (x[i], x[i+1], x[i+2]), in numpy.unique's order of the reference's hash values
 * (5, 7, 11, 15, 19, 21 = argsort (2,1,0), (1,2,0), (2,0,1), (0,2,1), (1,0,2), (0,1,2)): bin_weights_host[6] the sum of the
 * windows' variances, bin_windows_host[6] the number of windows (a pattern without windows is absent from the reference's list).
 * The argsort is numpy's (insertion on three values, NaNs last, ties in index order), the variance numpy.var's expression; up to
 * 65536 windows the sums are taken one by one in index order like the reference's cumsum, longer signals in segments of 4096
 * windows added in order (deterministic, equal to rounding).  The entropy itself is six logarithms on the host:
 * p = w / sum(w); -sum(p log2 p) [/ log2(6)].  knots_host (optional): x's knot count (ITD_DETECT_KNOTS, plain rules) from the same
 * launch — a window's middle sample is a knot or not; MEITD.py:346-351, :373-378 ask for both; then ITD_ERR_NONFINITE (count filled
 * in) if x holds a NaN.
 *
 * itd_wpe_f64: the same pass for any `order` 2 .. 5 (MEITD.py:79 takes the order; MEITD itself only passes 3): sums_host and
 * windows_host have order^order entries, indexed by the reference's hash value sum(argsort[k] * order^k); the entropy is drawn from
 * the entries with windows, in ascending hash order (numpy.unique's).  Sums in index order per hash up to 65536 windows, in
 * segments of 16384 beyond.  A side path: its work grows with order^order x windows.
 *
 * itd_baseline_extract_spline2_f64: itd_baseline_extract_spline_f64 plus baseline_knots_host [batch], the knot count of every
 * PRODUCED baseline, one synchronisation for both (the device form of itd_baseline_extract_spline_host2_f64).
 *
 * itd_subtract_f64: out = a - b, elementwise (MEITD.py:453), enqueued.   itd_copy: a copy ordered on the engine's stream (or
 * `stream`): kind 0 device -> host, 1 host -> device, 2 device -> device, 3 zero fill (src ignored); wait != 0 returns when done. */
int itd_count_knots_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride, int32_t mode,
                        int32_t *counts_host, void *stream);
int itd_wpe3_f64(itd_engine *e, const double *x_dev, int64_t n, double *bin_weights_host, int64_t *bin_windows_host,
                 int32_t *knots_host, void *stream);
int itd_wpe_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t order, double *sums_host, int64_t *windows_host, void *stream);
int itd_baseline_extract_spline2_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                                     int32_t min_extrema, double *baseline_dev, int64_t baseline_stride, double *rot_dev,
                                     int64_t rot_stride, int32_t *knots_host, int32_t *baseline_knots_host, void *stream);
int itd_subtract_f64(itd_engine *e, const double *a_dev, const double *b_dev, double *out_dev, int64_t count, void *stream);
int itd_copy(itd_engine *e, void *dst, const void *src, int64_t bytes, int32_t kind, int32_t wait, void *stream);
/* MEITD's whole selection loop (MEITD.py:395-534 with retrieve_proper_rotation :344-368 and determine_if_first_is_proper_rotation
 * :371-392) on ONE device-resident signal of 3 .. 8192 samples as one launch of one workgroup (ABI revision 10): the extractions,
 * extrema counts and entropy tests above run behind one another on the device and the branch is taken there — no host round trip
 * per operator (the host-driven loop: ~107 of them for a 3000-sample signal).  Only where the engine's solver setting gives the
 * parallel-in-knots form for such a signal (ITD_SPLINE_PARALLEL, or ITD_SPLINE_AUTO and n >= 1024); otherwise ITD_ERR_INVALID_ARG.
 * rows_dev: (6 + 2 * 22) rows of n float64 in one allocation — rows 0..5 working rows (the signal in row 5 on entry), rows 6..27
 * the kept rotations taken off the signal ("high", MEITD.py:447), rows 28..49 those taken off its baselines ("low", :450).
 * wpemax: MEITD's WPEMAX.  result_host[24]: status, number of high rows, number of low rows, the working row that holds the
 * residual, entropy probes taken, extractions run, turns of the loop, 0, then sixteen diagnostic words — the launch's time in
 * units of 10 ns spent in the probes, the extractions, the extrema counts, the row copies / subtractions, the probes' ordered sums,
 * their entropies, two zeros, the extractions' phases (knots, knot values, rows, forward sweep, backward sweep, evaluation), two
 * zeros.  status 0: delivered; 1: the signal has fewer than four
 * extrema (MEITD.py:411-413 returns zeros and the data); 2 (a NaN in a row), 3 (an extraction met fewer than two knots: scipy
 * raises there), 4 (more than 1024 probes): NOT delivered — run the loop from the host, which reproduces the reference's behaviour
 * for those.  probe_log_host (log_cap entries of 88 bytes: double w[6]; int32 c[6]; int32 count; int32 0; double entropy): every
 * entropy probe's six weighted sums, window counts, extrema count and the normalised entropy the device drew from them with its
 * own log2 — the caller re-draws each with the reference's numpy expression and compares the threshold test
 * (wpe < WPEMAX and not wpe < 0.2, MEITD.py:364 / :387), the only use an entropy has: selections identical to the reference's by
 * construction, not by the two log2 agreeing.  Synchronous. */
int itd_meitd_small_f64(itd_engine *e, double *rows_dev, int64_t n, double wpemax, int32_t *result_host, void *probe_log_host,
                        int32_t log_cap, void *stream);
/* crossways_itd_baseline_extract(data), siftED2D.ipynb cell 1, for `planes` images of rows x cols float64 (the ensemble
 * members of retrieve_statistical_image_component go through in one call): the operator over every row, then over every
 * column of that; over every column, then every row of that; the mean of the two.  Transposes and the mean run on the GPU. */
int itd_crossways_f64(itd_engine *e, const double *img_dev, int32_t planes, int32_t rows, int32_t cols, int32_t min_extrema,
                      double *out_dev, void *stream);
int itd_crossways_host_f64(itd_engine *e, const double *img_host, int32_t planes, int32_t rows, int32_t cols, int32_t min_extrema,
                           double *out_host);

/* ---- instantaneous amplitude / phase / frequency of a proper rotation --------------------------
 * The time-frequency-energy step the reference describes (README.md:13-21, 41-55) but does not implement; definitions of the
 * paper it quotes (Frei & Osorio 2007, single-wave analysis): half waves between zero crossings, amplitude = the half wave's
 * largest |x|, phase by arcsin(x / amplitude) placed in the wave's quadrant, frequency = forward phase difference / 2 pi in
 * cycles per sample.  rot_dev: one row of the decomposition (or any oscillation about zero); any output may be NULL. */
int itd_instantaneous_f64(itd_engine *e, const double *rot_dev, int64_t n, double *amp_dev, double *phase_dev, double *freq_dev,
                          void *stream);
int itd_instantaneous_host_f64(itd_engine *e, const double *rot_host, int64_t n, double *amp_host, double *phase_host,
                               double *freq_host);

/* ---- batched single-level operators (ABI revision 6): asynchronous on `stream`, no host synchronisation, graph-capturable ---
 * The reference applies its single-level functions row by row (siftED2D.ipynb cell 1: itd_baseline_extract over the rows of an
 * image under numba.prange) and along channels (itd.cpp:40-44).  Signal b starts at x_dev + b * x_stride.  These calls work in
 * workspaces of their own (grown on demand: the first call of a size allocates — so capture a call only after one of its size has
 * run), apart from the decomposition's.
 *
 * itd_baseline_extract_batch_f64: itd_baseline_extract (ITD.py:79-121) of every signal: rot_dev / base_dev [batch] rows of n.
 *   info_dev (optional) [batch]: the signal's interior knot count; -1 - count if the signal holds a NaN — its rows then follow
 *   the plain rules, not detect_peaks' NaN branch (ITD.py:46-51): run that signal through itd_baseline_extract_* .
 * itd_detect_batch_f64: the knots of every signal by predicate `mode` (ITD_DETECT_*): idx_dev (optional) [batch] lists at
 *   idx_stride >= n - 2 (the knots alone, ascending); info_dev (optional) as above.  idx_dev == NULL: counts only (no list is
 *   built).  batch <= 65535.
 * itd_baseline_extract_cubic_batch_f64: itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122) of every signal.
 *   extrema_dev: idx + 1 knots as in itd_baseline_extract_cubic_f64 — ONE list for every signal (extrema_stride = 0: "retain the
 *   extrema and reuse them ... along multiple channels", itd.cpp:40-44) or one per signal (extrema_stride >= idx + 1);
 *   NULL = every signal's own knots by itd.cpp:159-169.  A signal with fewer than 2 knots or an invalid list keeps its
 *   baseline row untouched.  info_dev (optional) [batch]: the idx used; -1 = invalid list, -2 = NaN in the signal.  batch <= 65535. */
int itd_baseline_extract_batch_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride, double *rot_dev,
                                   int64_t rot_stride, double *base_dev, int64_t base_stride, int32_t *info_dev, void *stream);
int itd_detect_batch_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride, int32_t mode, int32_t *idx_dev,
                         int64_t idx_stride, int32_t *info_dev, void *stream);
int itd_baseline_extract_cubic_batch_f64(itd_engine *e, const double *x_dev, int64_t n, int32_t batch, int64_t x_stride,
                                         const int32_t *extrema_dev, int64_t extrema_stride, int64_t idx, double *baseline_dev,
                                         int64_t baseline_stride, int32_t *info_dev, void *stream);

/* ---- block-wise (streaming) operation for unbounded signals (ABI revision 6) ------------------------------
 * The recipe of the comment at itd.cpp:31-38 — "use a circular buffer with modulous tracking to rotate the samples / re-assess
 * extrema in the entire buffer every iteration / use from the last extrema in the first buffer to the first extrema in the last
 * buffer / set the first and last baseline knots manually to said values / update the j array / compute only the baseline[i]
 * array for the inner third of the buffer overall / rotate buffers, rinse and repeat" — and itd.cpp:40-44's retained extrema
 * along channels.  There is no reference code for it; oracle/stream_oracle.py states the recipe the tests hold this to.
 *
 * A stream keeps the last three blocks of `channels` channels in a device-resident ring.  A push stores one block of every
 * channel and, from the second push on, emits the baseline (and rotation) of the PREVIOUS block — latency one block; flush
 * emits the last block and empties the stream.  Window = the previous, the emitted and the next block (two blocks at either
 * end of the stream, one if the stream holds a single block).
 *   ITD_STREAM_CUBIC   itd_baseline_extract_fast (itd_fourier_decomposition.py:49-122) on the window's extrema (itd.cpp:161-168)
 *                      from `margin` extrema in front of the emitted block to margin + 2 behind it (the recipe's literal
 *                      choice is margin 1; the two extra keep the operator's uncomputed second-to-last knot value, K[idx-1]
 *                      = 0, outside the block); the first and last knot value are the data there (itd.cpp:35).  Fewer than 4
 *                      such extrema: the block is emitted unchanged (itd.cpp:170-172).  shared_knots: the extrema of channel
 *                      0's window serve every channel (itd.cpp:40-44).
 *   ITD_STREAM_LINEAR  itd_baseline_extract (ITD.py:79-121) on the window, every channel on its own knots.  The operator is
 *                      local (two knots either side of a sample), so wherever the blocks hold a few knots the emitted stream is
 *                      bit-identical to the whole-signal operator's rows.
 * Ring, knot detection, knot selection and the operator all run on the device: a device-form push / flush enqueues its
 * launches on `stream` and returns (no host synchronisation; `*emitted` is known from the block count alone); the host forms
 * copy through pinned staging and synchronise once per call.  A NaN in a window makes the host forms return
 * ITD_ERR_NONFINITE (the block is emitted under plain comparison rules); the device forms record it for itd_stream_status. */
typedef struct itd_stream itd_stream;   /* opaque; not thread-safe */
#define ITD_STREAM_CUBIC 0
#define ITD_STREAM_LINEAR 1
int itd_stream_create(itd_stream **out, int device_id, int64_t block /* samples, >= 8 */, int32_t channels, int32_t kind,
                      int32_t margin /* cubic: >= 1 */, int32_t shared_knots);
void itd_stream_destroy(itd_stream *s);
int itd_stream_reset(itd_stream *s);            /* forget the blocks held and the recorded status */
int64_t itd_stream_blocks(const itd_stream *s); /* blocks held since create / reset / flush */
const char *itd_stream_last_error(const itd_stream *s);
/* block_dev [channels] rows of `block` samples at in_stride; baseline_dev / rot_dev (optional) likewise, written when
 * *emitted = 1 (every push but the first; every flush of a non-empty stream) */
int itd_stream_push_f64(itd_stream *s, const double *block_dev, int64_t in_stride, double *baseline_dev, int64_t baseline_stride,
                        double *rot_dev, int64_t rot_stride, int32_t *emitted, void *stream);
int itd_stream_flush_f64(itd_stream *s, double *baseline_dev, int64_t baseline_stride, double *rot_dev, int64_t rot_stride,
                         int32_t *emitted, void *stream);
/* host forms: contiguous [channels][block] arrays; rot_host optional */
int itd_stream_push_host_f64(itd_stream *s, const double *block_host, double *baseline_host, double *rot_host, int32_t *emitted);
int itd_stream_flush_host_f64(itd_stream *s, double *baseline_host, double *rot_host, int32_t *emitted);
/* synchronises the device; *status = 0, or 2 if some window since create / reset held a NaN */
int itd_stream_status(itd_stream *s, int32_t *status);

/* ---- introspection for benchmarks -------------------------------------------------------------
 * hipEvent pairs on the launch stream.  Extraction launches are dispatched with their own start/stop events
 * (hipExtLaunchKernel: the events take the dispatch's own begin/end timestamps, so they agree with rocprofv3's kernel durations); the
 * whole-decomposition span uses two marker records.  itd_set_kernel_timing(e, K) enables recording for up to K
 * decompositions (K = 0 disables) and resets the tallies; itd_get_kernel_timing sums the elapsed time of every
 * recorded launch of class `which`. */
#define ITD_TIME_EXTRACT 0        /* k_extract<float64 in>: levels >= 1, the dominant kernel (24 B/sample) */
#define ITD_TIME_EXTRACT_L0 1     /* k_extract on the caller's signal (level 0; 20 B/sample for float32) */
#define ITD_TIME_EXTRACT_FINAL 2  /* k_extract of the "Out of time!" level (writes rotation+baseline only) */
#define ITD_TIME_DECOMPOSE 3      /* first launch .. last launch of one whole decomposition */
#define ITD_TIME_SCAN0 4          /* k_scan0: the level-0 knot scan of the caller's signal (4 B/sample for float32) */
#define ITD_TIME_KF_APPLY 5       /* k_kf_apply: the one pass over the samples for all fused levels (8 B read + 8 B per row written) */
#define ITD_TIME_KF_KNOTS 6       /* k_kf_knots: the knot side of the fused levels (hand-over and every level's step), one launch */
int itd_set_kernel_timing(itd_engine *e, int max_decompositions);
/* instrument only every stride-th decomposition (launches with events cost ~2 us more each, the span's marker records ~5 us each) */
int itd_set_kernel_timing_stride(itd_engine *e, int stride);
int itd_get_kernel_timing(itd_engine *e, int32_t which, double *ms_total, int32_t *launches);
/* the individual durations behind that sum (ms), in launch order: at most `cap` are written, *count = how many exist */
int itd_get_kernel_timing_samples(itd_engine *e, int32_t which, double *ms_out, int32_t cap, int32_t *count);
/* mode 1: only the level-0 launch of each decomposition carries events (one completion signal per step instead of one per launch);
 * itd_get_step_periods then gives the time from one decomposition's first launch to the next one's (ms): the per-step period of
 * back-to-back calls, from the dispatches' own timestamps.  mode 0 (default): every launch class. */
int itd_set_kernel_timing_mode(itd_engine *e, int32_t mode);
int itd_get_step_periods(itd_engine *e, double *ms_out, int32_t cap, int32_t *count);

#ifdef __cplusplus
}
#endif
#endif /* PYITD_HIP_H */
