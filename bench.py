#!/usr/bin/env python3
"""bench.py — headline benchmark of the ITD hot path on MI355X.

Metric (BASELINE.json): Msamples/s of the full 8-level ITD + achieved HBM GB/s.
One "step" = one complete decomposition (level-0 knot scan, 9 extractions, stop rules, row fix-up) of the synthetic
input already resident in HBM.

  N = 1   workload = BASELINE configs[1]: one 2^24-sample float32 sum-of-sines+noise signal, 8 levels
          (max_iteration = 7 -> 9 rows), one MI355X.
  N > 1   workload = BASELINE configs[3]'s shard: the batch of N x 1024 signals x 2^20 samples (signal b: noise draw
          b mod 16, frequencies scaled by 1 + b/8192), contiguous shards of 1024 signals per GPU
          (pyitd_amd.distributed.ShardedBatch), NO data-path collective; the only communication is the RCCL all-gather of
          the per-signal summaries.  Weak scaling: value = all ranks' samples / max-over-ranks time.

`python bench.py --gpus N` run plainly starts the N ranks itself: the parent process never touches the GPU, it spawns N
fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set per child) and relays rank 0's line.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE already in the environment) the
process IS one rank.

Printed JSON (one line, rank 0): the driver contract + "roofline" — the dominant kernel: with the sparse levels fused (the
default for this workload) k_kf_apply, the one pass over the samples for levels 2 .. 8 (8 B read + 7 x 8 B written per sample);
with --no-fuse k_extract<double> (24 B per sample and level) — timed with the launch's own hipEvents on the launch stream
inside the timed region, + the per-kernel fractions + "cpu_baseline" (the C oracle = single-thread port of the reference
algorithm, timed on this box's host at N = 1) and the other CPU legs SURVEY 8d lists (all host cores over independent signals,
numpy restatement, numba restatement).  A step whose fused levels refuse is repeated level by level when
its summary is read: the headline checks for that behind its timed region and times again level by level if it happened; the
batch legs and the N > 1 path read the summary inside every step.
The headline has an untimed warm-up of its own (--warm-ms of its own steps, then the W warm-up steps), so it does not depend on any
other leg.  At N = 1 the line also carries, all timed AFTER the headline and skipped by --no-extra: "config3_batch" (BASELINE
configs[2], 1024 x 2^20 signals), "short_signal_batches" (4096 x 4096 and 60 000 x 256 samples through the resident form and level by
level), "mid_size_signals" (single signals of 2^16 .. 2^20 samples, BASELINE configs[0]'s chirp among them), "f_rows" (the SURVEY 8f
operators: cubic baseline at 2^24, instantaneous step, totalextract2d on 512 x 512 beside the reference's recorded 10.1457 s, MEITD on a
golden signal, block-wise microseconds per block), "config5_audio" (BASELINE configs[4] on --wav PATH or the substitute clip) and
"headline_on_quantised_2p24".
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable copy rate)
LEVELS = 8                  # BASELINE configs[1]
MAX_ITERATION = LEVELS - 1  # -> LEVELS rotations + residual = 9 rows
LOG2N = 24
TIMING_STRIDE = int(os.environ.get("BENCH_TIMING_STRIDE", "4"))   # every n-th timed step carries hipEvents on its launches
SINES = ((1, 110, 0.1), (0.5, 440, 1.3), (0.25, 1760, 2.1), (0.125, 7040, 0.7))


def sines_noise(n, seed=0, fscale=1.0, fs=48000.0):
    """BASELINE config 2 recipe, float64 math cast to float32."""
    t = np.arange(n, dtype=np.float64) / fs
    x = np.zeros(n, dtype=np.float64)
    for a, f, p in SINES:
        x += a * np.sin(2 * np.pi * (f * fscale) * t + p)
    x += 0.05 * np.random.default_rng(seed).standard_normal(n)
    return x.astype(np.float32)


def batch_signals_device(torch, dev, b_lo, b_hi, n, fs=48000.0, rows_per_call=64):
    """Signals b_lo..b_hi-1 of the config 3 / 4 batch, synthesised on the GPU with the same recipe (noise draw b mod 16
    from numpy's default_rng on the host, sines in float64 on the device, cast to float32).  Values may differ from the host
    recipe in the last float64 ulp of sin(): this is synthetic bench input, the parity tests build theirs on the host."""
    noise = torch.from_numpy(np.stack([0.05 * np.random.default_rng(s).standard_normal(n) for s in range(16)])).to(dev)
    t = torch.arange(n, dtype=torch.float64, device=dev) / fs
    x = torch.empty((b_hi - b_lo, n), dtype=torch.float32, device=dev)
    for c0 in range(b_lo, b_hi, rows_per_call):
        c1 = min(c0 + rows_per_call, b_hi)
        b = torch.arange(c0, c1, dtype=torch.float64, device=dev)
        fsc = (1.0 + b / 8192.0)[:, None]
        acc = torch.zeros((c1 - c0, n), dtype=torch.float64, device=dev)
        for a, f, p in SINES:
            acc += a * torch.sin(2 * np.pi * (f * fsc) * t[None, :] + p)
        acc += noise[(torch.arange(c0, c1, device=dev) % 16)]
        x[c0 - b_lo:c1 - b_lo] = acc.to(torch.float32)
    return x


def algorithmic_bytes_per_sample(levels):
    """level 1 reads 4 B (f32) and writes 8+8; every later level reads 8 and writes 8+8."""
    return 20 + 24 * (levels - 1)


def own_bytes_per_sample(rows, first_fused_level):
    """What the engine itself moves per sample for a `rows`-row result: level 0 (20 B) and every level below the first fused one
    (24 B) as one launch each, then ONE pass for the fused levels: 8 B read + 8 B per row written.  first_fused_level 0 = one launch per
    level throughout (the reference's flow: the last level writes one row)."""
    if not first_fused_level:
        return 20.0 + 24.0 * (rows - 2) + 16.0 if rows >= 2 else 20.0
    L0 = int(first_fused_level)
    return 20.0 + 24.0 * (L0 - 1) + 8.0 + 8.0 * (rows - L0)


def library_build_id():
    """sha256 (first 16 hex digits) of the libpyitd_hip.so this process loads: tools/traffic.sh stamps its PMC figures with it, so a line
    can say whether profiles/traffic.json was measured on the build that is being timed."""
    import hashlib
    from pyitd_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    except OSError:
        return None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def preflight(args):
    """N > 1: fail fast, with a message that says what is missing, before any rank allocates — fewer visible devices than ranks;
    a per-rank footprint (rows + input + workspace) beyond the free HBM of this rank's device.  torch.cuda.device_count() does
    not initialise HIP on this image, so the plain launcher's parent stays free of the GPU."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d visible devices, this process sees %d (HIP_VISIBLE_DEVICES=%r, ROCR_VISIBLE_DEVICES=%r): "
                         "nothing was started.  On a one-GPU box the sharded path is rehearsed with --rehearse-one-gpu.\n"
                         % (args.gpus, args.gpus, have, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
        print(json.dumps({"error": "needs %d devices, %d visible" % (args.gpus, have), "n_gpus": args.gpus}))
        return 2
    if "WORLD_SIZE" in os.environ:          # a rank: its device's free memory against its shard's footprint
        n = 1 << (args.log2n if args.log2n != LOG2N else 20)
        if args.total_batch:                # strong scaling: the largest shard of the fixed batch
            args.batch = -(-args.total_batch // max(1, int(os.environ.get("WORLD_SIZE", "1"))))
        need = shard_footprint_bytes(args.batch, n)
        local = int(os.environ.get("LOCAL_RANK", "0"))
        free, total = torch.cuda.mem_get_info(local)
        if need > free:
            sys.stderr.write("bench.py rank %s: the shard of %d x 2^%d signals needs %.1f GB (rows %.1f + input %.1f + workspace %.1f), "
                             "device %d has %.1f GB free of %.1f: lower --batch\n"
                             % (os.environ.get("RANK"), args.batch, n.bit_length() - 1, need / 1e9, 8.0 * (MAX_ITERATION + 2) * n * args.batch / 1e9,
                                4.0 * n * args.batch / 1e9, (need - (8.0 * (MAX_ITERATION + 2) + 4.0) * n * args.batch) / 1e9, local, free / 1e9, total / 1e9))
            return 3
    return 0


def shard_footprint_bytes(batch, n):
    """HBM one rank's shard of `batch` signals of n samples takes: float64 rows [batch][M + 2][n], the float32 input, the engine's
    workspace (three rotating baseline slots = 24 B/sample, tile records ~0.4 B/sample, the fused levels' tables, flag words and
    first indices ~12 B/sample)."""
    return int(batch * n * (8.0 * (MAX_ITERATION + 2) + 4.0 + 24.0 + 0.4 + 12.0))


def topology_summary():
    """`rocm-smi --showtopo` in a few lines (link types and hop counts between the visible GPUs), for the N > 1 line's config."""
    try:
        out = subprocess.run(["rocm-smi", "--showtopotype", "--showtopohops"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if ln.strip() and not set(ln.strip()) <= set("=- ")]
        return keep[:40] if keep else None
    except Exception as ex:  # noqa: BLE001
        return "unavailable: %r" % (ex,)


def spawn_ranks(args):
    """The parent of a plain `bench.py --gpus N`: start N rank processes and relay rank 0's JSON line.  This process never
    initialises HIP (fresh children, no exec of a GPU-holding process).  Every rank's stderr is captured and relayed with a
    `[rank r]` prefix; the run has a deadline (--spawn-timeout): on expiry, or when any rank fails, the children this process
    started are killed and the exit code is non-zero."""
    import tempfile
    n = args.gpus
    port = _free_port()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        ef = tempfile.TemporaryFile()
        errs.append(ef)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=ef))
    deadline = time.time() + args.spawn_timeout
    out, rc, timed_out = b"", 0, False
    try:
        out, _ = procs[0].communicate(timeout=args.spawn_timeout)
        rc = procs[0].returncode
    except subprocess.TimeoutExpired:
        timed_out = True
    failed = []
    for r, p in enumerate(procs):
        if not timed_out and r > 0:
            try:
                p.wait(timeout=max(1.0, min(120.0, deadline - time.time())))
            except subprocess.TimeoutExpired:
                timed_out = True
        if timed_out and p.poll() is None:
            p.kill()          # the exact child this process started
            p.wait()
        if p.returncode:
            failed.append((r, p.returncode))
    for r, ef in enumerate(errs):
        ef.seek(0)
        for line in ef.read().decode(errors="replace").splitlines():
            sys.stderr.write("[rank %d] %s\n" % (r, line))
        ef.close()
    if timed_out:
        sys.stderr.write("bench.py: ranks did not finish within %d s (--spawn-timeout): killed\n" % args.spawn_timeout)
        return 124
    if failed:
        sys.stderr.write("bench.py: ranks failed: %s\n" % ", ".join("rank %d rc %d" % f for f in failed))
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc or (failed[0][1] if failed else 0)


class _StubEngine:
    """Stand-in for pyitd_amd.Engine in the CPU test of the launcher (tests/test_bench_launcher_cpu.py, --stub): no GPU,
    no compute; the summary encodes which signals the rank owned."""

    def __init__(self, lo, hi, rows):
        self.lo, self.hi, self.rows = lo, hi, rows

    def decompose_dev(self, *a):
        pass

    def set_timing(self, *a, **k):
        pass

    def summary(self, k):
        from pyitd_amd.distributed import MAX_ROWS
        kc = np.full((k, MAX_ROWS + 1), -1, np.int64)
        kc[:, 0] = np.arange(self.lo, self.hi)          # signal id in the first knot-count column
        return {"n_rows": np.full(k, self.rows, np.int32), "n_baselines": np.full(k, self.rows, np.int32),
                "stop": np.ones(k, np.int32), "nan_levels": np.full(k, -1, np.int32), "knot_counts": kc}


def run_rank(args):
    if args.no_fuse:
        os.environ["PYITD_FUSE_MODE"] = "1"     # every engine of this process: one launch per level throughout (pyitd_amd/engine.py)
    import torch
    import torch.distributed as dist

    # the GPU box gives a process a CPU quota (cgroup), not the host's 256 hardware threads: a pool of that many spinning
    # intra-op threads gets the whole process throttled for tens of milliseconds (tools/stall_probe.py); nothing here needs them
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = args.stub
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
    coll_backend = "gloo" if (stub or (args.rehearse_one_gpu and not args.try_rccl) or world == 1) else "nccl"
    if stub:
        dev = torch.device("cpu")
    else:
        if args.rehearse_one_gpu:
            local_rank = 0   # rehearsal on a one-GPU box: every rank computes on cuda:0, the summaries travel over gloo
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    group = None           # the group that carries the path's collectives (None = the default gloo group)
    if world > 1:
        import datetime
        # The control plane always comes up over gloo first (bounded: a rank that never arrives fails the others after the
        # timeout instead of hanging them).  RCCL is brought up as a SECOND group and probed; whether it works is agreed by an
        # all-reduce over gloo, so either every rank uses RCCL or every rank uses gloo — never a mix.
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.pg_timeout))
        if coll_backend == "nccl":
            ok, why = 1, ""
            try:
                group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=args.pg_timeout), device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe, group=group)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("RCCL all_reduce probe returned %r" % probe.item())
            except Exception as ex:  # noqa: BLE001
                ok, why = 0, repr(ex)[:160]
                sys.stderr.write("bench.py rank %d: RCCL did not come up (%s)\n" % (rank, why))
            agree = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)          # over gloo
            if int(agree.item()) != 1:
                group = None
                coll_backend = "gloo (fallback: RCCL did not come up on every rank%s)" % ((": " + why) if why else "")
    coll_dev = dev if coll_backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live

    from pyitd_amd.distributed import ShardedBatch

    M = MAX_ITERATION
    R = M + 2
    sharded = world > 1
    if sharded:
        n = 1 << (args.log2n if args.log2n != LOG2N else 20)
        per_gpu = args.batch
    else:
        n = 1 << args.log2n
        per_gpu = 1
    batch = per_gpu * world
    strong = bool(sharded and args.total_batch)
    if strong:
        # SURVEY 8d config 4 / BASELINE.md section 3: strong scaling at a FIXED batch (8192 signals over 8 GPUs; 2 and 4 GPUs take the
        # same batch in larger shards) — contiguous balanced shards (shard_range: the first batch % world ranks own one signal more)
        batch = args.total_batch
        per_gpu = -(-batch // world)
    if stub:
        from pyitd_amd.distributed import shard_range
        lo, hi = shard_range(batch, world, rank)
        sb = ShardedBatch(batch, n, M, world, rank, engine=_StubEngine(lo, hi, R))
        x = rows = None
        x_ptr = rows_ptr = sp = 0
        x_host = None
    else:
        import pyitd_amd
        sb = ShardedBatch(batch, n, M, world, rank, device=local_rank)
        if sharded:
            x = batch_signals_device(torch, dev, sb.lo, sb.hi, n)
            x_host = None
        else:
            x_host = sines_noise(n, seed=0)
            x = torch.from_numpy(x_host).to(dev)[None]
        rows = torch.empty((sb.n_local, R, n + int(os.environ.get("BENCH_ROW_PAD", "0"))), dtype=torch.float64, device=dev)   # (experiment builds: -DITD_ROW_PAD)
        stream = torch.cuda.Stream(device=dev)
        sp = stream.cuda_stream
        x_ptr, rows_ptr = x.data_ptr(), rows.data_ptr()
    eng = sb.engine

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        if dist.is_initialized():
            dist.barrier(group=group)

    def step():
        sb.decompose(x_ptr, np.float32, n, rows_ptr, None, sp)
        if sharded:
            # a batch's results are complete once its summary has been read: the few signals whose fused sparse levels refuse (about
            # one in a hundred of these 2^20-sample signals: a knot born from rounding at a deep level) are re-run there, on their
            # own — that work belongs inside the timed region (costs one synchronisation and a 0.4 MB copy per step)
            sb.local_summary()

    # Untimed warm-up inside bench.py, independent of every other leg: the headline's own step for at least --warm-ms (default
    # 150 ms) — an idle GPU needs tens of milliseconds of sustained work to reach the clocks it then holds, and the 11 ms of
    # a 20-step timed region are too short to get there (rounds 1-2 relied on the informational batch leg running first: 0.582
    # vs 0.557 ms per step with --no-extra).  Then the driver's W warm-up steps.
    # Everything that takes host time with the GPU idle happens BEFORE the last warm-up steps (creating the instrumentation's events:
    # ~1000 hipEventCreate calls on the first use; reading the summary), so that the timed region follows them directly: a 20-step
    # region is 9 ms, and a GPU left idle for a millisecond or two in front of it ran its first steps below its sustained clocks
    # (measured: 0.470-0.484 ms per step at 20 steps against 0.458-0.461 at 200 with the idle gap).
    if not sharded and not stub:
        eng.set_timing(args.steps, stride=TIMING_STRIDE)      # (allocates the events; re-armed below)
        eng.set_timing(0)
    # (the summary first: reading it leaves the GPU idle for a few hundred microseconds, and 20 steps timed a millisecond after
    #  such a gap ran 5-7 % below the rate the same build sustains — 0.423 against 0.396 ms, profiles/r04 — so nothing but steps
    #  lies between the warm-up below and the timed region)
    step()
    sync()
    summ = sb.local_summary()
    t_w = time.perf_counter()
    while not stub and (time.perf_counter() - t_w) * 1e3 < args.warm_ms:
        for _ in range(8):
            step()
        sync()
    # on every 4th step of the timed region the launches carry their own hipEvent pair (hipExtLaunchKernel: the dispatch's
    # begin/end timestamps, the same thing rocprofv3 reports); such a launch costs ~2 us more, hence the stride
    for _ in range(args.warmup):
        step()
    if not sharded and not stub:
        eng.set_timing(args.steps, stride=TIMING_STRIDE)
    def timed_region():
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        barrier()
        return time.perf_counter() - t0

    fused_refused = False
    rep0 = 0 if stub else eng.fuse_repeats
    fix0 = 0 if stub else eng.fuse_signal_repairs
    elapsed_local = timed_region()
    if not stub:
        # A decomposition whose fused sparse levels failed their verification is REPEATED level by level when its summary is read
        # (itd_get_summary) — i.e. behind the timed region.  Such a timing would leave work out: read the summary now, and if a
        # repeat happened (on any rank's shard) time the steps again — the engine then runs them level by level by itself.
        sb.local_summary()
        fused_refused = eng.fuse_repeats > rep0
        if dist.is_initialized():
            fr = torch.tensor([1 if fused_refused else 0], dtype=torch.int32)
            dist.all_reduce(fr, op=dist.ReduceOp.MAX)
            fused_refused = bool(fr.item())
        if fused_refused:
            from pyitd_amd.engine import FUSE_OFF
            eng.set_fuse_mode(FUSE_OFF)
            if not sharded:
                eng.set_timing(args.steps, stride=TIMING_STRIDE)
            elapsed_local = timed_region()
    elapsed = elapsed_local
    per_rank_ms = [elapsed_local / args.steps * 1e3]
    if dist.is_initialized():
        tt = torch.tensor([elapsed_local], dtype=torch.float64, device=coll_dev)
        parts = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(parts, tt, group=group)
        per_rank_ms = [float(p.item()) / args.steps * 1e3 for p in parts]
        elapsed = max(float(p.item()) for p in parts)

    # the only inter-GPU traffic of the path: the per-signal summaries (a few hundred bytes per signal), all-gathered — timed on its
    # own, apart from the compute (a second gather: the first one pays the collective's set-up)
    table, gather_ms, repairs_by_rank = None, None, None
    if dist.is_initialized():
        table = sb.gather(group=group, device=coll_dev)
        barrier()
        tg = time.perf_counter()
        table = sb.gather(group=group, device=coll_dev)
        sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
        mine = torch.tensor([0.0 if stub else (eng.fuse_signal_repairs - fix0) / max(args.steps, 1), 0.0 if stub else float(eng.fuse_repeats - rep0),
                             gather_ms], dtype=torch.float64, device=coll_dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        repairs_by_rank = [[round(float(v), 3) for v in p.tolist()] for p in parts]

    if not sharded and not stub:
        from pyitd_amd.engine import (TIME_DECOMPOSE, TIME_EXTRACT, TIME_EXTRACT_FINAL, TIME_EXTRACT_L0, TIME_KF_APPLY, TIME_KF_KNOTS,
                                      TIME_SCAN0)
        timing = {k: eng.kernel_timing(t) for k, t in (("ext", TIME_EXTRACT), ("l0", TIME_EXTRACT_L0), ("fin", TIME_EXTRACT_FINAL),
                                                        ("dec", TIME_DECOMPOSE), ("scan0", TIME_SCAN0), ("kfa", TIME_KF_APPLY), ("kfk", TIME_KF_KNOTS))}
        fuse_repeats = eng.fuse_repeats
        samples = {k: eng.kernel_timing_samples(t) * 1e3 for k, t in (("ext", TIME_EXTRACT), ("l0", TIME_EXTRACT_L0), ("kfa", TIME_KF_APPLY),
                                                                       ("kfk", TIME_KF_KNOTS), ("dec", TIME_DECOMPOSE))}
        eng.set_timing(0)
        # The distribution of the per-step time: >= 64 further back-to-back steps in which only each step's level-0 launch carries
        # events (one completion signal per step); the period from one step's first dispatch to the next one's, device timestamps
        step_dist = None
        try:
            extra = max(64, args.steps)
            eng.set_timing_mode(1)
            eng.set_timing(extra + 1, stride=1)
            for _ in range(extra + 1):
                step()
            sync()
            per = eng.step_periods() * 1e3
            eng.set_timing(0)
            eng.set_timing_mode(0)
            if len(per) >= 8:
                step_dist = {"steps": int(len(per)), "mean_us": round(float(per.mean()), 2), "median_us": round(float(np.median(per)), 2),
                             "p10_us": round(float(np.percentile(per, 10)), 2), "p90_us": round(float(np.percentile(per, 90)), 2),
                             "min_us": round(float(per.min()), 2), "max_us": round(float(per.max()), 2),
                             "how": "a second region of back-to-back steps behind the timed one; period between consecutive steps' first "
                                    "dispatches (hipExtLaunchKernel start events on the level-0 launch only)"}
        except Exception as ex:  # noqa: BLE001
            step_dist = {"error": repr(ex)[:160]}
    else:
        timing = None
        samples = None
        step_dist = None

    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return 0

    samples_per_step = n * batch          # (strong scaling: the fixed batch; weak: per_gpu x world)
    value = samples_per_step * args.steps / elapsed / 1e6
    ms_per_step = elapsed / args.steps * 1e3
    if sharded:
        workload = ("batch of %s signals x 2^%d float32 samples (sines+noise, draw b mod 16, f*(1+b/8192)), %d ITD levels "
                    "(max_iteration=%d, %d rows), contiguous shards of %s%d signals per GPU, device resident"
                    % (("%d (fixed: strong scaling)" % batch) if strong else ("%d x %d" % (world, per_gpu)), n.bit_length() - 1, LEVELS, M, R,
                       "at most " if strong else "", per_gpu))
    else:
        workload = ("single 2^%d-sample float32 sum-of-sines+noise, %d ITD levels (max_iteration=%d, %d rows), device resident"
                    % (args.log2n, LEVELS, M, R))
    out = {
        "metric": "Msamples/s full ITD (8 levels) + achieved HBM GB/s",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "signals_per_gpu": per_gpu,
            "samples_per_signal": n,
            "rows": int(summ["n_rows"][0]),
            "knots_per_level": [int(v) for v in summ["knot_counts"][0] if v >= 0],
            "sharding": "contiguous ranges of independent signals per GPU (ShardedBatch), no data-path collective; "
                        "all-gather of the per-signal summaries only",
            "collective_backend": None if world == 1 else ("nccl (RCCL)" if coll_backend == "nccl" else coll_backend),
            "device": None if stub else device_info(torch, dev),
            "xgmi_topology": topology_summary() if (world > 1 and not stub and not args.rehearse_one_gpu) else None,
            "shard_footprint_GB": None if (stub or not sharded) else round(shard_footprint_bytes(per_gpu, n) / 1e9, 1),
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "fused_levels_refused_and_retimed_level_by_level": fused_refused,
            "summary_read_every_step": bool(sharded),
            "signals_rerun_on_their_own_per_step_rank0": None if (stub or not sharded) else round((eng.fuse_signal_repairs - fix0) / max(args.steps, 1), 2),
            "summary_allgather_ms": None if gather_ms is None else round(gather_ms, 3),
            "per_rank_signals_rerun_per_step__whole_call_repeats__allgather_ms": repairs_by_rank,
            "rows_all_ranks": None if table is None else sorted(set(int(v) for v in table["n_rows"])),
            "signals_in_gathered_table": None if table is None else int(len(table["n_rows"])),
        },
        # the reference's level-by-level data flow (20 + 24 x 7 = 188 B/sample) over the measured time: an EQUIVALENT rate —
        # what a level-by-level engine would have to sustain to match this time — not bytes this engine moves (with levels 3..8 fused
        # it moves 108 B/sample fused from level 2, 124 from level 3: roofline.per_kernel_frac carries that figure)
        "hbm_reference_flow_equivalent_GBps": round(algorithmic_bytes_per_sample(LEVELS) * samples_per_step * args.steps / elapsed / 1e9, 1),
    }
    if args.rehearse_one_gpu:
        out["config"]["rehearsal"] = "all %d ranks shared cuda:0, summaries over gloo: a check of the sharded path, not a scaling measurement" % world
    if stub:
        out["config"]["stub"] = True
        out["config"]["table_signal_ids"] = [int(v) for v in table["knot_counts"][:, 0]] if table is not None else None
    if timing is not None:
        def avg_us(k):
            ms, cnt = timing[k]
            return ms / max(cnt, 1) * 1e3
        ext_cnt = timing["ext"][1]
        ext_us = avg_us("ext")
        achieved = 24.0 * n / (ext_us * 1e-6) / 1e9 if ext_cnt else 0.0
        traffic, traffic_src, traffic_stale = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_extract_f64_bytes_per_launch")
                traffic_stale = tj.get("library_build_id") != library_build_id()
                traffic_src = "profiles/traffic.json (%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 correction; " \
                              "a recorded measurement of %s, not of this run)" % (
                                  tj.get("round"), "an OLDER build of the library (traffic_stale)" if traffic_stale else "this very build of the library (build ids equal)")
            except Exception:
                traffic = None
        live = LIVE_TRAFFIC if (LIVE_TRAFFIC and "error" not in LIVE_TRAFFIC) else None
        if live and live.get("k_extract_f64"):
            traffic, traffic_src, traffic_stale = live["k_extract_f64"], live["source"], False

        def frac(bytes_per_sample, us):
            return round(bytes_per_sample * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if us > 0 else 0.0
        fused = timing["kfa"][1] > 0
        rows_out = int(summ["n_rows"][0])
        if fused:
            # The sparse levels ran fused (pyitd_amd/csrc/itd_knotfirst.hpp): levels 0 .. L0-1 one launch each, then ONE pass over the
            # samples for levels L0 .. (k_kf_apply) — the launch that dominates.  Its algorithmic bytes: 8 B read + 8 B per row written.
            L0 = 1 + timing["ext"][1] // max(timing["l0"][1], 1)          # extraction launches of levels >= 1 per decomposition, + level 0
            apply_bytes = 8.0 + 8.0 * (rows_out - L0)
            apply_us = avg_us("kfa")
            own_bytes = 20.0 + 24.0 * (L0 - 1) + apply_bytes
            traffic_a = None
            try:
                traffic_a = json.load(open(tpath)).get("k_kf_apply_bytes_per_launch")
            except Exception:
                pass
            if live and live.get("k_kf_apply"):
                traffic_a, traffic_src, traffic_stale = live["k_kf_apply"], live["source"], False
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "k_kf_apply (levels %d..%d of every tile in one pass, in registers: read 8 B + write 8 B x %d rows per sample)"
                          % (L0, rows_out - 1, rows_out - L0),
                "achieved": round(apply_bytes * n / (apply_us * 1e-6) / 1e9, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": frac(apply_bytes, apply_us),
                "traffic": traffic_a,
                "traffic_source": traffic_src if traffic_a else None,
                "traffic_stale": traffic_stale if traffic_a else None,
                "traffic_live_attempt": None if LIVE_TRAFFIC is None else (LIVE_TRAFFIC.get("error") or "ok"),
                "traffic_over_algorithmic": round(traffic_a / (apply_bytes * n), 3) if traffic_a else None,
                "library_build_id": library_build_id(),
                "algorithmic_bytes_per_sample": apply_bytes,
                # SURVEY 8d asks for both fractions: the bytes the launch READS over the peak (8 B/sample: the first fused level's input)
                # and all its bytes (frac).  The north star's ">= 40 % of the HBM-read roofline" is out of reach for a pass that writes
                # 8 B per row for every 8 B it reads; frac is the figure to read.
                "read_frac": frac(8.0, apply_us),
                "avg_launch_us": round(apply_us, 2),
                "launches_timed": timing["kfa"][1],
                "reference_flow_bytes_per_sample": 24.0 * (rows_out - L0),
                "note": "the reference's level-by-level data flow moves 24 B per sample and level for these levels (SURVEY 8d: %d B/sample); "
                        "the fused pass needs %d — frac is its own bytes over the peak" % (24 * (rows_out - L0), int(apply_bytes)),
                "level0_launch_us": round(avg_us("l0"), 2),
                "extract_launch_us": round(ext_us, 2),
                "knot_side_us": round(avg_us("kfk"), 2),
                "decompose_gpu_us": round(avg_us("dec"), 2),
                "launch_us_sigma": {k: round(float(v.std()), 2) for k, v in (("k_kf_apply", samples["kfa"]), ("level0", samples["l0"]),
                                                                             ("k_extract_f64", samples["ext"]), ("k_kf_knots", samples["kfk"])) if len(v)},
                "fuse_repeats": fuse_repeats,
                "per_kernel_frac": {
                    "k_kf_apply (%d B/sample)" % int(apply_bytes): frac(apply_bytes, apply_us),
                    "k_extract<double> levels 1..%d (24 B/sample)" % (L0 - 1): frac(24.0, ext_us),
                    "k_extract<float> level 0 (20 B/sample)": frac(20.0, avg_us("l0")),
                    "whole decomposition, own bytes (%d B/sample)" % int(own_bytes): frac(own_bytes, avg_us("dec")),
                },
                "whole_decomposition_reference_flow_equivalent_GBps": round(float(algorithmic_bytes_per_sample(LEVELS)) * n / (avg_us("dec") * 1e-6) / 1e9, 1),
            }
            out["config"]["launch_form"] = ("levels 0..%d one launch each; levels %d..%d fused: the knot-side recursion in one launch (k_kf_knots) + one "
                                            "verified sample pass (k_kf_apply)" % (L0 - 1, L0, rows_out - 1))
        else:
            out["roofline"] = {
                "bound": "hbm",
                "kernel": "k_extract<double> (levels>=1: read 8 B + write 16 B per sample)",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_stale": traffic_stale if traffic else None,
                "library_build_id": library_build_id(),
                "avg_launch_us": round(ext_us, 2),
                "launches_timed": ext_cnt,
                "read_frac": frac(8.0, ext_us) if ext_cnt else 0.0,
                "read_frac_note": "the north star's '>= 40 % of HBM-read roofline' read as 8 B/sample / t / peak (SURVEY 8d) needs t <= 42 us "
                                  "per 2^24-sample level, i.e. 24 B/sample of total traffic at 9.6 TB/s: above the 8 TB/s peak, unreachable "
                                  "for any kernel that also writes the two float64 rows; frac (all 24 algorithmic bytes) is the figure to read",
                "level0_launch_us": round(avg_us("l0"), 2),
                "scan0_launch_us": round(avg_us("scan0"), 2),
                "final_launch_us": round(avg_us("fin"), 2),
                "decompose_gpu_us": round(avg_us("dec"), 2),
                "fuse_repeats": fuse_repeats,
                "per_kernel_frac": {
                    "extract_levels_ge1 (24 B/sample)": frac(24.0, ext_us),
                    "level0 pair: k_scan0 + k_extract<float> (20 B/sample)": frac(20.0, avg_us("scan0") + avg_us("l0")),
                    "k_extract<float> level 0 alone (20 B/sample)": frac(20.0, avg_us("l0")),
                    "k_scan0 (4 B/sample)": frac(4.0, avg_us("scan0")),
                    "final extract (16 B/sample)": frac(16.0, avg_us("fin")),
                    "whole decomposition (188 B/sample)": frac(float(algorithmic_bytes_per_sample(LEVELS)), avg_us("dec")),
                },
            }
            out["config"]["launch_form"] = "one launch per level"
    if step_dist is not None:
        out["step_time_distribution"] = step_dist
    if sharded and not stub:
        # N > 1: every GPU runs its shard as chunks whose launches overlap on two streams, so a single launch's duration is not
        # a rate; the figure is the whole decomposition's algorithmic bytes (188 B/sample) per GPU over the max-over-ranks time.
        # The per-launch figure of the dominant kernel is the N = 1 line's.
        # own bytes: fused from level 2 (the default for these shards) the engine moves 20 + 24 + (8 + 8 x 7) = 108 B per sample; level by level
        # (--no-fuse, or a refused and re-timed run) the reference flow's 188
        ran_fused = not fused_refused and not args.no_fuse
        lvl = eng.last_fuse_level if ran_fused else 0
        own_bytes = own_bytes_per_sample(R, lvl)
        per_gpu_gbps = own_bytes * float(n) * sb.n_local * args.steps / elapsed / 1e9     # (rank 0's shard: the largest one)
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "whole decomposition per GPU, own bytes (%d B/sample: %s); per-launch figures: the N = 1 line"
                      % (int(own_bytes), ("levels 0-%d one launch each, levels %d-8 fused" % (lvl - 1, lvl)) if lvl else "one launch per level"),
            "achieved": round(per_gpu_gbps, 1),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(per_gpu_gbps / HBM_PEAK_GBPS, 4),
            "traffic": None,
            "per_gpu": True,
            "reference_flow_equivalent_GBps_per_gpu": round(algorithmic_bytes_per_sample(LEVELS) * float(n) * sb.n_local * args.steps / elapsed / 1e9, 1),
        }
        out["cpu_baseline"] = None   # timed on rank 0 at N = 1 only (the N = 1 line of the same build carries all four CPU legs)
    if world == 1 and not stub and not args.no_cpu_baseline:
        out.update(cpu_legs(x_host, n, M, summ, rows, args))
    if world == 1 and not stub and not args.no_extra and args.log2n == LOG2N:
        # informational legs, all timed AFTER the headline (never at its cost: each in its own try)
        del rows, x
        torch.cuda.empty_cache()
        for key, leg in (("config3_batch", lambda: batch_leg(torch, dev)), ("short_signal_batches", lambda: short_signal_leg(torch, dev)),
                         ("mid_size_signals", lambda: mid_size_leg(torch, dev)), ("many_mid_size_signals", lambda: many_mid_size_leg(torch, dev)),
                         ("f_rows", lambda: f_rows_leg(torch, dev))):
            try:
                out[key] = leg()
            except Exception as ex:  # noqa: BLE001
                out[key] = {"error": repr(ex)[:200]}
    wav = args.wav or os.environ.get("PYITD_WAV")
    if world == 1 and not stub and (wav or not args.no_extra):
        # BASELINE configs[4]: the operator's wav, or the substitute clip the tests use (tests/golden/radio8000_input.npz)
        try:
            out["config5_audio"] = audio_leg(torch, dev, wav)
        except Exception as ex:  # noqa: BLE001
            out["config5_audio"] = {"error": repr(ex)[:200], "input": wav or "substitute clip"}
    if world == 1 and not stub and not args.no_extra and args.log2n == LOG2N:
        # the headline signal rounded to 16 bits: exact ties everywhere, the fused sparse levels refuse it — what the refusal path costs
        try:
            out["headline_on_quantised_2p24"] = quantised_leg(torch, dev, x_host, n, M)
        except Exception as ex:  # noqa: BLE001
            out["headline_on_quantised_2p24"] = {"error": repr(ex)[:200]}
        try:
            out["headline_with_valid_flags"] = valid_flags_leg(torch, dev, x_host, n, M)
        except Exception as ex:  # noqa: BLE001
            out["headline_with_valid_flags"] = {"error": repr(ex)[:200]}
    print(json.dumps(out))
    sys.stdout.flush()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


def device_info(torch, dev):
    """What the line was measured on: device name, visibility mask, RCCL version (for the first real N > 1 runs)."""
    info = {"name": torch.cuda.get_device_name(dev), "index": dev.index,
            "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
            "hip": getattr(torch.version, "hip", None)}
    try:
        info["uuid"] = str(torch.cuda.get_device_properties(dev).uuid)
    except Exception:  # noqa: BLE001
        pass
    try:
        info["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        info["rccl"] = None
    return info


def load_wav_mono(path):
    """BASELINE.md section 3 / SURVEY 8d config 5: mono or first channel, float32 in [-1, 1]."""
    from scipy.io import wavfile          # bench / tests only
    sr, a = wavfile.read(path)
    if a.ndim > 1:
        a = a[:, 0]
    if a.dtype.kind == "i":
        a = a.astype(np.float64) / float(np.iinfo(a.dtype).max + 1)
    elif a.dtype.kind == "u":             # 8-bit wavs are unsigned
        a = (a.astype(np.float64) - 128.0) / 128.0
    return sr, np.asarray(a, dtype=np.float32)


def quantised_leg(torch, dev, x_host, n, M, calls=20):
    """The headline's 2^24-sample signal rounded to 16 bits (int16 PCM as float32: the reference's own domain, PyITD.ipynb cell 2):
    exact ties everywhere, so the fused sparse levels refuse it and the call runs level by level.  Every call reads its summary
    (that is where a refused call is repeated): ms per complete decomposition, host round trip included, in the steady state the
    engine settles into; then the same with the repair enqueued on the device (itd_set_device_repair)."""
    import pyitd_amd
    q = (np.round(x_host.astype(np.float64) / np.abs(x_host).max() * 32767.0) / 32768.0).astype(np.float32)
    ties = int(np.count_nonzero(q[1:] == q[:-1]))
    x = torch.from_numpy(q).to(dev)
    rows = torch.empty((M + 2, n), dtype=torch.float64, device=dev)
    out = {"workload": "the headline signal rounded to 16 bits (x / max|x| * 32767, rounded, / 32768; float32), %d levels" % (M + 1),
           "exact_ties_of_neighbouring_samples": ties}
    for key, repair in (("summary_every_call", False), ("device_side_repair", True)):
        eng = pyitd_amd.Engine(n, 1, dev.index or 0)
        eng.set_device_repair(repair)
        valid = torch.zeros((1,), dtype=torch.int32, device=dev)
        if repair:
            eng.set_valid_flags(valid.data_ptr())
        torch.cuda.synchronize()
        first = None
        for i in range(3):
            t0 = time.perf_counter()
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
            s = eng.summary(1)
            if first is None:
                first = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
            if not repair:
                s = eng.summary(1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        s = eng.summary(1)
        out[key] = {"ms_per_decomposition": round(dt * 1e3, 4), "Msamples_per_s": round(n / dt / 1e6, 1),
                    "first_call_ms": round(first, 3), "fuse_repeats": eng.fuse_repeats, "device_repairs": eng.device_repairs,
                    "valid_flag": int(valid.item()) if repair else None,
                    "hbm_reference_flow_equivalent_GBps": round(algorithmic_bytes_per_sample(M + 1) * n / dt / 1e9, 1),
                    "rows": int(s["n_rows"][0])}
        eng.close()
    return out


def valid_flags_leg(torch, dev, x_host, n, M, calls=40):
    """The headline signal the way a stream-ordered consumer runs it: itd_set_valid_flags alone — behind every decomposition one short
    launch (k_verdict) writes valid_dev[0], nobody reads a summary between the calls.  What such a consumer pays when nothing refuses
    (the headline's own timed region reads no summary either, but carries no verdict launch)."""
    import pyitd_amd
    x = torch.from_numpy(x_host).to(dev)
    rows = torch.empty((M + 2, n), dtype=torch.float64, device=dev)
    valid = torch.zeros((1,), dtype=torch.int32, device=dev)
    eng = pyitd_amd.Engine(n, 1, dev.index or 0)
    eng.set_valid_flags(valid.data_ptr())
    torch.cuda.synchronize()
    for _ in range(10):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
    eng.summary(1)
    for _ in range(60):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / calls
    ok = int(valid.item())
    s = eng.summary(1)
    out = {"workload": "the headline's signal, itd_set_valid_flags only (no summary between the calls), %d calls" % calls,
           "ms_per_decomposition": round(dt * 1e3, 4), "Msamples_per_s": round(n / dt / 1e6, 1), "valid_flag": ok,
           "fuse_repeats": eng.fuse_repeats, "rows": int(s["n_rows"][0]), "first_fused_level": eng.last_fuse_level}
    eng.close()
    return out


def many_mid_size_leg(torch, dev, M=MAX_ITERATION):
    """MANY independent mid-size signals in flight (the north star's one-signal-per-stream; siftED2D.ipynb cell 1 runs rows under
    prange): 64 and 256 signals of 2^16 / 2^18 samples, 8 levels, (i) as ONE batch call (the engine's chunks over its own streams:
    default geometry, and the best of a sweep over itd_set_batch_streams x itd_set_batch_chunk), (ii) one signal per call over a pool
    of 4 / 8 / 16 engines, each with a stream of its own, driven by host threads.  Gsamples/s and the fraction of the HBM peak on the
    result's own bytes (input once, every produced row once: 4 + 8 x rows B/sample)."""
    import pyitd_amd
    from concurrent.futures import ThreadPoolExecutor
    out = []
    for batch, log2n in ((64, 16), (256, 16), (64, 18), (256, 18)):
        n = 1 << log2n
        x = batch_signals_device(torch, dev, 0, batch, n)
        rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        item = {"signals": batch, "samples_per_signal": n}
        own = None

        def time_batch(streams, chunk, reps=10):
            eng = pyitd_amd.Engine(n, batch, dev.index or 0)
            if streams:
                eng.set_batch_streams(streams)
            if chunk:
                eng.set_batch_chunk(chunk)
            # (warm-up by time, not by count: three calls are ~1 ms, and an idle GPU needs tens of ms to reach its sustained clocks — the
            #  first figure of a shape, the default geometry's, read 8 % slower than the same geometry inside the sweep behind it)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.03:
                for _ in range(3):
                    eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
                s = eng.summary(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            s = eng.summary(batch)
            dt = (time.perf_counter() - t0) / reps
            lvl, rep = eng.last_fuse_level, eng.fuse_repeats + eng.fuse_signal_repairs
            eng.close()
            return dt, s, lvl, rep
        dt, s, lvl, rep = time_batch(0, 0)
        own = (4.0 * n * batch + 8.0 * n * float(s["n_rows"].sum()))
        item["one_batch_call_default"] = {"us": round(dt * 1e6, 1), "Gsamples_per_s": round(batch * n / dt / 1e9, 2),
                                          "own_bytes_frac_of_peak": round(own / dt / 1e9 / HBM_PEAK_GBPS, 4), "first_fused_level": lvl,
                                          "repeats_or_repairs": rep}
        best = None
        sweep = {}
        for streams in (1, 2, 3, 4):
            for chunk in sorted(set(c for c in (max(1, batch // 16), max(1, batch // 8), max(1, batch // 4), max(1, batch // 2), batch))):
                d2, _, l2, _ = time_batch(streams, chunk, reps=6)
                sweep["%d streams x %d signals per sequence" % (streams, chunk)] = round(d2 * 1e6, 1)
                if best is None or d2 < best[0]:
                    best = (d2, streams, chunk, l2)
        item["one_batch_call_best_of_sweep"] = {"us": round(best[0] * 1e6, 1), "streams": best[1], "signals_per_sequence": best[2],
                                                "Gsamples_per_s": round(batch * n / best[0] / 1e9, 2),
                                                "own_bytes_frac_of_peak": round(own / best[0] / 1e9 / HBM_PEAK_GBPS, 4), "first_fused_level": best[3]}
        item["sweep_us"] = sweep
        pools = {}
        for P in (4, 8, 16):
            engs = [pyitd_amd.Engine(n, 1, dev.index or 0) for _ in range(P)]

            def work(k):
                e = engs[k]
                for b in range(k, batch, P):
                    e.decompose_dev(x[b].data_ptr(), np.float32, n, 1, n, M, rows[b].data_ptr(), None, None)
                return k
            with ThreadPoolExecutor(max_workers=P) as ex:
                list(ex.map(work, range(P)))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    list(ex.map(work, range(P)))
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
            for e in engs:
                e.summary(1)
                e.close()
            pools["%d engines" % P] = {"us": round(dt * 1e6, 1), "Gsamples_per_s": round(batch * n / dt / 1e9, 2),
                                       "own_bytes_frac_of_peak": round(own / dt / 1e9 / HBM_PEAK_GBPS, 4)}
        item["one_signal_per_call_engine_pool"] = pools
        out.append(item)
        del x, rows
    return {"workload": "independent float32 signals (the batch recipe), %d levels, device resident; every figure = all signals once" % (M + 1),
            "shapes": out}


def audio_leg(torch, dev, path, log2n=22, max_iteration=9):
    """BASELINE configs[4] on real audio: the wav (or, without one, the substitute clip of the tests) tiled to 2^22 samples
    (numpy.resize), 10 levels, device resident; knot indices of every level bit-exact against the CPU oracle (the checker, outside
    the timed calls)."""
    import pyitd_amd
    from oracle import cpu_oracle
    if path:
        sr, a = load_wav_mono(path)
        label = os.path.basename(path)
    else:
        a = np.load(os.path.join(ROOT, "tests", "golden", "radio8000_input.npz"))["x"].astype(np.float32)
        sr, label = 8000, "substitute clip (tests/golden/radio8000_input.npz: the reference's demo clip is not redistributable; --wav overrides)"
    n = 1 << log2n
    x_host = np.resize(a, n).astype(np.float32)
    R = max_iteration + 2
    x = torch.from_numpy(x_host).to(dev)
    rows = torch.empty((R, n), dtype=torch.float64, device=dev)
    bases = torch.empty((R, n), dtype=torch.float64, device=dev)
    eng = pyitd_amd.Engine(n, 1, dev.index or 0)
    torch.cuda.synchronize()
    for _ in range(4):          # (every warm-up call reads its summary: the engine's back-offs — a later first fused level, level by
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, max_iteration, rows.data_ptr(), bases.data_ptr(), None)   # level for a while —
        s = eng.summary(1)      #  settle before the timed calls, as they do in a workload that keeps coming)
    t_w = time.perf_counter()   # (... and 30 ms of the timed form itself: the CPU oracle of the leg in front left the GPU idle, at idle clocks)
    while time.perf_counter() - t_w < 0.03:
        for _ in range(10):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, max_iteration, rows.data_ptr(), None, None)
        torch.cuda.synchronize()
    s = eng.summary(1)
    rep_warm = eng.fuse_repeats
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, max_iteration, rows.data_ptr(), None, None)
    s = eng.summary(1)
    dt = (time.perf_counter() - t0) / 10
    nr = int(s["n_rows"][0])
    ref = cpu_oracle.itd_lean(x_host, max_iteration, want_knots=True)
    knots_ok = ref["rows"].shape[0] == nr
    from pyitd_amd.itd import _engine_for
    helper = _engine_for(n, dev.index or 0)
    level_in = x_host.astype(np.float64)
    for j in range(min(nr, len(ref["knots"]))):       # level j's input: the signal, then the stored baselines
        got = helper.detect_host(level_in, 0)
        knots_ok = knots_ok and np.array_equal(got, ref["knots"][j])
        level_in = bases[j].cpu().numpy()
    rows_ok = bool(np.array_equal(rows[:nr].cpu().numpy().view(np.uint64), ref["rows"].view(np.uint64)))
    # the bytes the timed form itself moves per sample (rows only, no baselines buffer): level launches 20 / 24 B (the last one writes one
    # row: 16), the fused levels 8 B in + 8 B per row, a capped call also the baseline it hands on (8 B) and the scan of it (8 B)
    L0f, capf = eng.last_fuse_level, eng.last_fuse_cap
    if not L0f:
        own_bytes = 20.0 + 24.0 * (nr - 2) + 16.0
    elif not capf:
        own_bytes = 20.0 + 24.0 * (L0f - 1) + 8.0 + 8.0 * (nr - L0f)
    else:
        own_bytes = 20.0 + 24.0 * (L0f - 1) + 8.0 + 8.0 * (capf - L0f) + 8.0 + 8.0 + 24.0 * max(nr - 1 - capf, 0) + 16.0
    out = {"input": label, "sample_rate": int(sr), "samples_in_file": int(a.shape[0]),
           "exact_ties_of_neighbouring_samples": int(np.count_nonzero(x_host[1:] == x_host[:-1])),
           "workload": "clip tiled to 2^%d float32 samples (numpy.resize), %d levels (max_iteration=%d)" % (log2n, max_iteration + 1, max_iteration),
           "rows": nr, "knots_per_level": [int(v) for v in s["knot_counts"][0] if v >= 0],
           "ms_per_decomposition": round(dt * 1e3, 4), "value": round(n / dt / 1e6, 1), "unit": "Msamples/s",
           "timing": "10 calls back to back + one summary (a refused fused call is repeated there), wall clock",
           "fuse_repeats_in_warm_up": rep_warm, "fuse_repeats_in_timed_calls": eng.fuse_repeats - rep_warm,
           "first_fused_level_of_timed_calls": eng.last_fuse_level,
           # capped fused levels (itd_set_fuse_cap, automatic): the engine learned from the first refusal at which level this input's fused form
           # fails and keeps the levels in front of it fused; 0 = no cap
           "fused_levels_capped_at": eng.last_fuse_cap,
           "own_bytes_per_sample": own_bytes, "own_bytes_frac_of_peak": round(own_bytes * n / dt / 1e9 / HBM_PEAK_GBPS, 4),
           # the reference flow's bytes over the measured time (an equivalent rate: 20 + 24 (L - 1) + 16 B/sample, the last level writes one row)
           "hbm_reference_flow_equivalent_GBps": round((20.0 + 24.0 * (nr - 2) + 16.0) * n / dt / 1e9, 1),
           "knot_indices_bit_exact_every_level": bool(knots_ok), "rows_bit_exact": rows_ok}
    eng.close()
    return out


def f_rows_leg(torch, dev):
    """SURVEY 8(f) operators, informational: ms, Msamples/s and the fraction of the HBM peak on each one's own byte model."""
    import ctypes
    import pyitd_amd
    from pyitd_amd import _lib, spline, streaming
    L = _lib.load()
    n = 1 << 24
    x = torch.from_numpy(sines_noise(n).astype(np.float64)).to(dev)
    out = {}
    eng = pyitd_amd.Engine(n, 1, dev.index or 0)
    base = torch.empty(n, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)

    def timed(f, reps=10):
        f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    # natural cubic (itd_fourier_decomposition.py:49-122), knots detected (itd.cpp:159-169): asynchronous batch entry
    dt = timed(lambda: L.itd_baseline_extract_cubic_batch_f64(eng._h, x.data_ptr(), n, 1, n, None, 0, 0, base.data_ptr(), n, info.data_ptr(), None))
    m = int(info[0].item())
    # byte model: signal read by the detection (8) + baseline written (8) per sample; per knot: ordered list written and read by
    # both sweeps and the evaluation (4 x 4), K / forward / final arrays written once and read once (3 x 16)
    by = 16.0 * n + 64.0 * m
    out["cubic_2p24"] = {"operator": "itd_baseline_extract_fast, knots by itd.cpp:161-168 (%d knots)" % m, "ms": round(dt * 1e3, 4),
                         "Msamples_per_s": round(n / dt / 1e6, 1), "bytes_model": "16 B/sample + 64 B/knot", "frac_of_peak": round(by / dt / 1e9 / HBM_PEAK_GBPS, 4)}
    # instantaneous amplitude / phase / frequency (README.md:13-21, 41-55): reads the rotation twice (8 + 8), writes three rows (24)
    rot = x - base
    amp, ph, fr = (torch.empty(n, dtype=torch.float64, device=dev) for _ in range(3))
    dt = timed(lambda: L.itd_instantaneous_f64(eng._h, rot.data_ptr(), n, amp.data_ptr(), ph.data_ptr(), fr.data_ptr(), None))
    out["instantaneous_2p24"] = {"ms": round(dt * 1e3, 4), "Msamples_per_s": round(n / dt / 1e6, 1), "bytes_model": "8 + 24 B/sample (+ 8 for the crossing scan)",
                                 "frac_of_peak": round(40.0 * n / dt / 1e9 / HBM_PEAK_GBPS, 4)}
    eng.close()
    del x, base, rot, amp, ph, fr
    # totalextract2d on a 512 x 512 image: the only timing the reference records (siftED2D.ipynb cell 3: 10.1457 s)
    img = np.random.default_rng(5).integers(0, 256, (512, 512)).astype(np.float64)
    np.random.seed(1)
    spline.totalextract2d(img, verbose=False)
    t0 = time.perf_counter()
    spline.totalextract2d(img, verbose=False)
    dt = time.perf_counter() - t0
    out["totalextract2d_512x512"] = {"ms": round(dt * 1e3, 3), "reference_recorded_s": 10.1457, "reference_source": "siftED2D.ipynb cell 3 (author's machine, numba)",
                                     "note": "host arrays in and out, host noise generation included"}
    # MEITD (MEITD.py:395-534) on the golden two-tone + noise signal (3000 samples): the arrays stay on the GPU, scalars come back
    try:
        from pyitd_amd import meitd
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "spline", "meitd_two_tone_noise.npz"))
        meitd.MEITD(g["x"].copy())
        dt = timed(lambda: meitd.MEITD(g["x"].copy()), 5)
        hi, lo, _ = meitd.MEITD(g["x"].copy())
        out["meitd_two_tone_noise_3000"] = {"ms": round(dt * 1e3, 3), "components": [int(len(hi)), int(len(lo))],
                                            "same_selection_as_reference": bool(hi.shape == g["high"].shape and lo.shape == g["low"].shape),
                                            "form": "signal uploaded once, components downloaded once; extraction, extrema count and the entropy's sums on the GPU"}
    except Exception as ex:  # noqa: BLE001
        out["meitd_two_tone_noise_3000"] = {"error": repr(ex)[:200]}
    # block-wise operation (itd.cpp:31-44): microseconds per pushed 4096-sample block, device form
    Lb, nb = 4096, 64
    xs = torch.from_numpy(np.cumsum(np.random.default_rng(1).standard_normal(Lb * nb)) * 0.05).to(dev)[None].contiguous()
    ob, orot = torch.empty_like(xs), torch.empty_like(xs)
    for kind in ("cubic", "linear"):
        st = streaming.Stream(Lb, 1, kind, margin=8)
        sp = torch.cuda.current_stream().cuda_stream

        def run():
            for k in range(nb):
                o = max(k - 1, 0) * Lb
                st.push_dev(xs[:, k * Lb:].data_ptr(), Lb * nb, ob[:, o:].data_ptr(), Lb * nb, orot[:, o:].data_ptr(), Lb * nb, sp)
            st.flush_dev(ob[:, (nb - 1) * Lb:].data_ptr(), Lb * nb, orot[:, (nb - 1) * Lb:].data_ptr(), Lb * nb, sp)
        dt = timed(run, 5)
        out["stream_%s_block4096" % kind] = {"us_per_block": round(dt / nb * 1e6, 2), "form": "device buffers, asynchronous pushes"}
        st.close()
    return out


def batch_leg(torch, dev, batch=1024, log2n=20, steps=5):
    """BASELINE configs[2] on this GPU: batch x 2^log2n float32 signals (SURVEY 8d recipe), 8 levels, device resident."""
    import pyitd_amd
    n = 1 << log2n
    x = batch_signals_device(torch, dev, 0, batch, n)
    rows = torch.empty((batch, MAX_ITERATION + 2, n), dtype=torch.float64, device=dev)
    eng = pyitd_amd.Engine(n, batch, dev.index or 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()   # the signals were synthesised on torch's stream, the engine runs on `stream`: without this the
    #                            first call read rows not yet written (all-zero signals -> one row; the lines of rounds 1-2
    #                            showed rows_per_signal [1, 9] for that reason; the timed calls were never affected)
    for _ in range(2):
        eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, MAX_ITERATION, rows.data_ptr(), None, stream.cuda_stream)
    def timed():
        # every step reads its summary: the few signals whose fused sparse levels refuse are re-run there, on their own, inside the
        # timed region
        nonlocal steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, MAX_ITERATION, rows.data_ptr(), None, stream.cuda_stream)
            eng.summary(batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps
    eng.summary(batch)
    rep0, fix0 = eng.fuse_repeats, eng.fuse_signal_repairs
    dt = timed()
    s = eng.summary(batch)     # of the last timed call
    repaired = (eng.fuse_signal_repairs - fix0) / steps
    refused = eng.fuse_repeats > rep0
    if refused:                # a signal's fused levels failed their verification: that call was repeated behind the timed region
        from pyitd_amd.engine import FUSE_OFF
        eng.set_fuse_mode(FUSE_OFF)
        dt = timed()
        s = eng.summary(batch)
    lvl = 0 if refused else eng.last_fuse_level
    # the same batch PIPELINED (itd_set_batch_pipeline(1): chunk k's sample pass on the engine's second stream beside chunk k + 1's knot side,
    # behind a gate) — the form round 6 built to hide the knot side and measured slower than the rotating chunks (profiles/r06/experiments):
    # in the line so that every box shows the comparison
    pipe_ms = None
    if not refused:
        try:
            eng.set_batch_pipeline(1)
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, MAX_ITERATION, rows.data_ptr(), None, stream.cuda_stream)
            eng.summary(batch)
            steps_keep, steps = steps, 3
            pipe_ms = round(timed() * 1e3, 3)
            steps = steps_keep
            eng.set_batch_pipeline(0)
        except Exception:  # noqa: BLE001
            pipe_ms = None
    alg = algorithmic_bytes_per_sample(LEVELS) * batch * n / dt / 1e9
    own = own_bytes_per_sample(MAX_ITERATION + 2, lvl)
    out = {"workload": "batch of %d x 2^%d float32 signals (draw b mod 16, f*(1+b/8192)), %d levels, device resident" % (batch, log2n, LEVELS),
            "value": round(batch * n / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "hbm_reference_flow_equivalent_GBps": round(alg, 1),
            "bytes_note": "equivalent = the reference flow's 20 + 24 x 7 = 188 B/sample (SURVEY 8d) over the measured time: what a "
                          "level-by-level engine would have to sustain, not a measured traffic (it may exceed what the memory system "
                          "delivers); the engine itself moves own_bytes_per_sample (first_fused_level: the levels from there on take one "
                          "pass over the samples): frac_of_peak_own_bytes",
            "first_fused_level": lvl, "own_bytes_per_sample": own,
            "frac_of_peak_own_bytes": round(own * batch * n / dt / 1e9 / HBM_PEAK_GBPS, 4),
            "rows_per_signal": sorted(set(int(v) for v in s["n_rows"])),
            "fused_levels_refused_and_retimed_level_by_level": refused,
            "signals_rerun_on_their_own_per_step": repaired, "summary_read_every_step": True, "order": "timed after the headline",
            "pipelined_form_ms_per_step": pipe_ms}
    eng.close()
    return out


def short_signal_leg(torch, dev, steps=10):
    """Batches of SHORT signals (2^24 samples in all) through the resident form (one launch, one workgroup per signal, the signal in
    LDS: pyitd_amd/csrc/itd_resident.hpp) and through the level-by-level engine — informational, timed after the headline."""
    import pyitd_amd
    from pyitd_amd.engine import RESIDENT_AUTO, RESIDENT_OFF
    out = []
    g = torch.Generator(device="cpu").manual_seed(1)
    for batch, n in ((4096, 4096), (60000, 256)):
        t = torch.arange(n, dtype=torch.float64) / 8000.0
        x = (torch.sin(2 * np.pi * 110 * t)[None, :] + 0.5 * torch.sin(2 * np.pi * 440 * t + 1.3)[None, :]
             + 0.05 * torch.randn((batch, n), generator=g, dtype=torch.float64)).to(torch.float32).to(dev)
        rows = torch.empty((batch, MAX_ITERATION + 2, n), dtype=torch.float64, device=dev)
        eng = pyitd_amd.Engine(n, batch, dev.index or 0)
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        item = {"signals": batch, "samples_per_signal": n}
        for key, mode in (("resident_ms", RESIDENT_AUTO), ("level_by_level_ms", RESIDENT_OFF)):
            eng.set_resident_mode(mode)
            for _ in range(3):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, MAX_ITERATION, rows.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, MAX_ITERATION, rows.data_ptr(), None, stream.cuda_stream)
            torch.cuda.synchronize()
            item[key] = round((time.perf_counter() - t0) / steps * 1e3, 4)
            if mode == RESIDENT_AUTO:
                s = eng.summary(batch)
                item["rows_per_signal"] = sorted(set(int(v) for v in s["n_rows"]))
                item["resident_repeats"] = eng.resident_repeats
                # the resident launch reads the signal once and writes every row once (PMC: profiles/r02/session6_resident_rocprof.txt)
                alg = (4.0 * batch * n + 8.0 * n * float(s["n_rows"].sum())) / (item[key] * 1e-3) / 1e9
                item["resident_hbm_algorithmic_GBps"] = round(alg, 1)
                item["resident_frac_of_peak"] = round(alg / HBM_PEAK_GBPS, 4)
        item["resident_Msamples_per_s"] = round(batch * n / item["resident_ms"] / 1e3, 1)
        out.append(item)
        eng.close()
        del x, rows
    return {"workload": "float32 sines + noise, max_iteration=%d, device resident, %d timed calls per figure" % (MAX_ITERATION, steps),
            "batches": out}


def mid_size_leg(torch, dev, calls=200):
    """Single signals between the one-workgroup form (<= 8192 samples) and the sizes whose launches are memory bound (>= 2^21):
    BASELINE configs[0]'s signal (a 2^16-sample float32 chirp, 4 levels) and the headline recipe at 2^16 / 2^18 / 2^20 samples, 8 levels,
    device resident, one call after the other on one stream.  Every launch here is bound by its ~6.5 us boundary:
    informational, with the fraction of the HBM peak on the result's own bytes (input once, every produced row once)."""
    import pyitd_amd
    out = []
    t = np.arange(1 << 16, dtype=np.float64) / float(1 << 16)
    chirp = np.sin(2 * np.pi * (50.0 * t + 0.5 * (8000.0 - 50.0) * t * t)).astype(np.float32)   # SURVEY 8d, config 1 (8050 / 3884 / 1997 / 800 knots)
    for name, x_host, M in (("configs[0]: chirp, 2^16 samples, 4 levels", chirp, 3), ("sines + noise, 2^16 samples, 8 levels", sines_noise(1 << 16), 7),
                            ("sines + noise, 2^18 samples, 8 levels", sines_noise(1 << 18), 7), ("sines + noise, 2^20 samples, 8 levels", sines_noise(1 << 20), 7)):
        n = len(x_host)
        x = torch.from_numpy(x_host).to(dev)
        rows = torch.empty((M + 2, n), dtype=torch.float64, device=dev)
        eng = pyitd_amd.Engine(n, 1, dev.index or 0)
        torch.cuda.synchronize()
        for _ in range(20):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
        s = eng.summary(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        s = eng.summary(1)
        own = 4.0 * n + 8.0 * n * float(s["n_rows"][0])
        out.append({"signal": name, "us_per_decomposition": round(dt * 1e6, 2), "Msamples_per_s": round(n / dt / 1e6, 1), "rows": int(s["n_rows"][0]),
                    "own_bytes_frac_of_peak": round(own / dt / 1e9 / HBM_PEAK_GBPS, 4)})
        eng.close()
        del x, rows
    return {"workload": "one signal per call, device resident, %d calls per figure" % calls, "signals": out}


def cpu_legs(x_host, n, M, summ, rows, args):
    """CPU baselines timed on this box's host cores (rank 0, N = 1 only), each on a bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cpu_oracle, numba_itd, numpy_itd   # checkers/baselines only: never part of the measured GPU path
    cpu_oracle.lib()
    out = {}
    # (1) the C oracle, one thread (the reference's level recursion is serial), the full signal: whole-signal runs until
    #     ~10 s of CPU work (at most 8 runs); the best run counts
    tc, runs, spent, ref = None, 0, 0.0, None
    while runs < 8 and spent < 10.0:
        tc0 = time.perf_counter()
        ref = cpu_oracle.itd_lean(x_host, M)
        dt = time.perf_counter() - tc0
        tc = dt if tc is None else min(tc, dt)
        runs += 1
        spent += dt
    out["cpu_baseline"] = {
        "value": round(n / tc / 1e6, 3),
        "unit": "Msamples/s",
        "cores": 1,
        "kind": "port",
        "sample": "the same 2^%d-sample signal, %d levels, best of %d whole-signal runs of the C oracle (gcc -O3 -ffp-contract=off, single "
                  "thread: the reference's level recursion is serial), %.2f s per run" % (args.log2n, LEVELS, runs, tc),
        "host_cpus": os.cpu_count(),
    }
    out["parity"] = {
        "knot_counts_equal": [int(v) for v in summ["knot_counts"][0][: len(ref["knot_counts"])]] == ref["knot_counts"].tolist(),
        "rows_bit_exact": bool(np.array_equal(rows[0, : ref["rows"].shape[0]].cpu().numpy().view(np.uint64),
                                              ref["rows"].view(np.uint64))),
    }
    del ref
    # (2) all host cores over independent signals (the batch form: one 2^20 signal per task, ctypes releases the GIL)
    cores = os.cpu_count() or 1
    k_sig = max(64, cores)
    n1 = 1 << 20
    work = [sines_noise(n1, seed=s) for s in range(16)]
    cpu_oracle.itd_lean(work[0], M)
    tc0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as ex:
        list(ex.map(lambda i: cpu_oracle.itd_lean(work[i % 16], M)["rows"].shape[0], range(k_sig)))
    dt = time.perf_counter() - tc0
    out["cpu_baseline_all_cores"] = {
        "value": round(k_sig * n1 / dt / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "%d independent 2^20-sample signals (16 distinct draws), %d levels, C oracle on %d host threads, %.2f s"
                  % (k_sig, LEVELS, cores, dt),
    }
    # (3) the numpy restatement (oracle/numpy_itd.py), one thread, one 2^20 signal
    numpy_itd.itd(work[0][: 1 << 16], M)
    tc0 = time.perf_counter()
    numpy_itd.itd(work[0], M)
    dt = time.perf_counter() - tc0
    out["cpu_baseline_numpy"] = {
        "value": round(n1 / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": "one 2^20-sample signal, %d levels, vectorised numpy restatement (oracle/numpy_itd.py), %.2f s" % (LEVELS, dt),
    }
    # (4) numba: the upstream ITD_numba.py cannot be imported anywhere; the build's own njit restatement
    #     of it (oracle/numba_itd.py) is timed when numba is importable on this box
    if numba_itd.AVAILABLE:
        numba_itd.itd(work[0][: 1 << 14], M)     # compile
        tc0 = time.perf_counter()
        numba_itd.itd(work[0], M)
        dt = time.perf_counter() - tc0
        out["cpu_baseline_numba"] = {
            "value": round(n1 / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "one 2^20-sample signal, %d levels, njit restatement of ITD_numba.py's algorithm (oracle/numba_itd.py), %.2f s" % (LEVELS, dt),
        }
    else:
        out["cpu_baseline_numba"] = {"value": None, "unit": "Msamples/s", "available": False,
                                     "note": "numba is not importable on this box (not in the image, no network); "
                                             "oracle/numba_itd.py holds the njit restatement that would be timed"}
    return out


LIVE_TRAFFIC = None    # measure_traffic_live(): {"k_kf_apply": bytes per launch, "k_extract_f64": ..., "source": ...} or {"error": ...}


def measure_traffic_live(timeout_s=110):
    """roofline.traffic measured IN THIS RUN: two child runs of this file (3 steps of the headline each) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` — separate passes, as MI355X_MICROARCH.md's HBM section prescribes, with
    its gfx950 correction (FETCH_SIZE reports half the bytes of a wide coalesced streaming read) — BEFORE this process touches the GPU (a
    process that has initialised the GPU must not start programs).  What tools/traffic.sh records into profiles/traffic.json, live: HBM-side
    bytes per launch of the dominant kernels.  Any failure (no rocprofv3, a time-out) leaves the recorded figure in the line, labelled."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return {"error": "rocprofv3 not on PATH"}
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_PC_SAMPLING_BETA_ENABLED") is not None:
        return {"error": "this run is itself under a profiler"}
    out = tempfile.mkdtemp(prefix="pyitd_pmc_", dir="/tmp")
    acc = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", os.path.join(out, c), "--",
                   sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-extra", "--warm-ms", "0", "--steps", "3", "--warmup", "1"]
            env = dict(os.environ, TMPDIR="/tmp")
            # (a session of its own: a pass that outlives its time-out is killed with everything it started — no profiled child may
            #  still hold the GPU when this process starts timing)
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
                p.wait()
                return {"error": "the %s pass exceeded %d s" % (c, timeout_s)}
            if rc != 0:
                return {"error": "rocprofv3 --pmc %s: exit code %d" % (c, rc)}
            vals = {}
            for f in glob.glob(os.path.join(out, c, "*", "*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == c:
                        vals.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
            acc[c] = {k: sum(v) / len(v) for k, v in vals.items()}
        res = {"source": "measured in this run, before its timed region: two child runs of bench.py (3 steps of the headline) under rocprofv3 --kernel-trace "
                         "--pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 "
                         "(gfx950: FETCH_SIZE reports half of a wide coalesced streaming read; MI355X_MICROARCH.md, HBM section)"}
        for key, pat in (("k_kf_apply", "k_kf_apply"), ("k_extract_f64", "k_extract<double")):
            ks = [k for k in acc["FETCH_SIZE"] if pat in k and k in acc["WRITE_SIZE"]]
            if ks:
                res[key] = 2.0 * acc["FETCH_SIZE"][ks[0]] * 1024.0 + acc["WRITE_SIZE"][ks[0]] * 1024.0
        return res
    except Exception as ex:  # noqa: BLE001
        return {"error": repr(ex)[:160]}
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=LOG2N, help="N = 1: samples of the single signal (24); N > 1: samples per signal (20)")
    ap.add_argument("--batch", type=int, default=1024, help="N > 1: signals per GPU (weak scaling: the batch grows with N)")
    ap.add_argument("--total-batch", type=int, default=0, help="N > 1: a FIXED batch over all GPUs instead (strong scaling; SURVEY 8d config 4: 8192), "
                                                              "balanced contiguous shards; `scaling` = \"strong\"")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational BASELINE configs[2] leg (1024 x 2^20 signals)")
    ap.add_argument("--try-rccl", action="store_true",
                    help="with --rehearse-one-gpu: ask for RCCL first all the same (two ranks on one GPU: it refuses, which exercises "
                         "the fallback of the control plane to gloo)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 on a one-GPU box: all ranks compute on cuda:0 and gather over gloo (checks the sharded path end to end; "
                         "the line says so and is not a scaling measurement)")
    ap.add_argument("--no-fuse", action="store_true", help="one launch per level throughout (itd_set_fuse_mode OFF): the round-2 engine")
    ap.add_argument("--warm-ms", type=float, default=150.0, help="untimed warm-up of the headline's own step before the W warm-up steps (ms)")
    ap.add_argument("--spawn-timeout", type=int, default=900, help="plain --gpus N: seconds until the parent kills its ranks")
    ap.add_argument("--pg-timeout", type=int, default=120, help="seconds a rank waits for the process group / a collective")
    ap.add_argument("--wav", default=None, help="BASELINE configs[4] on a user-supplied wav (mono / first channel, float32 in [-1, 1], "
                                                "numpy.resize to 2^22, 10 levels): adds config5_audio to the line (also: PYITD_WAV)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # CPU test of the launcher only
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: do not measure roofline.traffic in this run (two PMC passes of "
                                                                      "a 3-step child run, ~40 s, before the timed region); --no-extra implies it")
    args = ap.parse_args()
    if args.gpus > 1 and not args.stub and not args.rehearse_one_gpu:
        rc = preflight(args)
        if rc:
            return rc
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not (args.stub or args.no_extra or args.no_live_traffic or args.no_fuse):
        global LIVE_TRAFFIC
        LIVE_TRAFFIC = measure_traffic_live()
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
