"""`python bench.py --gpus N`, run plainly, must start N ranks itself (the parent never touches the GPU), shard the batch
over them and gather the per-signal table.  Here with --stub: gloo ranks on CPU, the engine replaced by a stand-in that
only reports which signals the rank owned — this test is about the launcher and the sharding, not about compute."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=300, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_gathers_the_table():
    d = _run(["--gpus", "2", "--batch", "3", "--log2n", "10"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["signals_per_gpu"] == 3 and d["config"]["samples_per_signal"] == 1024
    assert d["config"]["signals_in_gathered_table"] == 6
    assert d["config"]["table_signal_ids"] == [0, 1, 2, 3, 4, 5]          # batch order: rank 0's shard, then rank 1's
    assert len(d["config"]["per_rank_ms_per_step"]) == 2
    assert "1024 signals" not in d["config"]["workload"] and "2 x 3 signals" in d["config"]["workload"]


def test_under_an_external_launcher_the_process_is_one_rank():
    # what the driver does for N > 1: WORLD_SIZE etc. are already set -> no spawning; a 1-rank world runs alone
    d = _run(["--gpus", "1", "--log2n", "10"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert d["n_gpus"] == 1 and d["config"]["signals_in_gathered_table"] is None


def test_more_ranks_than_devices_fails_fast_with_a_message():
    """`bench.py --gpus 8` where fewer devices are visible starts nothing: exit code 2, the reason on stderr and in one JSON line
    (no rank ever allocates or waits for a rendezvous)."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["HIP_VISIBLE_DEVICES"] = ""          # whatever this host has: none visible
    e["ROCR_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2"], capture_output=True, text=True, timeout=120, env=e)
    assert p.returncode == 2
    assert "needs 8 visible devices" in p.stderr and "--rehearse-one-gpu" in p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["error"].startswith("needs 8 devices")


def test_total_batch_is_strong_scaling_with_balanced_shards():
    """`--total-batch T`: a FIXED batch over all ranks (SURVEY 8d config 4's strong-scaling form): balanced contiguous shards (the
    first T % N ranks own one signal more), `scaling` = "strong", the gathered table in batch order, the all-gather timed on its own
    and every rank's repair counts in the line."""
    d = _run(["--gpus", "2", "--total-batch", "7", "--log2n", "10"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["signals_in_gathered_table"] == 7 and d["config"]["table_signal_ids"] == list(range(7))
    assert d["config"]["signals_per_gpu"] == 4 and "7 (fixed: strong scaling)" in d["config"]["workload"]
    assert d["config"]["summary_allgather_ms"] is not None and len(d["config"]["per_rank_signals_rerun_per_step__whole_call_repeats__allgather_ms"]) == 2
    assert d["value"] > 0


def test_eight_ranks_with_an_uneven_total_batch():
    """The shape of the driver's SCALE run that no round has had hardware for: eight ranks (gloo on CPU, the engine stubbed), a fixed
    batch that does not divide by eight (8195 = 8 x 1024 + 3): the first three ranks own 1025 signals, the rest 1024, every signal
    appears once in the gathered table, in batch order, and the line carries one entry per rank."""
    d = _run(["--gpus", "8", "--total-batch", "8195", "--log2n", "10"])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["signals_in_gathered_table"] == 8195
    assert d["config"]["table_signal_ids"] == list(range(8195))
    assert d["config"]["signals_per_gpu"] == 1025
    assert len(d["config"]["per_rank_ms_per_step"]) == 8
    assert len(d["config"]["per_rank_signals_rerun_per_step__whole_call_repeats__allgather_ms"]) == 8
    assert d["value"] > 0


def test_eight_ranks_weak_scaling_form():
    d = _run(["--gpus", "8", "--batch", "5", "--log2n", "10"])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["signals_in_gathered_table"] == 40 and d["config"]["table_signal_ids"] == list(range(40))
