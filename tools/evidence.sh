#!/bin/bash
# One GPU session that collects a round's evidence into gpurun_out/<tag>/ (copied into profiles/<round>/ afterwards):
#   the -m gpu suite's log; bench.py in the driver's form and with 200 steps; rocprofv3 --kernel-trace --stats of the driver's form;
#   the PMC traffic passes (tools/traffic.sh); the knot side's phase profile (variants/libprof.so, built with -DITD_PROF=1, if present);
#   the delivery rates of the fused levels by signal family; fuzz slices (default mode, batches, everything fused from level 3 / 2);
#   the suite again in the fused levels' other modes (tools/suite_modes.sh).
# usage (through gpurun): bash tools/evidence.sh r05
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
tag=${1:-r05}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -2 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 200 --no-extra --no-cpu-baseline > $O/bench_steps200.json 2>/dev/null || exit 1
# (the same with the first fused level pinned to 3: the sample pass of round 4's shape — six rows — with this round's checks)
PYITD_FUSE_LEVEL=3 timeout -k 10 300 python bench.py --steps 100 --no-extra --no-cpu-baseline > $O/bench_first_fused_level3.json 2>/dev/null || exit 1
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err ) || exit 1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && cut -c1-160 $O/kernel_stats.csv | head -8
bash tools/traffic.sh $tag > $O/traffic_summary.txt 2>&1; tail -1 $O/traffic_summary.txt | cut -c1-400
# (the line in the driver's form reads the traffic record of THIS build: traffic_stale false)
cp gpurun_out/traffic.json profiles/traffic.json
timeout -k 10 900 python bench.py > $O/bench_default_form.json 2> $O/bench_default_form.err || exit 1
if [ -f variants/libprof.so ]; then PYITD_HIP_LIB=variants/libprof.so timeout -k 10 200 python tools/knots_prof.py > $O/knots_phase_profile.txt 2>&1 || exit 1; fi
timeout -k 10 900 python tools/kf_rates.py 12 11 > $O/kf_delivery_rates.txt 2>&1 || exit 1; tail -2 $O/kf_delivery_rates.txt
FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 timeout -k 10 600 python tools/fuzz_parity.py 3000 601 > $O/fuzz_3000_long_fused.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused.txt
FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 PYITD_FUSE_LEVEL=2 timeout -k 10 600 python tools/fuzz_parity.py 3000 602 > $O/fuzz_3000_long_fused_level2.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused_level2.txt
timeout -k 10 600 python tools/fuzz_parity.py 20000 603 > $O/fuzz_20000_default.txt 2>&1 || exit 1; tail -1 $O/fuzz_20000_default.txt
timeout -k 10 600 python tools/fuzz_parity.py batch 1500 604 > $O/fuzz_1500_batches.txt 2>&1 || exit 1; tail -1 $O/fuzz_1500_batches.txt
# MEITD: the loop as one launch against one launch per operator (wall time, operators one by one, 1000 random signals), the spline solver
timeout -k 10 300 python tools/meitd_bench.py > $O/meitd_wall_time.txt 2>&1 || exit 1; cut -c1-100 $O/meitd_wall_time.txt
timeout -k 10 300 python tools/meitd_ops.py > $O/meitd_ops.txt 2>&1 || exit 1
timeout -k 10 600 python tools/meitd_fuzz.py 5000 7 > $O/meitd_fuzz_5000.txt 2>&1 || exit 1; tail -1 $O/meitd_fuzz_5000.txt
timeout -k 10 300 python tools/spline_long_bench.py > $O/spline_long_signal.txt 2>&1 || exit 1; tail -1 $O/spline_long_signal.txt
# the block-wise operators against the recipe's CPU statement, plain and with poisoned workspaces; the parity fuzz with poisoned workspaces
timeout -k 10 600 python tools/stream_fuzz.py 5000 9 > $O/stream_fuzz_5000.txt 2>&1 || exit 1; tail -1 $O/stream_fuzz_5000.txt
PYITD_POISON=1 timeout -k 10 600 python tools/stream_fuzz.py 2000 10 > $O/stream_fuzz_2000_poisoned.txt 2>&1 || exit 1; tail -1 $O/stream_fuzz_2000_poisoned.txt
PYITD_POISON=1 timeout -k 10 600 python tools/fuzz_parity.py 5000 605 > $O/fuzz_5000_default_poisoned.txt 2>&1 || exit 1; tail -1 $O/fuzz_5000_default_poisoned.txt
timeout -k 10 600 python tools/ops_fuzz.py 30000 2 2>/dev/null > $O/ops_fuzz_30000.txt || exit 1; tail -1 $O/ops_fuzz_30000.txt
PYITD_POISON=1 timeout -k 10 600 python tools/ops_fuzz.py 5000 3 2>/dev/null > $O/ops_fuzz_5000_poisoned.txt || exit 1; tail -1 $O/ops_fuzz_5000_poisoned.txt
timeout -k 10 600 python tools/tfe_fuzz.py 3000 1 > $O/tfe_fuzz_3000.txt 2>&1 || exit 1; tail -1 $O/tfe_fuzz_3000.txt
timeout -k 10 600 python tools/spline_fuzz.py 3000 2 > $O/spline_fuzz_3000.txt 2>&1 || exit 1; tail -1 $O/spline_fuzz_3000.txt
timeout -k 10 600 python tools/batch_ops_fuzz.py 5000 2 > $O/batch_ops_fuzz_5000.txt 2>&1 || exit 1; tail -1 $O/batch_ops_fuzz_5000.txt
timeout -k 10 600 python tools/repair_fuzz.py 500 2 2>/dev/null > $O/repair_fuzz_500.txt || exit 1; tail -1 $O/repair_fuzz_500.txt
bash tools/suite_modes.sh $tag || exit 1
python - <<PY
import json
for f in ("bench_default_form", "bench_steps200", "bench_first_fused_level3", "bench_under_rocprof"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, d["ms_per_step"], r["frac"], r["avg_launch_us"], r["level0_launch_us"], r["extract_launch_us"], r["knot_side_us"])
PY
