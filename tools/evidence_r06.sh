#!/bin/bash
# Round 6's evidence in three GPU sessions (tools/evidence.sh's steps, regrouped so that each call stays well inside gpurun's limit):
#   bash tools/evidence_r06.sh core   the -m gpu log; bench.py in the driver's form (live PMC traffic), with 200 steps, under rocprofv3
#                                     --kernel-trace --stats; the recorded traffic passes (profiles/traffic.json); a batch timeline
#   bash tools/evidence_r06.sh modes  the suite in the fused levels' other modes
#   bash tools/evidence_r06.sh fuzz   delivery rates by signal family; fuzz slices (fused long signals from level 3 / 2, capped, default, batches, pipelined batches)
# Output: gpurun_out/r06/ (copied into profiles/r06/ afterwards).
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r06; mkdir -p $O
export TMPDIR=/tmp
case ${1:-core} in
core)
  timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -2 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
  timeout -k 10 300 python bench.py --steps 200 --no-extra --no-cpu-baseline > $O/bench_steps200.json 2>/dev/null || exit 1
  ( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof.err ) || exit 1
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats.csv && cut -c1-160 $O/kernel_stats.csv | head -8
  rm -rf $O/prof
  bash tools/traffic.sh r06 > $O/traffic_summary.txt 2>&1; tail -1 $O/traffic_summary.txt | cut -c1-400
  cp gpurun_out/traffic.json profiles/traffic.json; cp gpurun_out/traffic.json $O/traffic.json; rm -rf gpurun_out/traffic_r06
  timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_form.json 2> $O/bench_default_form.err || exit 1
  bash tools/pipeline_trace.sh r06 "64 0 3 2 0" "64 1 3 2 0" || exit 1
  timeout -k 10 300 python tools/pipeline_sweep.py 1024 20 0,8 > $O/pipeline_sweep_1024.txt 2>&1 || exit 1; tail -4 $O/pipeline_sweep_1024.txt
  python - <<PY
import json
for f in ("bench_default_form", "bench_steps200", "bench_under_rocprof"):
    d = json.load(open("$O/%s.json" % f)); r = d["roofline"]
    print(f, d["ms_per_step"], r["frac"], r["avg_launch_us"], r["level0_launch_us"], r["extract_launch_us"], r["knot_side_us"], r.get("traffic_live_attempt"))
PY
  ;;
modes)
  bash tools/suite_modes.sh r06 || exit 1
  ;;
fuzz)
  timeout -k 10 900 python tools/kf_rates.py 12 11 > $O/kf_delivery_rates.txt 2>&1 || exit 1; tail -2 $O/kf_delivery_rates.txt
  FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 timeout -k 10 600 python tools/fuzz_parity.py 3000 601 > $O/fuzz_3000_long_fused.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused.txt
  FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 PYITD_FUSE_LEVEL=2 timeout -k 10 600 python tools/fuzz_parity.py 3000 602 > $O/fuzz_3000_long_fused_level2.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused_level2.txt
  FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 PYITD_FUSE_CAP=5 timeout -k 10 600 python tools/fuzz_parity.py 3000 606 > $O/fuzz_3000_long_fused_cap5.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused_cap5.txt
  FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 PYITD_FUSE_LEVEL=2 PYITD_FUSE_CAP=6 timeout -k 10 600 python tools/fuzz_parity.py 3000 607 > $O/fuzz_3000_long_fused_level2_cap6.txt 2>&1 || exit 1; tail -1 $O/fuzz_3000_long_fused_level2_cap6.txt
  timeout -k 10 600 python tools/fuzz_parity.py 20000 603 > $O/fuzz_20000_default.txt 2>&1 || exit 1; tail -1 $O/fuzz_20000_default.txt
  timeout -k 10 600 python tools/fuzz_parity.py batch 1500 604 > $O/fuzz_1500_batches.txt 2>&1 || exit 1; tail -1 $O/fuzz_1500_batches.txt
  PYITD_FUSE_MIN=65536 PYITD_BATCH_PIPELINE=1 timeout -k 10 600 python tools/fuzz_parity.py batch 1500 608 > $O/fuzz_1500_batches_pipelined.txt 2>&1 || exit 1; tail -1 $O/fuzz_1500_batches_pipelined.txt
  ;;
esac
