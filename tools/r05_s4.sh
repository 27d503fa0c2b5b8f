#!/bin/bash
# round 5, session 4: automatic first fused level (2 where it pays), full suite, fuzz on the fused path, the bench line, the batch
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s4; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -4 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 timeout -k 10 600 python tools/fuzz_parity.py 1500 501 > $O/fuzz_1500_long_fused.txt 2>&1; rc=$?; tail -3 $O/fuzz_1500_long_fused.txt; [ $rc -ne 0 ] && exit $rc
FUZZ_MIN_N=65536 PYITD_FUSE_MIN=65536 PYITD_FUSE_LEVEL=2 timeout -k 10 600 python tools/fuzz_parity.py 1500 502 > $O/fuzz_1500_long_fused_level2.txt 2>&1; rc=$?; tail -3 $O/fuzz_1500_long_fused_level2.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/fuzz_parity.py 3000 503 > $O/fuzz_3000_default.txt 2>&1; rc=$?; tail -2 $O/fuzz_3000_default.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > $O/bench_default_form.json 2> $O/bench_err.log; rc=$?; python - <<PY
import json
d=json.load(open("$O/bench_default_form.json")); r=d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "apply", r["avg_launch_us"], "L0", r["level0_launch_us"], "L1", r.get("extract_launch_us"), "knots", r.get("knot_side_us"))
print({k: d[k] for k in d if k in ("config3_batch","config5_audio","headline_on_quantised_2p24")})
PY
exit $rc
