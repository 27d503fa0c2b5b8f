#!/bin/bash
# one GPU session on the fused levels: the fused tests, the knot side's phase profile (variants/libprof.so), two short bench runs
# usage (through gpurun): bash tools/kf_session.sh NAME
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/${1:-kf}; mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_gpu_fused.py -q -m gpu -x > $O/fused.log 2>&1; rc=$?; tail -3 $O/fused.log
[ $rc -ne 0 ] && exit $rc
if [ -f variants/libprof.so ]; then PYITD_HIP_LIB=variants/libprof.so timeout -k 10 200 python tools/knots_prof.py > $O/knots_prof.txt 2>&1 || exit 1; cat $O/knots_prof.txt; fi
for i in 1 2; do timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 > $O/bench_$i.json 2>/dev/null || exit 1; done
python - <<PY
import json
for f in ("1","2"):
    d=json.load(open("$O/bench_%s.json"%f)); r=d["roofline"]
    print(f, "ms/step", d["ms_per_step"], "apply", r["avg_launch_us"], "L0", r["level0_launch_us"], "L1-2", r["extract_launch_us"], "knots", r["knot_side_us"], "span", r["decompose_gpu_us"])
PY
