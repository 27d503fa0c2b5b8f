#!/bin/bash
# round 5, session 6: the bench line with its new legs, the knot side's phase profile, the PMC traffic passes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python bench.py > $O/bench_default_form.json 2> $O/bench_err.log; rc=$?; tail -3 $O/bench_err.log; [ $rc -ne 0 ] && exit $rc
python - <<PY
import json
d=json.load(open("$O/bench_default_form.json")); r=d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "read_frac", r.get("read_frac"), "apply", r["avg_launch_us"], "L0", r["level0_launch_us"], "L1", r.get("extract_launch_us"), "knots", r.get("knot_side_us"), "stale", r.get("traffic_stale"))
print("valid flags:", d.get("headline_with_valid_flags"))
m = d.get("many_mid_size_signals")
if m and "shapes" in m:
    for it in m["shapes"]:
        print(it["signals"], it["samples_per_signal"], it["one_batch_call_default"], it["one_batch_call_best_of_sweep"], it["one_signal_per_call_engine_pool"])
else: print(m)
print("batch:", {k: d["config3_batch"].get(k) for k in ("ms_per_step","first_fused_level","frac_of_peak_own_bytes","signals_rerun_on_their_own_per_step")})
print("audio:", {k: d["config5_audio"].get(k) for k in ("ms_per_decomposition","fuse_repeats","rows_bit_exact")})
print("f_rows meitd:", d["f_rows"].get("meitd_two_tone_noise_3000"))
PY
PYITD_HIP_LIB=variants/libprof.so timeout -k 10 200 python tools/knots_prof.py > $O/knots_phase_profile.txt 2>&1; rc=$?; tail -45 $O/knots_phase_profile.txt; [ $rc -ne 0 ] && exit $rc
bash tools/traffic.sh r05 > $O/traffic_summary.txt 2>&1; tail -1 $O/traffic_summary.txt | cut -c1-600
exit 0
