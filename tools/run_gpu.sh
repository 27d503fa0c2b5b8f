#!/bin/bash
# One parametrised GPU session (replaces the per-session scripts of rounds 1-2):  tools/run_gpu.sh NAME step [step ...]
# Output goes to gpurun_out/NAME/.  Steps stop at the first failure (no GPU step after a failed / killed one).
#   tests[:EXPR]   pytest -m gpu (optionally -k EXPR)         bench[:ARGS]   python bench.py ARGS
#   prof           rocprofv3 --kernel-trace --stats of bench.py          py:SCRIPT[:ARGS]   python tools/SCRIPT ARGS
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
NAME=$1; shift
O=gpurun_out/$NAME
mkdir -p "$O"
export TMPDIR=/tmp
for step in "$@"; do
    kind=${step%%:*}; arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
    case $kind in
        tests) if [ -n "$arg" ]; then timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "$arg" > "$O/pytest_$(echo "$arg" | tr ' /' '__').log" 2>&1; rc=$?; tail -3 "$O/pytest_$(echo "$arg" | tr ' /' '__').log"
               else timeout -k 10 1100 python -m pytest tests -x -q -m gpu > "$O/pytest_gpu.log" 2>&1; rc=$?; tail -3 "$O/pytest_gpu.log"; fi ;;
        bench) timeout -k 10 600 python bench.py $arg > "$O/bench_$(echo "$arg" | tr ' /-' '___').json" 2> "$O/bench_err.log"; rc=$?; tail -c 1500 "$O/bench_$(echo "$arg" | tr ' /-' '___').json" ;;
        prof)  ( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o bench -- python3 "$OLDPWD/bench.py" --no-extra --no-cpu-baseline --warm-ms 20 > "$OLDPWD/$O/bench_under_rocprof.json" 2> "$OLDPWD/$O/prof_err.log" ); rc=$?
               f=$(find "$O/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats.csv" && head -12 "$O/kernel_stats.csv" ;;
        py)    script=${arg%%:*}; sargs=${arg#*:}; [ "$sargs" = "$arg" ] && sargs=""
               timeout -k 10 900 python tools/$script $sargs > "$O/${script%.py}.txt" 2>&1; rc=$?; tail -40 "$O/${script%.py}.txt" ;;
        *) echo "unknown step $step"; rc=2 ;;
    esac
    echo "== $step -> rc $rc"
    [ $rc -ne 0 ] && exit $rc
done
exit 0
