#!/bin/bash
# round 5, session 1: the fused levels' tests with the complete verification + fault injection, the cost of the checks (A/B), the suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s1; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; tail -15 $O/pytest_fused.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab.sh variants/libnoverify.so variants/libwaves0.so > $O/ab_verify.txt 2>&1; cat $O/ab_verify.txt
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -5 $O/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
exit 0
