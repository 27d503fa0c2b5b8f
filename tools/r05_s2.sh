#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s2; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; tail -15 $O/pytest_fused.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab.sh variants/*.so > $O/ab_verify.txt 2>&1; cat $O/ab_verify.txt
exit 0
