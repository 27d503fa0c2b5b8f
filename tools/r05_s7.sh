#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s7; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_a.log 2>&1; rc=$?; tail -4 $O/pytest_a.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab.sh variants/libtwoloops.so > $O/ab_level0.txt 2>&1; cat $O/ab_level0.txt
exit 0
