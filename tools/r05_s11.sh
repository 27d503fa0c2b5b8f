#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s11; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_spline.py -x -q -m gpu > $O/pytest_spline.log 2>&1; rc=$?; tail -5 $O/pytest_spline.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/meitd_bench.py > $O/meitd_wall_time.txt 2>&1; cat $O/meitd_wall_time.txt
timeout -k 10 300 python tools/meitd_ops.py > $O/meitd_ops.txt 2>&1; cat $O/meitd_ops.txt
exit 0
