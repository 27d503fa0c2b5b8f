#!/bin/bash
# round 5, session 3: the hand-over layout (first fused level 2 with 64-tile ranges)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s3; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; tail -5 $O/pytest_fused.log; [ $rc -ne 0 ] && exit $rc
PYITD_FUSE_LEVEL=2 timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "deliver or headline or batch or graph or host_api or final or small_ranges" > $O/pytest_fused_L2.log 2>&1; rc=$?; tail -5 $O/pytest_fused_L2.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/fuse_level_probe.py > $O/fuse_level_probe.txt 2>&1; cat $O/fuse_level_probe.txt
exit 0
