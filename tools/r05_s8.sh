#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s8; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; tail -3 $O/pytest_fused.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab.sh variants/libgather0.so variants/libcheck1.so > $O/ab.txt 2>&1; cat $O/ab.txt
timeout -k 10 900 python tools/kf_rates.py 12 11 > $O/kf_delivery_rates.txt 2>&1; rc=$?; tail -36 $O/kf_delivery_rates.txt
exit $rc
