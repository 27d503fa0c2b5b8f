#!/bin/bash
# round 2, sixth session, first call: the resident form of short signals (itd_resident.hpp) — parity tests, then the short-signal
# batch rates with and without it.  Everything lands under gpurun_out/s6/
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.log 2>&1; rc=$?; tail -15 $O/pytest_resident.log
[ $rc -eq 0 ] || exit $rc
SMALL_SHAPES_MAX=4096 PYITD_RESIDENT_MODE=1 timeout -k 10 200 python tools/small_batch_bench.py > $O/small_batches_level_by_level.txt 2>&1 && \
timeout -k 10 200 python tools/small_batch_bench.py > $O/small_batches_resident.txt 2>&1
echo "--- level by level"; cat $O/small_batches_level_by_level.txt; echo "--- resident (n <= 4096)"; cat $O/small_batches_resident.txt
