#!/bin/bash
# round 2, sixth session: resident form, size classes — parity, then rates per class against the level-by-level engine
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.log 2>&1; rc=$?; tail -5 $O/pytest_resident.log
[ $rc -eq 0 ] || exit $rc
SMALL_RESIDENT_SHAPES=1 PYITD_RESIDENT_MODE=1 timeout -k 10 200 python tools/small_batch_bench.py > $O/resident_classes_level_by_level.txt 2>&1 && \
SMALL_RESIDENT_SHAPES=1 timeout -k 10 200 python tools/small_batch_bench.py > $O/resident_classes_resident.txt 2>&1
echo "--- level by level"; cat $O/resident_classes_level_by_level.txt; echo "--- resident (n <= 4096)"; cat $O/resident_classes_resident.txt
