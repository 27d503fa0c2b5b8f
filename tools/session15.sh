#!/bin/bash
# round 2, sixth session: resident form with rank windows (n <= 8192): parity, then rates of the shipped build (8 wavefronts per
# SIMD, 64 VGPRs + spills) against builds compiled for 6 / 4 (variants/res_minw_*.so) and the level-by-level engine
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6h
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -x -q > $O/pytest_resident.log 2>&1; rc=$?; tail -5 $O/pytest_resident.log
[ $rc -eq 0 ] || exit $rc
{ echo "--- level by level"; SMALL_RESIDENT_SHAPES=1 PYITD_RESIDENT_MODE=1 timeout -k 10 200 python tools/small_batch_bench.py 2>&1 | grep " x ";
  echo "--- resident, shipped (8 wavefronts per SIMD)"; SMALL_RESIDENT_SHAPES=1 timeout -k 10 200 python tools/small_batch_bench.py 2>&1 | grep " x ";
  for f in variants/res_minw_*.so; do echo "--- resident, $f"; PYITD_HIP_LIB=$PWD/$f SMALL_RESIDENT_SHAPES=1 timeout -k 10 200 python tools/small_batch_bench.py 2>&1 | grep " x "; done; } > $O/resident_windows.txt
cat $O/resident_windows.txt
