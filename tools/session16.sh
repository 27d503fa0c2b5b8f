#!/bin/bash
# resident form: builds compiled for 5 / 6 / 7 wavefronts per SIMD against the shipped 8
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6h
mkdir -p $O
{ echo "--- resident, shipped (8 wavefronts per SIMD)"; SMALL_RESIDENT_SHAPES=1 timeout -k 10 200 python tools/small_batch_bench.py 2>&1 | grep " x ";
  for f in variants/res_minw_*.so; do echo "--- resident, $f"; PYITD_HIP_LIB=$PWD/$f SMALL_RESIDENT_SHAPES=1 timeout -k 10 200 python tools/small_batch_bench.py 2>&1 | grep " x "; done; } > $O/resident_minw.txt
cat $O/resident_minw.txt
