#!/bin/bash
# timing-only ablations of the resident kernel (variants/res_abl_<bits>.so, built with -DITD_RES_ABL=<bits>; bit 64 = never stop
# naturally, the base every other build is compared with): which phase costs what
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6
mkdir -p $O
for f in variants/res_abl_*.so; do
  echo "== $f"
  PYITD_HIP_LIB=$PWD/$f SMALL_SHAPE=60000x256,16384x1024,4096x4096,1x1024 timeout -k 10 120 python tools/small_batch_bench.py 2>&1 | grep " x " || exit 1
done > $O/resident_ablation.txt
cat $O/resident_ablation.txt
