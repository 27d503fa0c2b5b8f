#!/bin/bash
# per-phase instruction counts of the fused level-0 launch by ablation builds (-DITD_ABL_R=...: timing-only skeletons, results wrong)
# usage: bash tools/l0_abl.sh variants/libabl65536.so variants/libabl2048.so ...   (65536: no scan_publish; 2048: no by-rank phases)
cd $GRAFT_REPO_ROOT
for v in "" "$@"; do
  if [ -n "$v" ]; then export PYITD_HIP_LIB=$GRAFT_REPO_ROOT/$v; fi
  echo "== ${v:-shipped}"
  rm -rf gpurun_out/l0_abl
  bash tools/pmc.sh l0_abl "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS" --steps 2 --warmup 1 2>/dev/null | grep -A7 "k_extract<float" | head -8
done
