#!/bin/bash
# SQ counters of the path's kernels (separate PMC passes, kernel-trace only) and the -m gpu suite in the fused levels' other modes.
# usage (through gpurun): bash tools/r04_counters.sh
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r04c; mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '+')
  bash tools/pmc.sh r04c/$tag "$set" --steps 3 --warmup 1 > $O/sq_$tag.txt 2>&1 || exit 1
done
cat $O/sq_*.txt > $O/sq_counters_all.txt
PYITD_FUSE_MODE=1 timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_fuse_off.log 2>&1; echo "fuse off rc=$?"; tail -1 $O/pytest_gpu_fuse_off.log
PYITD_FUSE_MIN=65536 timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_fuse_min_65536.log 2>&1; echo "fuse min rc=$?"; tail -1 $O/pytest_gpu_fuse_min_65536.log
PYITD_FUSE_MIN=65536 PYITD_FUSE_RANGE=16 timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_fuse_min_65536_range16.log 2>&1; echo "fuse min range16 rc=$?"; tail -1 $O/pytest_gpu_fuse_min_65536_range16.log
