#!/bin/bash
# The -m gpu suite with the fused sparse levels in their other modes (every Engine reads these variables at creation: pyitd_amd/engine.py):
# off; fusing every signal of >= 65536 samples; that with 16-tile knot-side workgroups; that with the first fused level pinned to 2; that with
# the fused levels capped at level 5 (the rest level by level behind them); that with batches pipelined (itd_set_batch_pipeline);
# and with every workspace the library allocates filled with 0xFF bytes first (PYITD_POISON=1: a kernel that reads what nobody wrote fails).
# usage (through gpurun): bash tools/suite_modes.sh r05
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
tag=${1:-r05}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
rc_all=0
run() { name=$1; shift; env "$@" timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_$name.log 2>&1; rc=$?; echo "$name: $(tail -1 $O/pytest_gpu_$name.log)"; [ $rc -ne 0 ] && rc_all=$rc; }
run fuse_off PYITD_FUSE_MODE=1
run fuse_min_65536 PYITD_FUSE_MIN=65536
run fuse_min_65536_range16 PYITD_FUSE_MIN=65536 PYITD_FUSE_RANGE=16
run fuse_min_65536_level2 PYITD_FUSE_MIN=65536 PYITD_FUSE_LEVEL=2
run fuse_min_65536_cap5 PYITD_FUSE_MIN=65536 PYITD_FUSE_CAP=5
run fuse_min_65536_pipeline PYITD_FUSE_MIN=65536 PYITD_BATCH_PIPELINE=1
run poisoned_workspaces PYITD_POISON=1
run poisoned_workspaces_fuse_min_65536 PYITD_POISON=1 PYITD_FUSE_MIN=65536
exit $rc_all
