#!/bin/bash
# level-0 A/B helper (GPU box): the shipped library against variant libraries (PYITD_HIP_LIB), alternating, many steps each
# usage: bash tools/l0_ab.sh [variant.so ...]
cd $GRAFT_REPO_ROOT
one() {
python bench.py --no-cpu-baseline --no-extra --steps ${STEPS:-80} --warmup 5 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-24s' % sys.argv[1], 'ms/step', d['ms_per_step'], 'levels>=1', r['avg_launch_us'], 'level0', r['level0_launch_us'], 'final', r['final_launch_us'])" "$1"
}
for rep in 1 2; do
  one shipped
  for v in "$@"; do PYITD_HIP_LIB=$v one $v; done
done
