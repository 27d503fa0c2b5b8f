#!/bin/bash
# Same-call A/B: bench.py's headline against the shipped library and every variant build given (default: variants/*.so), twice
# each (boxes of the pool differ by up to 20 %, runs on one box by 1-2 %: only same-call comparisons count).
#   usage (GPU box): bash tools/ab.sh [variant.so ...]      build a variant: hipcc ... -DITD_...=... -o variants/NAME.so
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(variants/*.so)
for round in 1 2; do
  for f in "" "${libs[@]}"; do
    PYITD_HIP_LIB=${f:+$PWD/$f} python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('${f:-shipped}'.ljust(30), 'step %.4f ms   dominant %.1f us   level0 %.1f   levels 1-2 %.1f   knot side %s' % (d['ms_per_step'], r['avg_launch_us'], r['level0_launch_us'], r.get('extract_launch_us', 0), r.get('knot_side_us')))"
  done
done
