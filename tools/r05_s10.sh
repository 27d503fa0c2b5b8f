#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
O=gpurun_out/r05_s10; mkdir -p $O; export TMPDIR=/tmp
one() { PYITD_HIP_LIB=${1:+$PWD/$1} BENCH_ROW_PAD=$2 python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('${1:-shipped}'.ljust(30), 'step %.4f ms   apply %.1f us   level0 %.1f   level 1 %.1f   knot side %s' % (d['ms_per_step'], r['avg_launch_us'], r['level0_launch_us'], r.get('extract_launch_us', 0), r.get('knot_side_us')))"; }
for rep in 1 2; do one "" 0; one variants/libpad512.so 512; one variants/libpad1536.so 1536; done > $O/ab_row_pad.txt 2>&1
cat $O/ab_row_pad.txt
