#!/bin/bash
# bench each diagnostic/variant build under variants/ (level-0 / levels>=1 / final extraction times)
for f in variants/*.so; do
  PYITD_HIP_LIB=$PWD/$f python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$f $PYITD_WAVES_PER_CU'.ljust(34), 'level0 %.1f us   levels>=1 %.1f us   final %.1f us   step %.3f ms' % (r['level0_launch_us'], r['avg_launch_us'], r['final_launch_us'], d['ms_per_step']))"
done
