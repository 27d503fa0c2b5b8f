#!/bin/bash
# GPU session 2 of round 2: the fused level-0 launch
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02_pytest2.log 2>&1; echo "pytest rc $?" >> $O/r02_pytest2.log
tail -15 $O/r02_pytest2.log
PYITD_LEVEL0_MODE=1 timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02_pytest2_records.log 2>&1; echo "pytest rc $?" >> $O/r02_pytest2_records.log
tail -4 $O/r02_pytest2_records.log
timeout 300 python bench.py --no-cpu-baseline > $O/r02_bench2.json 2> $O/r02_bench2.err; python3 -c "
import json; d=json.load(open('$O/r02_bench2.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['level0_launch_us'], r['scan0_launch_us'], r['final_launch_us'], r['per_kernel_frac'])"
PYITD_LEVEL0_MODE=1 timeout 300 python bench.py --no-cpu-baseline > $O/r02_bench2_records.json 2>/dev/null; python3 -c "
import json; d=json.load(open('$O/r02_bench2_records.json')); r=d['roofline']; print('records mode', d['value'], d['ms_per_step'], r['avg_launch_us'], r['level0_launch_us'], r['scan0_launch_us'], r['final_launch_us'])"
bash tools/prof.sh r02_stats2 --steps 20 --warmup 3
timeout 200 python tools/fuzz_parity.py 400 7 > $O/r02_fuzz2.txt 2>&1; tail -2 $O/r02_fuzz2.txt
timeout 200 python tools/fuzz_parity.py batch 60 8 >> $O/r02_fuzz2.txt 2>&1; tail -1 $O/r02_fuzz2.txt
