#!/bin/bash
# evidence for profiles/r02_*: kernel stats of the shipped build, traffic counters, unprofiled bench line, batch + big-N runs
O=gpurun_out
timeout 300 python bench.py > $O/r02_bench_final.json 2> $O/r02_bench_final.err
bash tools/prof.sh r02_stats_final --steps 20 --warmup 3
bash tools/traffic.sh r02
timeout 300 python bench.py --no-cpu-baseline --log2n 27 --steps 5 --warmup 2 > $O/r02_bench_2p27.json 2>/dev/null
timeout 600 python tools/batch_bench.py --chunks 0,1024 --steps 5 > $O/r02_batch_final.txt 2>&1
timeout 300 python tools/cubic_bench.py > $O/r02_cubic_bench.txt 2>&1
tail -3 $O/r02_batch_final.txt; cat $O/r02_cubic_bench.txt
