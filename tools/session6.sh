#!/bin/bash
# round 2, second session, closing evidence of the shipped build: green pytest log, unprofiled bench line (all CPU legs), kernel stats,
# traffic counters, the chain's line
O=gpurun_out
python -m pytest tests -q -m gpu > $O/r02s2_pytest_gpu.log 2>&1; tail -2 $O/r02s2_pytest_gpu.log
timeout 400 python bench.py > $O/r02s2_bench_unprofiled.json 2> $O/r02s2_bench.err
bash tools/prof.sh r02s2_stats --steps 20 --warmup 3
bash tools/traffic.sh r02
timeout 300 python bench.py --no-cpu-baseline --chain > $O/r02s2_bench_chain.json 2>/dev/null
timeout 300 python tools/batch_bench.py --chunks 0 --steps 5 > $O/r02s2_batch.txt 2>&1; tail -1 $O/r02s2_batch.txt
