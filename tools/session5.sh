#!/bin/bash
# round 2, session 2: the one-launch chain — bench line, kernel stats, instruction counters (SQ) next to the level-by-level engine's
O=gpurun_out
timeout 300 python bench.py --no-cpu-baseline --chain > $O/r02_bench_chain.json 2> $O/r02_bench_chain.err
bash tools/prof.sh r02_stats_chain --steps 20 --warmup 3 --chain
bash tools/pmc.sh r02_pmc_chain_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --steps 5 --warmup 1 --chain
bash tools/pmc.sh r02_pmc_chain_b "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" --steps 5 --warmup 1 --chain
bash tools/pmc.sh r02_pmc_classic_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" --steps 5 --warmup 1
bash tools/pmc.sh r02_pmc_classic_b "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" --steps 5 --warmup 1
