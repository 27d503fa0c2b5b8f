#!/bin/bash
# GPU session 1 of round 2 (run from the repo root on the GPU box)
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02_pytest1.log 2>&1; echo "pytest rc $?" >> $O/r02_pytest1.log
tail -5 $O/r02_pytest1.log
timeout 300 python bench.py > $O/r02_bench1.json 2> $O/r02_bench1.err; tail -c 3000 $O/r02_bench1.json
bash tools/prof.sh r02_stats1 --steps 20 --warmup 3
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
bash tools/pmc.sh r02_sq1 "$SQ1" --steps 3 --warmup 1 > $O/r02_sq1.txt 2>&1
bash tools/pmc.sh r02_sq2 "$SQ2" --steps 3 --warmup 1 > $O/r02_sq2.txt 2>&1
bash tools/pmc.sh r02_tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" --steps 3 --warmup 1 > $O/r02_tcc1.txt 2>&1
cat $O/r02_sq1.txt $O/r02_sq2.txt $O/r02_tcc1.txt
bash tools/ab.sh > $O/r02_ab_early.txt 2>&1; cat $O/r02_ab_early.txt
timeout 600 python tools/batch_bench.py > $O/r02_batch1.txt 2>&1; cat $O/r02_batch1.txt
timeout 300 python tools/nan_bench.py > $O/r02_nan1.txt 2>&1; cat $O/r02_nan1.txt
