#!/bin/bash
# round 2, fifth session: re-validation of the rebuilt checkout (GPU box).  Everything lands under gpurun_out/s5/
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err && \
timeout 300 python bench.py --gpus 2 --rehearse-one-gpu --batch 128 --steps 5 --warmup 2 > $O/bench_2rank_rehearsal.json 2> $O/bench_2rank_rehearsal.err
cat $O/bench_2rank_rehearsal.json
echo done
