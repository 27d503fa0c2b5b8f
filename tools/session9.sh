#!/bin/bash
# round 2, fourth session: evidence of the final build (GPU box).  Everything lands under gpurun_out/s4f/
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4f
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
PYITD_LEVEL0_MODE=1 python -m pytest tests -q -m gpu > $O/pytest_gpu_record_driven_level0.log 2>&1; tail -1 $O/pytest_gpu_record_driven_level0.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err
bash tools/prof.sh s4f/prof --steps 100 --warmup 30 > $O/kernel_stats.txt
bash tools/traffic.sh r02 > $O/traffic_summary.txt && cp gpurun_out/traffic.json $O/traffic.json
timeout 300 python bench.py --gpus 2 --rehearse-one-gpu --batch 128 --steps 5 --warmup 2 > $O/bench_2rank_rehearsal.json 2> /dev/null
timeout 300 python tools/batch_bench.py --chunks 0 --steps 5 > $O/batch_1024x2p20.txt 2>&1; tail -1 $O/batch_1024x2p20.txt
timeout 200 python tools/small_batch_bench.py > $O/small_batches.txt 2>&1
{ echo "-- single"; timeout 300 python tools/fuzz_parity.py 8000 3030; echo "-- batch"; timeout 300 python tools/fuzz_parity.py batch 600 3031;
  echo "-- records"; PYITD_LEVEL0_MODE=1 timeout 300 python tools/fuzz_parity.py 2000 3032; } > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
echo done
