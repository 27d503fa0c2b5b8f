#!/bin/bash
# round-2 third session: evidence of the final build (GPU box).  Everything lands under gpurun_out/s3/
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
bash tools/prof.sh s3/prof --steps 100 --warmup 30 > gpurun_out/s3/kernel_stats.txt
bash tools/session7.sh > /dev/null
for i in 1 2 3 4 5; do echo "== pass $i"; cat gpurun_out/l0_p$i.txt; done > gpurun_out/s3/level0_counters.txt
bash tools/traffic.sh r02 > gpurun_out/s3/traffic_summary.txt
cp gpurun_out/traffic.json gpurun_out/s3/traffic.json
PYITD_HIP_LIB=variants/libprof.so python tools/level0_prof.py 24 > gpurun_out/s3/level0_phases.txt
python tools/graph_bench.py > gpurun_out/s3/graph_bench.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s3/bench_driver_form.json 2> gpurun_out/s3/bench_driver_form.err
python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s3/bench_cold_clocks.json
python bench.py --no-extra --no-cpu-baseline --steps 200 --warmup 60 > gpurun_out/s3/bench_200_steps.json
python tools/batch_bench.py > gpurun_out/s3/batch_1024x2p20.txt 2>&1 || true
echo done
