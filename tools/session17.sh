#!/bin/bash
# round 2, sixth session: evidence of the final build (resident form with rank windows, n <= 8192) (GPU box).  Everything lands under gpurun_out/s6z/
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6z
mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; tail -2 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit $rc
PYITD_RESIDENT_MODE=1 timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest_gpu_resident_off.log 2>&1; rc=$?; tail -1 $O/pytest_gpu_resident_off.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err || exit 1
bash tools/prof.sh s6z/prof --steps 100 --warmup 30 > $O/kernel_stats.txt || exit 1
timeout -k 10 300 python bench.py --gpus 2 --rehearse-one-gpu --batch 128 --steps 5 --warmup 2 > $O/bench_2rank_rehearsal.json 2> /dev/null || exit 1
{ echo "-- resident range, single"; FUZZ_MAX_N=8192 FUZZ_NO_NAN=1 timeout -k 10 300 python tools/fuzz_parity.py 6000 4040;
  echo "-- resident range, single, NaN inputs mixed in"; FUZZ_MAX_N=8192 timeout -k 10 300 python tools/fuzz_parity.py 3000 4041;
  echo "-- resident range, batch"; FUZZ_MAX_N=8192 FUZZ_NO_NAN=1 timeout -k 10 300 python tools/fuzz_parity.py batch 600 4042;
  echo "-- all lengths, single"; timeout -k 10 300 python tools/fuzz_parity.py 3000 4043;
  echo "-- all lengths, batch"; timeout -k 10 300 python tools/fuzz_parity.py batch 300 4044; } > $O/fuzz.txt 2>&1; grep -c MISMATCH $O/fuzz.txt; grep "cases,\|batches," $O/fuzz.txt
echo done
