#!/bin/bash
# round 2, sixth session: the GPU suite in the engine's other diagnostic modes (record-driven level 0, the one-launch chain) after the
# SigState / dispatch changes of the resident form, and a longer fuzz
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6q
mkdir -p $O
PYITD_LEVEL0_MODE=1 timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest_gpu_record_driven_level0.log 2>&1; tail -1 $O/pytest_gpu_record_driven_level0.log
PYITD_CHAIN_MODE=0 timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu_chain_mode.log 2>&1; tail -1 $O/pytest_gpu_chain_mode.log
{ echo "-- resident only, single, 3 .. 8192, NaN / inf inputs mixed in"; PYITD_RESIDENT_MODE=2 FUZZ_MAX_N=8192 timeout -k 10 600 python tools/fuzz_parity.py 100000 6060;
  echo "-- resident only, batch"; PYITD_RESIDENT_MODE=2 FUZZ_MAX_N=8192 timeout -k 10 600 python tools/fuzz_parity.py batch 5000 6061;
  echo "-- automatic, all lengths, single"; timeout -k 10 600 python tools/fuzz_parity.py 10000 6062; } > $O/fuzz_long.txt 2>&1
grep -c "MISMATCH\|EXCEPTION" $O/fuzz_long.txt; grep "cases,\|batches,\|^--" $O/fuzz_long.txt
