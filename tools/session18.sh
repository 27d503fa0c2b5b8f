#!/bin/bash
# round 2, sixth session: fuzz of the resident form with the repeat forbidden (PYITD_RESIDENT_MODE=2: any signal the kernel could not
# finish itself would be an EXCEPTION) — NaN / infinite inputs, plateaus, all families, lengths 3 .. 8192
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6k
mkdir -p $O
{ echo "-- resident only, single, NaN / inf inputs mixed in"; PYITD_RESIDENT_MODE=2 FUZZ_MAX_N=8192 timeout -k 10 400 python tools/fuzz_parity.py 20000 5050;
  echo "-- resident only, batch, NaN / inf inputs mixed in"; PYITD_RESIDENT_MODE=2 FUZZ_MAX_N=8192 timeout -k 10 400 python tools/fuzz_parity.py batch 1500 5051;
  echo "-- automatic mode, all lengths, single"; timeout -k 10 300 python tools/fuzz_parity.py 4000 5052;
  echo "-- automatic mode, all lengths, batch"; timeout -k 10 300 python tools/fuzz_parity.py batch 400 5053; } > $O/fuzz.txt 2>&1
grep -c "MISMATCH\|EXCEPTION" $O/fuzz.txt; grep "cases,\|batches,\|^--" $O/fuzz.txt; grep "MISMATCH\|EXCEPTION" $O/fuzz.txt | head -5
