cd "${GRAFT_REPO_ROOT}" || exit 1
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 300 python tools/meitd_fuzz.py 3000 17 > $O/meitd_fuzz_3000.txt 2>&1; tail -1 $O/meitd_fuzz_3000.txt
timeout -k 10 300 python tools/stream_fuzz.py 3000 19 > $O/stream_fuzz_3000.txt 2>&1; tail -1 $O/stream_fuzz_3000.txt
timeout -k 10 300 python tools/ops_fuzz.py 20000 12 2>/dev/null > $O/ops_fuzz_20000.txt; tail -1 $O/ops_fuzz_20000.txt
timeout -k 10 300 python tools/tfe_fuzz.py 2000 11 > $O/tfe_fuzz_2000.txt 2>&1; tail -1 $O/tfe_fuzz_2000.txt
timeout -k 10 300 python tools/spline_fuzz.py 2000 12 > $O/spline_fuzz_2000.txt 2>&1; tail -1 $O/spline_fuzz_2000.txt
timeout -k 10 300 python tools/batch_ops_fuzz.py 4000 12 > $O/batch_ops_fuzz_4000.txt 2>&1; tail -1 $O/batch_ops_fuzz_4000.txt
timeout -k 10 300 python tools/threads_fuzz.py 1500 4 5 > $O/threads_fuzz_4x1500.txt 2>&1; tail -1 $O/threads_fuzz_4x1500.txt
