cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4a
for v in r03 notie new; do
  if [ $v = new ]; then unset PYITD_HIP_LIB; else export PYITD_HIP_LIB=$R/variants/lib$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4a/prof_$v -o bench -- python3 $R/bench.py --no-extra --no-cpu-baseline --warm-ms 20 --steps 50 > $R/gpurun_out/r4a/prof_$v.json 2> $R/gpurun_out/r4a/prof_$v.err || exit 1
  f=$(find $R/gpurun_out/r4a/prof_$v -name '*kernel_stats.csv' | head -1); echo "== $v"; head -8 $f | cut -c1-150
done
