#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args]
# kernel-trace + stats of bench.py into gpurun_out/<tag>/, prints a compact per-kernel table.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
grep '^{' $out/bench.log | tail -1 > $out/bench.json
python3 - "$out" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
for f in glob.glob(out + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:58].ljust(60), r['Calls'].rjust(5), ('%.1f' % (float(r['AverageNs'])/1e3)).rjust(9), 'us', r['Percentage'].rjust(6), '%')
try:
    d = json.load(open(out + '/bench.json'))
    print('value', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], 'avg_us', d['roofline']['avg_launch_us'])
except Exception as e:
    print('no bench json', e)
PY
