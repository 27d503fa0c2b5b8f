#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc.sh <tag> "<counters>" [bench args]
# one PMC pass of bench.py (kernel-trace only, as gpurun requires), prints per-kernel counter means.
tag=$1; shift
ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --warm-ms 0 "$@" > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%-4d mean=%.4g' % (c, len(v), sum(v) / len(v)))
PY
