#!/bin/bash
# per-dispatch counters of the last decomposition (levels 1..7 of k_extract<double>), one PMC pass
tag=$1; shift; ctrs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --warm-ms 0 --steps 1 --warmup 1 > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = []
for f in glob.glob(out + '/*/*counter_collection.csv'):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    if 'k_extract' not in r['Kernel_Name'] and 'k_detect' not in r['Kernel_Name']: continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'][:34]})
    d[r['Counter_Name']] = float(r['Counter_Value'])
keys = sorted({k for d in disp.values() for k in d if k != 'name'})
print('disp'.ljust(6), 'kernel'.ljust(36), ' '.join(k[-14:].rjust(14) for k in keys))
for i, d in list(disp.items())[-10:]:
    print(str(i).ljust(6), d['name'].ljust(36), ' '.join(('%.4g' % d.get(k, 0)).rjust(14) for k in keys))
PY
