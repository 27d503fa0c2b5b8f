#!/bin/bash
# rocprofv3 kernel-trace timelines of a short fused batch: tools/pipeline_trace.sh TAG "batch pipe calls streams chunk" ...  -> gpurun_out/TAG/timeline_<args>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
for a in "$@"; do
    name=$(echo "$a" | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$TAG/t_$name -o trace -- python3 $R/tools/pipeline_trace.py $a > $R/gpurun_out/$TAG/run_$name.log 2>&1 || exit 1
    f=$(find $R/gpurun_out/$TAG/t_$name -name "*kernel_trace.csv" | head -1)
    python3 $R/tools/timeline.py $f 90 > $R/gpurun_out/$TAG/timeline_$name.txt
    rm -rf $R/gpurun_out/$TAG/t_$name
done
