#!/bin/bash
# usage (GPU box, repo root): bash tools/resident_prof.sh
# The resident kernel under rocprofv3: kernel-trace stats of one shape (4096 x 4096 samples), then FETCH_SIZE / WRITE_SIZE in
# separate passes (gfx950 correction of MI355X_MICROARCH.md: reads = 2 x FETCH_SIZE); prints per-launch time and traffic against the
# algorithmic bytes (4 B in + 8 B per sample and written row).
out=$GRAFT_REPO_ROOT/gpurun_out/resident_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export SMALL_SHAPE=4096x4096
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/tools/small_batch_bench.py > $out/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/tools/small_batch_bench.py > $out/$c.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT
python3 - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
line = [l for l in open(out + "/stats.log") if " x " in l][-1]
rows = [int(v) for v in re.search(r"rows \[(.*)\]", line).group(1).split(",")]
print(line.strip())
for f in glob.glob(out + "/stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_resident" in r["Name"]:
            print("k_resident: %s launches, %.1f us average (rocprofv3 --kernel-trace --stats)" % (r["Calls"], float(r["AverageNs"]) / 1e3))
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(out + "/" + c + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "k_resident" in r["Kernel_Name"]:
                v.append(float(r["Counter_Value"]))
    tot[c] = sum(v) / max(len(v), 1)
reads, writes = 2 * tot["FETCH_SIZE"] * 1024, tot["WRITE_SIZE"] * 1024
n = 4096 * 4096
print("per launch: reads %.1f MB (2 x FETCH_SIZE), writes %.1f MB (WRITE_SIZE); per sample: %.2f B read, %.2f B written" % (reads / 1e6, writes / 1e6, reads / n, writes / n))
print("algorithmic: 4 B read per sample; written 8 B per sample and row, rows per signal in this batch: %s (min .. max = %.0f .. %.0f B per sample)" % (rows, 8 * min(rows), 8 * max(rows)))
PY
