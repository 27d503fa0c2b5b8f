#!/bin/bash
# usage (GPU box, repo root): bash tools/traffic.sh <round-tag>
# HBM-side bytes per launch of every kernel of bench.py from the PMC counters FETCH_SIZE and WRITE_SIZE (separate passes,
# kernel-trace only), with the gfx950 correction of MI355X_MICROARCH.md applied; writes gpurun_out/traffic.json
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/traffic_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra --warm-ms 0 --steps 3 --warmup 1 > $out/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - "$out" "$tag" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
allk = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + c + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    allk[c] = {k: {"dispatches": len(v), "mean_KB": sum(v) / len(v)} for k, v in acc.items()}
# the level-by-level launch of a level >= 1 (with the sparse levels fused from level 2 only its TIES form runs: level 1)
dom = [k for k in allk["FETCH_SIZE"] if "k_extract<double" in k and ", false," in k and "true>" not in k] or \
      [k for k in allk["FETCH_SIZE"] if "k_extract<double" in k]
dom = dom[0] if dom else None
import hashlib, os
sys.path.insert(0, os.getcwd())
from pyitd_amd import _lib
res = {"round": tag, "kernel": dom, "all_kernels": allk,
       "library_build_id": hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16],
       "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section) -> reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for streaming stores",
       "commands": ["rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --warm-ms 0 --steps 3 --warmup 1",
                    "rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-extra --warm-ms 0 --steps 3 --warmup 1"]}
if dom:
    f, w = allk["FETCH_SIZE"][dom]["mean_KB"], allk["WRITE_SIZE"][dom]["mean_KB"]
    res["counters_KB_per_launch"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    res["k_extract_f64_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    res["algorithmic_bytes_per_launch"] = 24.0 * (1 << 24)
    res["ratio_traffic_over_algorithmic"] = res["k_extract_f64_bytes_per_launch"] / res["algorithmic_bytes_per_launch"]
    res["note"] = ("reads above the algorithmic 8 B/sample: the +-64-tile count windows (512 B per 512-sample tile), the tile's own 128-byte "
                   "record and the first 64 bytes of four neighbours' records; FETCH_SIZE counts requests that leave L2, Infinity-Cache hits included")
ap = [k for k in allk["FETCH_SIZE"] if "k_kf_apply" in k]
if ap:     # the fused sparse levels' sample pass: 8 B read + 8 B per row (9 rows - the first fused level at the headline configuration)
    f, w = allk["FETCH_SIZE"][ap[0]]["mean_KB"], allk["WRITE_SIZE"][ap[0]]["mean_KB"]
    n_f64 = sum(v["dispatches"] for k, v in allk["FETCH_SIZE"].items() if "k_extract<double" in k)
    n_f32 = sum(v["dispatches"] for k, v in allk["FETCH_SIZE"].items() if "k_extract<float" in k)
    L0 = 1 + round(n_f64 / max(n_f32, 1))
    res["first_fused_level"] = L0
    res["k_kf_apply_counters_KB_per_launch"] = {"FETCH_SIZE": f, "WRITE_SIZE": w}
    res["k_kf_apply_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    res["k_kf_apply_algorithmic_bytes_per_launch"] = (8.0 + 8.0 * (9 - L0)) * (1 << 24)
    res["k_kf_apply_ratio_traffic_over_algorithmic"] = res["k_kf_apply_bytes_per_launch"] / res["k_kf_apply_algorithmic_bytes_per_launch"]
json.dump(res, open(out + "/../traffic.json", "w"), indent=1)
print(json.dumps({k: res.get(k) for k in ("kernel", "counters_KB_per_launch", "k_extract_f64_bytes_per_launch", "ratio_traffic_over_algorithmic", "k_kf_apply_bytes_per_launch", "k_kf_apply_ratio_traffic_over_algorithmic")}))
PY
