#!/bin/bash
# usage: bash tools/isa_stats.sh [out-file] [extra hipcc -D flags...]
# Per-kernel resource usage of the shipped build (hipcc -Rpass-analysis=kernel-resource-usage): VGPRs, SGPRs, spills,
# occupancy (waves per SIMD), LDS bytes.  Cross-compiles for gfx950, no GPU needed.  Kept per round under profiles/.
out=${1:-/dev/stdout}; shift
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Rpass-analysis=kernel-resource-usage "$@" \
      -o /tmp/isa_stats_$$.so pyitd_amd/csrc/itd_engine.hip 2>&1 | python3 -c '
import re, sys
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: .*Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
import subprocess
def dem(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        return n
print("%-62s %5s %5s %6s %6s %4s %6s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "occ", "LDS"))
for r in rows:
    print("%-62s %5s %5s %6s %6s %4s %6s" % (dem(r["name"])[:62], r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("VGPRs Spill", "?"),
                                             r.get("SGPRs Spill", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
' > "$out"
rm -f /tmp/isa_stats_$$.so
