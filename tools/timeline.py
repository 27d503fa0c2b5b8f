"""Reads a rocprofv3 kernel-trace CSV and prints the last call's launches as a timeline: start / end (us from the first launch shown), queue, kernel.
usage: python tools/timeline.py trace_kernel_trace.csv [last N launches]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-N:]
t0 = int(rows[0]["Start_Timestamp"])
qs = {}
for r in rows:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    name = r["Kernel_Name"]
    name = name[name.find("k_"):][:40] if "k_" in name else name[:40]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f  %7.1f us  q%d %s%s" % (s, e, e - s, q, "        " * q, name))
