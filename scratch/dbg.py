import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyitd_amd
from oracle import cpu_oracle
cpu_oracle.lib()
rng = np.random.default_rng(0)
def check(name, x, m):
    ref = cpu_oracle.itd(x, m)
    eng = pyitd_amd.Engine(len(x), 1, 0)
    out = eng.decompose_host(x, m, True)
    rows = out["rows"] if isinstance(out, dict) else out[0]
    kc = out["knot_counts"] if isinstance(out, dict) else None
    print(name, "n", len(x), "rows", rows.shape[0], "ref rows", ref["rows"].shape[0], "kc", None if kc is None else list(kc[:6]), "ref kc", list(ref["knot_counts"][:6]))
    R = min(rows.shape[0], ref["rows"].shape[0])
    for j in range(R):
        a = rows[j].view(np.uint64); b = ref["rows"][j].view(np.uint64)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            print("  row", j, "mismatches", len(bad), "first", bad[:8], "gpu", rows[j][bad[:3]], "ref", ref["rows"][j][bad[:3]])
            break
    else:
        print("  rows ok")
for n in (100, 512, 513, 700, 1024, 5000):
    check("noise", rng.standard_normal(n), 3)
t = np.arange(20000) / 20000.0
check("chirp", np.sin(2 * np.pi * (5 + 200 * t) * t), 4)
check("chirp32", np.sin(2 * np.pi * (5 + 200 * t) * t).astype(np.float32), 4)
