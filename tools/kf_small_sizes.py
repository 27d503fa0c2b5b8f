"""The fused sparse levels forced (ITD_FUSE_ONLY) on sizes they are never chosen for: n = 3 ... 70 000 around the tile boundaries,
few and many levels, all fuzz families.  Every case must either equal the oracle bit for bit or be refused — never wrong, never a
fault.  usage (GPU box): python tools/kf_small_sizes.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_ONLY, RESIDENT_OFF
from pyitd_amd import ITDError
from oracle import cpu_oracle
from helpers import fuzz_signal, canon_u64

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
tally = {"delivered": 0, "refused": 0, "WRONG": 0}
for c in range(cases):
    n = int(rng.choice([3, 4, 5, 63, 64, 65, 511, 512, 513, 1023, 1024, 1025, 1537, 4097, int(rng.integers(3, 3000)), int(rng.integers(3, 70000))]))
    m = int(rng.integers(2, 12))
    L0 = int(rng.integers(2, m + 1))
    kind = int(rng.integers(0, 8))
    x = fuzz_signal(rng, kind, n)
    if not np.all(np.isfinite(x)):
        continue
    if kind != 7 and rng.random() < 0.5:
        x = x.astype(np.float32)
    ref = cpu_oracle.itd_lean(x, m)
    eng = pyitd_amd.Engine(n, 1, 0)
    eng.set_resident_mode(RESIDENT_OFF)
    eng.set_fuse_mode(FUSE_ONLY)
    eng.set_fuse_level(L0)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    try:
        eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), None, None)
        s = eng.summary(1)
        nr = int(s["n_rows"][0])
        ok = nr == ref["rows"].shape[0] and np.array_equal(canon_u64(rows[:nr].cpu().numpy()), canon_u64(ref["rows"]))
        tally["delivered" if ok else "WRONG"] += 1
        if not ok:
            print("WRONG: case %d kind %d n %d m %d L0 %d %s rows %d vs %d" % (c, kind, n, m, L0, x.dtype, nr, ref["rows"].shape[0]), flush=True)
    except ITDError:
        tally["refused"] += 1
    eng.close()
print(tally)
sys.exit(1 if tally["WRONG"] else 0)
