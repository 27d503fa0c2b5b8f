"""The fused sparse levels (itd_set_fuse_mode) against the CPU oracle and the level-by-level engine: rows, knot counts, stop
reasons bit for bit, or an honest refusal (FUSE_ONLY makes a failed verification an error instead of a silent repeat)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO, FUSE_OFF, FUSE_ONLY
from pyitd_amd import ITDError
from oracle import cpu_oracle
from helpers import sines_noise, fuzz_signal, chirp, load_golden, canon_u64

def run(name, x, m, L0=3, bases=False):
    n = len(x)
    ref = cpu_oracle.itd_lean(x, m)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    R = m + 2
    out = {}
    for mode in (FUSE_ONLY, FUSE_AUTO, FUSE_OFF):
        eng = pyitd_amd.Engine(n, 1, 0)
        eng.set_fuse_mode(mode); eng.set_fuse_level(L0)
        rows = torch.full((R, n), float("nan"), dtype=torch.float64, device="cuda")
        bs = torch.full((R, n), float("nan"), dtype=torch.float64, device="cuda") if bases else None
        torch.cuda.synchronize()
        try:
            eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), bs.data_ptr() if bases else None, None)
            s = eng.summary(1)
            nr = int(s["n_rows"][0])
            ok = nr == ref["rows"].shape[0] and np.array_equal(canon_u64(rows[:nr].cpu().numpy()), canon_u64(ref["rows"]))   # (NaN payloads canonical, as in the tests)
            kc = [int(v) for v in s["knot_counts"][0] if v >= 0]
            out[mode] = "ok" if ok else "MISMATCH rows %d vs %d" % (nr, ref["rows"].shape[0])
            if mode == FUSE_AUTO: out[mode] += " (repeats %d)" % eng.fuse_repeats
            if ok and kc[1:1 + len(ref["knot_counts"])] != ref["knot_counts"].tolist(): out[mode] += " KNOTCOUNTS %s vs %s" % (kc, ref["knot_counts"].tolist())
        except ITDError as ex:
            out[mode] = "refused: " + str(ex)[-60:]
        eng.close()
    print("%-26s n=%-9d m=%d L0=%d stop=%-7s only: %-40s auto: %-18s off: %s" % (name, n, m, L0, ref["stop"], out[FUSE_ONLY], out[FUSE_AUTO], out[FUSE_OFF]), flush=True)

radio = load_golden("radio8000_input")["x"]
run("sines 2^20 f32", sines_noise(1 << 20), 7)
run("sines 2^20 f32 L0=2", sines_noise(1 << 20), 7, 2)
run("sines 2^20 f32 L0=5", sines_noise(1 << 20), 7, 5)
run("sines 2^20 f32 bases", sines_noise(1 << 20), 7, 3, True)
run("sines 2^20 f64", sines_noise(1 << 20).astype(np.float64), 7)
run("sines 2^22 s3", sines_noise(1 << 22, seed=3, fscale=1 + 5 / 8192.), 7)
run("sines 2^24", sines_noise(1 << 24), 7)
run("sines 2^20 m=11", sines_noise(1 << 20, seed=2), 11)
run("sines ragged", sines_noise((1 << 20) + 777, seed=4), 7)
run("sines 2^17 m=20", sines_noise(1 << 17, seed=5), 20)
run("chirp 2^16", chirp(1 << 16), 5, 2)
run("radio tiled f32", np.resize(radio, 1 << 18).astype(np.float32), 9)
rng = np.random.default_rng(5)
for kind in range(8):
    x = fuzz_signal(rng, kind, 200000)
    if np.all(np.isfinite(x)):
        run("fuzz%d f64" % kind, x, 9)
