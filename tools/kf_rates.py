"""How often do the fused sparse levels (itd_set_fuse_mode) deliver, per signal family?  ITD_FUSE_ONLY on random draws: a case
either equals the C oracle bit for bit ("delivered") or the engine reports that the fused form cannot deliver it ("refused",
with the failure bits: 1 verification, 2 capacity, 4 non-finite, 8 ties, 16 a halo wait given up / a neighbour that gave up) — a third outcome would be a bug and is counted as
WRONG.  usage (GPU box): python tools/kf_rates.py [cases per family] [seed]"""
import os, sys, re
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_ONLY
from pyitd_amd import ITDError
from oracle import cpu_oracle
from helpers import canon_u64, kf_rates_draws

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
NAMES = ["white noise", "random walk", "quantised (plateaus)", "smooth + few knots", "sine + noise at a random level",
         "constant stretches with bursts", "alternating, random amplitudes", "extreme magnitudes", "sines + noise (the bench signal)",
         "the bench signal as 16-bit PCM", "the bench signal as 12-bit PCM"]


def one(x, m):
    n = len(x)
    ref = cpu_oracle.itd_lean(x, m)
    eng = pyitd_amd.Engine(n, 1, 0)
    eng.set_fuse_mode(FUSE_ONLY)
    xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    rows = torch.full((m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()        # (the fill runs on torch's stream, the engine on its own)
    try:
        for tiles in (64, 32, 16):          # (what the automatic mode does over consecutive calls: a list that outgrew its workgroup halves the ranges)
            eng.set_fuse_range(tiles)
            try:
                eng.decompose_dev(xd.data_ptr(), x.dtype, n, 1, n, m, rows.data_ptr(), None, None)
                s = eng.summary(1)
            except ITDError as ex:
                mm = re.search(r"fail bits (0x[0-9a-f]+)", str(ex))
                code = mm.group(1) if mm else "?"
                if code != "?" and (int(code, 16) & 2) and tiles > 16:      # capacity (+ 16: neighbours that met the workgroup that gave up)
                    continue
                return "refused " + code
            nr = int(s["n_rows"][0])
            ok = nr == ref["rows"].shape[0] and np.array_equal(canon_u64(rows[:nr].cpu().numpy()), canon_u64(ref["rows"]))
            if not ok:
                global n_wrong
                np.save("gpurun_out/wrong_case_%d.npy" % n_wrong, x)
                got = rows[:nr].cpu().numpy()
                bad = np.argwhere(canon_u64(got[: ref["rows"].shape[0]]) != canon_u64(ref["rows"][:nr]))
                print("   WRONG: n %d dtype %s m %d tiles %d rows %d/%d first mismatches %s -> gpurun_out/wrong_case_%d.npy"
                      % (n, x.dtype, m, tiles, nr, ref["rows"].shape[0], bad[:4].tolist(), n_wrong), flush=True)
                n_wrong += 1
            return ("delivered" if ok else "WRONG") + ("" if tiles == 64 else " (%d tiles)" % tiles)
    finally:
        eng.close()


n_wrong = 0


print("%-36s %-6s %s" % ("family", "levels", "outcomes over %d draws (n in 70 000 .. 400 000, float32 / float64)" % cases))
wrong = 0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
tallies = {}
for idx, kind, m, x in kf_rates_draws(seed, cases):      # (tests/helpers.py: the sweep's seed and a draw's index name a case for good)
    r = one(x, m)
    if r.startswith("WRONG"):
        print("   draw %d of kf_rates_draws(seed=%d, cases=%d)" % (idx, seed, cases), flush=True)
    t = tallies.setdefault((kind, m), {})
    t[r] = t.get(r, 0) + 1
    wrong += r == "WRONG"
for (kind, m), tally in sorted(tallies.items()):
    print("%-36s %-6d %s" % (NAMES[kind], m + 1, ", ".join("%s: %d" % kv for kv in sorted(tally.items()))), flush=True)
print("WRONG results: %d" % wrong)
sys.exit(1 if wrong else 0)
