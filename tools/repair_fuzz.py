"""Results that are final on the stream (itd_set_valid_flags / itd_set_device_repair) on random batches: a copy of rows_dev enqueued behind the
decomposition on the same stream, NO itd_get_summary in between — with the device-side repair every signal's rows as that consumer saw them
equal the oracle's bit for bit; without it the validity words say which ones do.  Signals of 65 536 .. 200 000 samples (the fused levels run),
families that the fused form delivers and families it refuses (coarse quantisation, plateaus, chirps), several calls per engine.
usage: python tools/repair_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import fuzz_signal, assert_bits_equal, sines_noise, chirp
from oracle import cpu_oracle as O
import pyitd_amd as P
from pyitd_amd.engine import FUSE_AUTO

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = repaired = invalid = 0
t0 = time.time()
for case in range(cases):
    n = int(rng.choice([65536, 70001, 1 << 17, 200000]))
    B = int(rng.integers(1, 6))
    m = int(rng.integers(3, 9))
    repair = bool(case % 3 != 2)
    eng = P.Engine(n, B, 0)
    eng.set_fuse_mode(FUSE_AUTO)
    eng.set_fuse_min_samples(65536)
    valid = torch.full((B,), -1, dtype=torch.int32, device="cuda")
    eng.set_valid_flags(valid.data_ptr())
    eng.set_device_repair(repair)
    s = torch.cuda.Stream()
    for call in range(int(rng.integers(1, 4))):            # (the back-offs and the generation counters carry over from call to call)
        xs = []
        for b in range(B):
            fam = int(rng.integers(0, 6))
            if fam == 0:
                x = sines_noise(n, seed=int(rng.integers(0, 1 << 30)))
            elif fam == 1:
                x = chirp(n)
            elif fam == 2:
                x = np.round(fuzz_signal(rng, 0, n) * int(rng.integers(2, 40))) / 8.0
            elif fam == 3:
                x = np.concatenate([np.zeros(int(rng.integers(1, 5000))), sines_noise(n, seed=int(rng.integers(0, 1 << 30)))])[:n]
            else:
                x = fuzz_signal(rng, int(rng.integers(0, 8)), n)
            xs.append(np.asarray(x, dtype=np.float32))
        xs = np.stack(xs)
        with np.errstate(all="ignore"):
            refs = [O.itd(x, m) for x in xs]
        xd = torch.from_numpy(xs).cuda()
        rows = torch.full((B, m + 2, n), float("nan"), dtype=torch.float64, device="cuda")
        seen = torch.empty_like(rows)
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            eng.decompose_dev(xd.data_ptr(), np.float32, n, B, n, m, rows.data_ptr(), None, s.cuda_stream)
            seen.copy_(rows, non_blocking=True)              # the stream-ordered consumer
            v = valid.clone()
        s.synchronize()
        v = v.cpu().numpy()
        what = "case %d call %d (%d x %d, %d levels, repair %s)" % (case, call, B, n, m, repair)
        try:
            if repair:
                nan_in = [bool(np.isnan(x).any()) for x in xs]
                assert all(v[b] == 1 or nan_in[b] for b in range(B)), what + ": validity %s" % v.tolist()
            for b in range(B):
                if v[b] == 1:
                    nr = refs[b]["rows"].shape[0]
                    assert_bits_equal(seen[b, :nr].cpu().numpy(), refs[b]["rows"], what + " signal %d as the consumer saw it" % b)
                else:
                    invalid += 1
            summ = eng.summary(B)
            for b in range(B):
                nr = int(summ["n_rows"][b])
                assert nr == refs[b]["rows"].shape[0], what + " signal %d rows" % b
                assert_bits_equal(rows[b, :nr].cpu().numpy(), refs[b]["rows"], what + " signal %d after the summary" % b)
        except AssertionError as ex:
            bad += 1
            print("MISMATCH " + str(ex)[:260])
    repaired += eng.device_repairs
    eng.close()
print("%d cases, %d mismatches; %d signals repaired on the device, %d reported not final (no repair asked for), %.1f s" % (cases, bad, repaired, invalid, time.time() - t0))
sys.exit(1 if bad else 0)
