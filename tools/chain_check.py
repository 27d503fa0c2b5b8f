"""Chain launch vs level-by-level engine on the GPU: rows bit for bit, summaries, timings.  usage: python tools/chain_check.py [log2n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyitd_amd.engine import Engine, CHAIN_OFF, CHAIN_ONLY, CHAIN_AUTO, TIME_CHAIN, TIME_DECOMPOSE  # noqa: E402
from tests.helpers import sines_noise, fuzz_signal  # noqa: E402


def run(eng, x_t, n, B, M, mode, want_bases, reps=1):
    eng.set_chain_mode(mode)
    rows = torch.full((B, M + 2, n), float("nan"), dtype=torch.float64, device="cuda")
    bases = torch.full((B, M + 2, n), float("nan"), dtype=torch.float64, device="cuda") if want_bases else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.decompose_dev(x_t.data_ptr(), np.float32 if x_t.dtype == torch.float32 else np.float64, n, B, n, M, rows.data_ptr(),
                          bases.data_ptr() if want_bases else None, torch.cuda.current_stream().cuda_stream)
    s = eng.summary(B)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return rows, bases, s, dt


def compare(tag, n, B, M, x, want_bases=False, reps=1):
    x_t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    eng = Engine(n, B)
    r0, b0, s0, t0 = run(eng, x_t, n, B, M, CHAIN_OFF, want_bases, reps)
    rep0 = eng.chain_repeats
    try:
        r1, b1, s1, t1 = run(eng, x_t, n, B, M, CHAIN_AUTO, want_bases, reps)
    except Exception as ex:  # noqa: BLE001
        print(tag, "CHAIN FAILED:", ex)
        return False
    repeated = eng.chain_repeats - rep0
    ok = True
    for b in range(B):
        nr = int(s0["n_rows"][b])
        if int(s1["n_rows"][b]) != nr or int(s1["stop"][b]) != int(s0["stop"][b]):
            print(tag, "summary differs", b, s0["n_rows"][b], s1["n_rows"][b], s0["stop"][b], s1["stop"][b])
            ok = False
            continue
        if not np.array_equal(s0["knot_counts"][b], s1["knot_counts"][b]):
            print(tag, "knot counts differ", b, s0["knot_counts"][b][:M + 4], s1["knot_counts"][b][:M + 4])
            ok = False
        a = r0[b, :nr].view(torch.int64)
        c = r1[b, :nr].view(torch.int64)
        nan_both = torch.isnan(r0[b, :nr]) & torch.isnan(r1[b, :nr])
        bad = ((a != c) & ~nan_both)
        nbad = int(bad.sum())
        if nbad:
            idx = torch.nonzero(bad)[0].tolist()
            print(tag, "rows differ: signal", b, nbad, "values; first at", idx, float(r0[b, idx[0], idx[1]]), float(r1[b, idx[0], idx[1]]))
            ok = False
        if want_bases:
            nb = int(s0["n_baselines"][b])
            a = b0[b, :nb].view(torch.int64)
            c = b1[b, :nb].view(torch.int64)
            nan_both = torch.isnan(b0[b, :nb]) & torch.isnan(b1[b, :nb])
            nbad = int(((a != c) & ~nan_both).sum())
            if nbad:
                print(tag, "baselines differ: signal", b, nbad)
                ok = False
    print("%-44s %s  level-by-level %.3f ms, chain %.3f ms%s" % (tag, "ok " if ok else "BAD", t0 * 1e3, t1 * 1e3, "  (repeated level by level)" if repeated else ""), flush=True)
    eng.close()
    return ok


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [12, 16, 20, 24]
    ok = True
    rng = np.random.default_rng(1)
    for n in (3, 4, 5, 100, 511, 512, 513, 1024, 1500, 5000, 65553):
        for kind in (0, 2, 3, 5):
            for dt in (np.float32, np.float64):
                x = fuzz_signal(rng, kind, n).astype(dt)
                ok &= compare("fuzz n=%d kind=%d %s M=5" % (n, kind, dt.__name__), n, 1, 5, x, want_bases=True)
    xb = np.stack([sines_noise(1 << 14, seed=b, fscale=1 + b / 8192) for b in range(24)])
    ok &= compare("batch 24 x 2^14 M=7", 1 << 14, 24, 7, xb)
    ok &= compare("batch 24 x 2^14 M=7 +bases", 1 << 14, 24, 7, xb, want_bases=True)
    for lg in sizes:
        n = 1 << lg
        x = sines_noise(n)
        ok &= compare("sines+noise 2^%d f32 M=7" % lg, n, 1, 7, x, reps=3)
        if lg <= 22:
            ok &= compare("sines+noise 2^%d f64 M=11 +bases" % lg, n, 1, 11, x.astype(np.float64), want_bases=True)
    print("ALL OK" if ok else "FAILURES")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
