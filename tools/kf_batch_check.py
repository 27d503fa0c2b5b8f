"""The fused sparse levels on a BATCH (bench.py's configs[2] recipe): does every signal of the batch take the fused form
(ITD_FUSE_ONLY: a refusal is an error carrying the failure bits), and what does a batch cost in each mode?
usage (GPU box): python tools/kf_batch_check.py [batch] [log2n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyitd_amd
from pyitd_amd.engine import FUSE_AUTO, FUSE_OFF, FUSE_ONLY
from pyitd_amd import ITDError
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, M = 1 << log2n, 7
dev = torch.device("cuda:0")
x = bench.batch_signals_device(torch, dev, 0, batch, n)
rows = torch.empty((batch, M + 2, n), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for name, mode in (("only", FUSE_ONLY), ("auto", FUSE_AUTO), ("off", FUSE_OFF)):
    eng = pyitd_amd.Engine(n, batch, 0)
    eng.set_fuse_mode(mode)
    try:
        for _ in range(2):
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
        s = eng.summary(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fix0 = eng.fuse_signal_repairs
        for _ in range(5):
            eng.decompose_dev(x.data_ptr(), np.float32, n, batch, n, M, rows.data_ptr(), None, None)
            s = eng.summary(batch)          # (the refusing signals are re-run here: inside the timed region)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("%-5s %8.3f ms  %.1f Gsamples/s  rows %s  whole-call repeats %d, signals re-run on their own per call %.1f" % (
            name, dt * 1e3, batch * n / dt / 1e9, sorted(set(int(v) for v in s["n_rows"])), eng.fuse_repeats, (eng.fuse_signal_repairs - fix0) / 5), flush=True)
    except ITDError as ex:
        print("%-5s refused: %s" % (name, str(ex)[-160:]), flush=True)
        # which signals?  one at a time
        bad = []
        for b in range(min(batch, 64)):
            e1 = pyitd_amd.Engine(n, 1, 0)
            e1.set_fuse_mode(FUSE_ONLY)
            try:
                e1.decompose_dev(x[b].data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, None)
                e1.summary(1)
            except ITDError as ex1:
                bad.append((b, str(ex1)[str(ex1).find("fail bits"):][:14]))
            e1.close()
        print("      signals refused on their own (first 64):", bad, flush=True)
    eng.close()
