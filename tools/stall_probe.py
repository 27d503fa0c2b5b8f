"""Where do the sporadic ~50 ms stalls of a launch loop come from?  Host time of every decompose call and of every
synchronisation, plus the GPU's own time per group of calls (events).  (GPU box)  usage: stall_probe.py [n] [groups]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyitd_amd  # noqa: E402
M = 7
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 200
per = 50
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)
t = torch.arange(n, dtype=torch.float64) / 8000.0
x = (torch.sin(2 * np.pi * 110 * t)[None, :] + 0.05 * torch.randn((1, n), generator=g, dtype=torch.float64)).to(torch.float32).to(dev)
rows = torch.empty((1, M + 2, n), dtype=torch.float64, device=dev)
eng = pyitd_amd.Engine(n, 1, 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
call_max, sync_t, gpu_t, tot_t = [], [], [], []
with torch.cuda.stream(stream):
    for gi in range(groups):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        worst = 0.0
        for _ in range(per):
            a = time.perf_counter()
            eng.decompose_dev(x.data_ptr(), np.float32, n, 1, n, M, rows.data_ptr(), None, stream.cuda_stream)
            worst = max(worst, time.perf_counter() - a)
        e1.record(stream)
        b = time.perf_counter()
        stream.synchronize()
        c = time.perf_counter()
        call_max.append(worst * 1e3); sync_t.append((c - b) * 1e3); gpu_t.append(e0.elapsed_time(e1)); tot_t.append((c - t0) * 1e3)
tot = np.array(tot_t)
print("n = %d, %d groups of %d decompositions: group wall ms median %.3f, max %.3f; groups over 3x median: %d" % (
    n, groups, per, np.median(tot), tot.max(), int((tot > 3 * np.median(tot)).sum())))
for gi in np.nonzero(tot > 3 * np.median(tot))[0][:10]:
    print("  group %d: wall %.2f ms, slowest single call on the host %.2f ms, wait in synchronize %.2f ms, GPU span by events %.2f ms" % (
        gi, tot_t[gi], call_max[gi], sync_t[gi], gpu_t[gi]))
print("typical group: slowest call %.3f ms, synchronize %.3f ms, GPU span %.3f ms" % (np.median(call_max), np.median(sync_t), np.median(gpu_t)))
eng.close()
