"""Block-wise operation (itd_stream_*): microseconds per pushed block, device form (asynchronous pushes, one synchronisation
at the end) and host form (numpy in / out, one synchronisation per push), cubic and tier-1 operator, 1 and 8 channels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyitd_amd import streaming

rng = np.random.default_rng(1)
for L in (4096, 65536):
    nb = 64 if L == 4096 else 16
    for C in (1, 8):
        x = np.cumsum(rng.standard_normal((C, L * nb)), axis=1) * 0.05 + np.sin(np.arange(L * nb) / 40.0)
        xd = torch.from_numpy(x).cuda()
        for kind in ("cubic", "linear"):
            st = streaming.Stream(L, C, kind, margin=8)
            base, rot = torch.empty_like(xd), torch.empty_like(xd)
            n = L * nb
            s = torch.cuda.current_stream().cuda_stream

            def run():
                for k in range(nb):
                    o = max(k - 1, 0) * L
                    st.push_dev(xd[:, k * L:].data_ptr(), n, base[:, o:].data_ptr(), n, rot[:, o:].data_ptr(), n, s)
                st.flush_dev(base[:, (nb - 1) * L:].data_ptr(), n, rot[:, (nb - 1) * L:].data_ptr(), n, s)
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            dev_us = (time.perf_counter() - t0) / 5 / nb * 1e6
            t0 = time.perf_counter()
            for k in range(nb):
                st.push(x[:, k * L:(k + 1) * L])
            st.flush()
            host_us = (time.perf_counter() - t0) / nb * 1e6
            print("block %6d x %d channels, %-6s: %7.1f us per block device form (%.0f Msamples/s), %7.1f us host form" % (
                L, C, kind, dev_us, L * C / dev_us, host_us))
            st.close()
