import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pyitd_amd.engine import Engine, CHAIN_AUTO
from tests.helpers import sines_noise
for lg, M, dt in ((20, 11, np.float64), (20, 11, np.float32), (20, 7, np.float32), (20, 9, np.float32), (18, 11, np.float32)):
    n = 1 << lg
    x = torch.from_numpy(sines_noise(n).astype(dt)).cuda()
    eng = Engine(n, 1)
    rows = torch.empty((M + 2, n), dtype=torch.float64, device="cuda")
    out = (ctypes.c_uint64 * 16)()
    for rep in range(3):
        t0 = time.perf_counter()
        eng.decompose_dev(x.data_ptr(), dt, n, 1, n, M, rows.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        s = eng.summary(1)
        dtm = time.perf_counter() - t0
        eng._L.itd_debug_chain_prof(eng._h, out, 0)
        print(lg, M, dt.__name__, "rep", rep, "%.3f ms" % (dtm * 1e3), "repeats", eng.chain_repeats, "give_up", int(out[15]), "rows", s["n_rows"], "knots", s["knot_counts"][0][:M + 3])
    eng.close()
