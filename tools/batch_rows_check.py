"""Which signals of a batch report an unexpected row count, and under which call pattern?
usage (GPU box): python tools/batch_rows_check.py [batch] [log2n]
Runs the bench's configs[2] batch (device-synthesised signals) through one engine several times — NULL stream / a caller
stream, one call / repeated calls before the summary is read — and prints the signals whose n_rows differs from m + 2."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import pyitd_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
LOG2N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
M = 7


def report(tag, s):
    bad = np.nonzero(s["n_rows"] != M + 2)[0]
    print("%-44s bad=%d %s" % (tag, len(bad), bad[:24].tolist()), flush=True)
    for b in bad[:4]:
        print("    signal %d: n_rows %d n_baselines %d stop %d nan_level %d knots %s" % (
            b, s["n_rows"][b], s["n_baselines"][b], s["stop"][b], s["nan_levels"][b], s["knot_counts"][b, :M + 3].tolist()), flush=True)
    return bad


def main():
    dev = torch.device("cuda:0")
    n = 1 << LOG2N
    x = bench.batch_signals_device(torch, dev, 0, B, n)
    bad_in = (~torch.isfinite(x)).any(dim=1).nonzero().flatten().tolist()
    print("signals with non-finite samples in the input:", bad_in, flush=True)
    rows = torch.empty((B, M + 2, n), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    for streams in (None, 1, 2):
        eng = pyitd_amd.Engine(n, B, 0)
        if streams is not None:
            eng.set_batch_streams(streams)
        tag = "streams=%s" % ("default" if streams is None else streams)
        eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
        report(tag + " NULL stream, call 1", eng.summary(B))
        eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, None)
        report(tag + " NULL stream, call 2", eng.summary(B))
        for k in range(2):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, stream.cuda_stream)
        report(tag + " caller stream, 2 calls, 1 summary", eng.summary(B))
        for k in range(3):
            eng.decompose_dev(x.data_ptr(), np.float32, n, B, n, M, rows.data_ptr(), None, stream.cuda_stream)
        bad = report(tag + " caller stream, 3 calls, 1 summary", eng.summary(B))
        torch.cuda.synchronize()
        if len(bad):
            b = int(bad[0])
            rec = rows[b].sum(dim=0)
            print("    signal %d: reconstruction error %.3e, row 1 abs max %.3e" % (
                b, float((rec - x[b].double()).abs().max()), float(rows[b, 1].abs().max())), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
