"""The C ABI used from plain C: a gcc-built client links libpyitd_hip.so directly (no Python in the data path)."""
import os
import subprocess

import numpy as np
import pytest

from helpers import assert_bits_equal, sines_noise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_client", "abi_client.c")
LIBDIR = os.path.join(ROOT, "pyitd_amd")


@pytest.fixture(scope="module")
def oracle():
    from oracle import cpu_oracle
    return cpu_oracle


def _build(tmp_path):
    exe = str(tmp_path / "abi_client")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", LIBDIR, "-lpyitd_hip", "-Wl,-rpath," + LIBDIR]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c_and_client_links(tmp_path):
    """No GPU needed: the header compiles as C99 and every entry point the client uses resolves at link time."""
    import pyitd_amd._lib as L
    L.build()          # no-op when the in-tree library is up to date
    _build(tmp_path)


@pytest.mark.gpu
def test_c_client_matches_oracle(tmp_path, oracle):
    exe = _build(tmp_path)
    n, m = 70001, 5
    x = sines_noise(n, seed=5, dtype=np.float64)
    p = subprocess.run([exe, str(n), str(m)], input=x.tobytes(), capture_output=True, check=True)
    head, _, payload = p.stdout.partition(b"\n")
    n_rows, stop = (int(v) for v in head.split())
    rows = np.frombuffer(payload, dtype=np.float64).reshape(n_rows, n)
    ref = oracle.itd(x, m)
    assert n_rows == ref["rows"].shape[0] and ("natural", "timeout")[stop] == ref["stop"]
    assert_bits_equal(rows, ref["rows"], "rows from the C client")
