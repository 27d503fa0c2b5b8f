"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool: CPU build only)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import glob, os, sys
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
from oracle import cpu_oracle
cpu_oracle._LIB_PATH = os.path.join(%(root)r, "oracle", "libitd_oracle_asan.so")
cpu_oracle.build = lambda force=False: cpu_oracle._LIB_PATH
from helpers import load_golden, sha
n = 0
for f in sorted(glob.glob(os.path.join(%(root)r, "tests", "golden", "*.npz"))):
    name = os.path.basename(f)[:-4]
    if name == "radio8000_input" or name.startswith("helpers_"):
        continue
    g = load_golden(name)
    r = cpu_oracle.itd(g["x"], int(g["max_iteration"]))
    assert sha(r["rows"]) == str(g["rows_sha256"]), name
    cpu_oracle.itd_lean(g["x"], int(g["max_iteration"]), want_knots=True)
    n += 1
print("clean", n)
'''


def test_oracle_is_clean_under_asan_and_ubsan():
    def libpath(name):
        p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
        return p if os.path.isabs(p) and os.path.exists(p) else None
    asan, ubsan = libpath("libasan.so"), libpath("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("libasan/libubsan not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=asan + " " + ubsan, ASAN_OPTIONS="detect_leaks=0")
    try:
        out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "clean 42" in out.stdout, out.stdout + out.stderr[-500:]
        assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]
    finally:
        p = os.path.join(ROOT, "oracle", "libitd_oracle_asan.so")
        if os.path.exists(p):
            os.remove(p)
