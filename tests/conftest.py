import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f != "radio8000_input.npz" and not f.startswith("helpers_"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The in-tree HIP library and the oracle normally arrive prebuilt (they travel with the snapshot); build them
    if they are missing or older than their sources (hipcc cross-compiles gfx950 without a GPU)."""
    from pyitd_amd import _lib
    _lib.build()
    from oracle import cpu_oracle
    cpu_oracle.build()
    # torch first: on this image torch's own HIP runtime refuses to start ("No HIP GPUs are available") once libpyitd_hip.so has
    # initialised the system one, whereas the other order works — and some GPU tests hold their device buffers in torch tensors
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
