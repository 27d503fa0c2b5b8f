"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/pyitd_hip.h declares (no compute without a GPU), and the host mirror keeps the reference's surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pyitd_amd import _lib
    _lib.build()
    return _lib.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "pyitd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(itd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = _declared_functions()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "libpyitd_hip.so does not export %s" % name
    from pyitd_amd._lib import ABI
    assert sorted(ABI) == names, "pyitd_amd/_lib.py prototypes and include/pyitd_hip.h disagree"


def test_abi_version_and_status_strings(lib):
    assert lib.itd_abi_version() == 11
    assert lib.itd_status_string(0) == b"ok"
    assert b"argument" in lib.itd_status_string(1)


def test_no_gpu_means_a_loud_failure_not_a_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert lib.itd_engine_create(ctypes.byref(h), 0, 1 << 12, 1) == 2   # ITD_ERR_NO_DEVICE
    import pyitd_amd
    with pytest.raises(pyitd_amd.ITDError):
        pyitd_amd.ITD().itd(np.sin(np.arange(100.0)))


def test_host_mirror_keeps_the_reference_surface():
    import inspect
    import pyitd_amd
    assert list(inspect.signature(pyitd_amd.ITD.__init__).parameters)[:2] == ["self", "extrema_detection"]
    assert list(inspect.signature(pyitd_amd.ITD.itd).parameters)[:3] == ["self", "data", "max_iteration"]
    assert inspect.signature(pyitd_amd.ITD.itd).parameters["max_iteration"].default == 11      # ITD.py:351
    assert inspect.signature(pyitd_amd.ITD.__call__).parameters["max_iterations"].default == 12  # ITD.py:189
    assert inspect.signature(pyitd_amd.itd).parameters["max_iteration"].default == 22           # ITD_numba.py:101
    with pytest.raises(AssertionError):
        pyitd_amd.ITD(extrema_detection="nope")
    d = pyitd_amd.ITD()
    with pytest.raises(ValueError):
        d.get_rotations()
    assert pyitd_amd.isin(np.array([1, 2, 3]), np.array([2])).tolist() == [False, True, False]


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under pyitd_amd/ may import, link or call it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pyitd_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), "%s mentions the oracle" % os.path.join(dirpath, f)


def test_shard_range_of_the_c_abi_is_the_python_one(lib):
    """itd_shard_range (for hosts without torch.distributed) and pyitd_amd.distributed.shard_range agree; bad requests are refused."""
    import ctypes
    from pyitd_amd.distributed import shard_range
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for batch, world in ((8192, 8), (1024, 8), (10, 3), (0, 4), (5, 8), (1, 1), (1000003, 7)):
        covered = 0
        for rank in range(world):
            assert lib.itd_shard_range(batch, world, rank, ctypes.byref(lo), ctypes.byref(hi)) == 0
            assert (lo.value, hi.value) == shard_range(batch, world, rank)
            assert lo.value == covered
            covered = hi.value
        assert covered == batch
    assert lib.itd_shard_range(8, 0, 0, ctypes.byref(lo), ctypes.byref(hi)) != 0
    assert lib.itd_shard_range(8, 4, 4, ctypes.byref(lo), ctypes.byref(hi)) != 0
    assert lib.itd_shard_scatter(None, None, 16, 8, 4, 1, 0, 0, None, None) != 0      # the root's array is missing


def test_meitd_entry_points_reject_bad_arguments_before_touching_a_device(lib):
    """ABI revision 8's device-row operators: a NULL engine or pointer is ITD_ERR_INVALID_ARG (1), decided before any HIP call."""
    w = (ctypes.c_double * 6)()
    c = (ctypes.c_int64 * 6)()
    k = ctypes.c_int32()
    assert lib.itd_wpe3_f64(None, None, 100, w, c, ctypes.byref(k), None) == 1
    assert lib.itd_count_knots_f64(None, None, 100, 1, 100, 0, ctypes.byref(k), None) == 1
    assert lib.itd_baseline_extract_spline2_f64(None, None, 100, 1, 100, 0, None, 100, None, 100, None, None, None) == 1
    assert lib.itd_subtract_f64(None, None, None, None, 10, None) == 1
    assert lib.itd_copy(None, None, None, 8, 0, 0, None) == 1


def test_meitd_mirror_keeps_the_reference_surface():
    """pyitd_amd.meitd's public functions take what MEITD.py's take (MEITD.py:79, :344, :371, :395, :536)."""
    import inspect
    from pyitd_amd import meitd
    assert list(inspect.signature(meitd.weighted_permutation_entropy).parameters)[:3] == ["time_series", "order", "normalize"]
    assert inspect.signature(meitd.weighted_permutation_entropy).parameters["order"].default == 3
    assert list(inspect.signature(meitd.retrieve_proper_rotation).parameters)[:2] == ["x", "WPEMAX"]
    assert list(inspect.signature(meitd.determine_if_first_is_proper_rotation).parameters)[:2] == ["x", "WPEMAX"]
    p = inspect.signature(meitd.MEITD).parameters
    assert list(p)[:3] == ["data", "max_iteration", "WPEMAX"] and p["max_iteration"].default == 40 and p["WPEMAX"].default == 0.6
    assert list(inspect.signature(meitd.XITD).parameters)[:1] == ["data"]
