"""ctypes loader for libpyitd_hip.so (the C ABI of include/pyitd_hip.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is present the product
path raises.  `build()` compiles the library in-tree with hipcc for gfx950.
"""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PYITD_HIP_LIB") or os.path.join(_HERE, "libpyitd_hip.so")  # env override: diagnostic builds
SOURCES = [os.path.join(_HERE, "csrc", "itd_engine.hip")]
# everything the library is compiled from (a header that is not listed here would not trigger a rebuild: list by pattern)
HEADERS = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hpp")) + glob.glob(os.path.join(_HERE, "csrc", "*.inc"))) + \
    [os.path.join(os.path.dirname(_HERE), "include", "pyitd_hip.h")]

MAX_ROWS = 22
MAX_ITERATION = 20
ABI_VERSION = 11

# name -> (restype, argtypes); mirrors include/pyitd_hip.h one to one
_P, _I64, _I32, _INT = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int
ABI = {
    "itd_abi_version": (_INT, []),
    "itd_status_string": (ctypes.c_char_p, [_INT]),
    "itd_last_error": (ctypes.c_char_p, [_P]),
    "itd_engine_create": (_INT, [ctypes.POINTER(_P), _INT, _I64, _I32]),
    "itd_engine_destroy": (None, [_P]),
    "itd_engine_workspace_bytes": (_I64, [_P]),
    "itd_engine_device": (_INT, [_P]),
    "itd_shard_range": (_INT, [_I64, _I32, _I32, _P, _P]),
    "itd_shard_scatter": (_INT, [_P, _P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _P]),
    "itd_decompose_f32": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _P, _P]),
    "itd_decompose_f64": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _P, _P]),
    "itd_get_summary": (_INT, [_P, _P, _P, _P, _P, _P]),
    "itd_decompose_host_f64": (_INT, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "itd_decompose_host_f32": (_INT, [_P, _P, _I64, _I32, _P, _P, _P, _P, _P, _P]),
    "itd_baseline_extract_f64": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    "itd_baseline_extract_f32": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    "itd_baseline_extract_host_f64": (_INT, [_P, _P, _I64, _P, _P, _P, _P, _P]),
    "itd_detect_f64": (_INT, [_P, _P, _I64, _I32, _P, _P, _P]),
    "itd_detect_f32": (_INT, [_P, _P, _I64, _I32, _P, _P, _P]),
    "itd_detect_host_f64": (_INT, [_P, _P, _I64, _I32, _P, _P]),
    "itd_knot_values_host_f64": (_INT, [_P, _P, _I64, _P, _I64, _P]),
    "itd_knot_values_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P]),
    "itd_baseline_extract_cubic_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_baseline_extract_cubic_f32": (_INT, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_baseline_extract_cubic_host_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_baseline_extract_iq_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_baseline_extract_iq_host_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_find_extrema_host_f64": (_INT, [_P, _P, _I64, _P, _P]),
    "itd_baseline_extract_spline_f64": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _I64, _P, _I64, _P, _P]),
    "itd_baseline_extract_spline_host_f64": (_INT, [_P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "itd_baseline_extract_spline_host2_f64": (_INT, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _P]),
    "itd_set_spline_solver": (_INT, [_P, _I32]),
    "itd_count_knots_host_f64": (_INT, [_P, _P, _I64, _I32, _I32, _P]),
    "itd_count_knots_f64": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _P]),
    "itd_wpe3_f64": (_INT, [_P, _P, _I64, _P, _P, _P, _P]),
    "itd_wpe_f64": (_INT, [_P, _P, _I64, _I32, _P, _P, _P]),
    "itd_baseline_extract_spline2_f64": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _I64, _P, _I64, _P, _P, _P]),
    "itd_subtract_f64": (_INT, [_P, _P, _P, _P, _I64, _P]),
    "itd_copy": (_INT, [_P, _P, _P, _I64, _I32, _I32, _P]),
    "itd_meitd_small_f64": (_INT, [_P, _P, _I64, ctypes.c_double, _P, _P, _I32, _P]),
    "itd_crossways_f64": (_INT, [_P, _P, _I32, _I32, _I32, _I32, _P, _P]),
    "itd_crossways_host_f64": (_INT, [_P, _P, _I32, _I32, _I32, _I32, _P]),
    "itd_instantaneous_f64": (_INT, [_P, _P, _I64, _P, _P, _P, _P]),
    "itd_instantaneous_host_f64": (_INT, [_P, _P, _I64, _P, _P, _P]),
    "itd_baseline_extract_batch_f64": (_INT, [_P, _P, _I64, _I32, _I64, _P, _I64, _P, _I64, _P, _P]),
    "itd_detect_batch_f64": (_INT, [_P, _P, _I64, _I32, _I64, _I32, _P, _I64, _P, _P]),
    "itd_baseline_extract_cubic_batch_f64": (_INT, [_P, _P, _I64, _I32, _I64, _P, _I64, _I64, _P, _I64, _P, _P]),
    "itd_stream_create": (_INT, [ctypes.POINTER(_P), _INT, _I64, _I32, _I32, _I32, _I32]),
    "itd_stream_destroy": (None, [_P]),
    "itd_stream_reset": (_INT, [_P]),
    "itd_stream_blocks": (_I64, [_P]),
    "itd_stream_last_error": (ctypes.c_char_p, [_P]),
    "itd_stream_push_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _I64, _P, _P]),
    "itd_stream_flush_f64": (_INT, [_P, _P, _I64, _P, _I64, _P, _P]),
    "itd_stream_push_host_f64": (_INT, [_P, _P, _P, _P, _P]),
    "itd_stream_flush_host_f64": (_INT, [_P, _P, _P, _P]),
    "itd_stream_status": (_INT, [_P, _P]),
    "itd_set_nan_input_mode": (_INT, [_P, _I32]),
    "itd_set_batch_chunk": (_INT, [_P, _I32]),
    "itd_set_batch_streams": (_INT, [_P, _I32]),
    "itd_set_batch_pipeline": (_INT, [_P, _I32]),
    "itd_set_level0_mode": (_INT, [_P, _I32]),
    "itd_set_host_keep_baselines": (_INT, [_P, _I32]),
    "itd_get_last_baselines_host": (_INT, [_P, _P, _I64, _I32]),
    "itd_set_valid_flags": (_INT, [_P, _P]),
    "itd_set_device_repair": (_INT, [_P, _I32]),
    "itd_get_device_repairs": (_I64, [_P]),
    "itd_set_kernel_timing_mode": (_INT, [_P, _I32]),
    "itd_get_kernel_timing_samples": (_INT, [_P, _I32, _P, _I32, _P]),
    "itd_get_step_periods": (_INT, [_P, _P, _I32, _P]),
    "itd_set_fuse_range": (_INT, [_P, _I32]),
    "itd_set_fuse_mode": (_INT, [_P, _I32]),
    "itd_set_fuse_level": (_INT, [_P, _I32]),
    "itd_set_fuse_min_samples": (_INT, [_P, _I64]),
    "itd_set_fuse_group": (_INT, [_P, _I32]),
    "itd_set_fuse_cap": (_INT, [_P, _I32]),
    "itd_get_last_fuse_cap": (_INT, [_P]),
    "itd_debug_kf_fault": (_INT, [_P, _I32, _I32, _I32, _I32, _I32]),
    "itd_debug_kf_fault_signal": (_INT, [_P, _I32]),
    "itd_debug_int_ratio_check": (_INT, [_INT, _I32, ctypes.POINTER(_I64)]),
    "itd_get_fuse_repeats": (_INT, [_P]),
    "itd_get_last_fuse_level": (_INT, [_P]),
    "itd_get_fuse_signal_repairs": (_I64, [_P]),
    "itd_set_resident_mode": (_INT, [_P, _I32]),
    "itd_get_resident_repeats": (_INT, [_P]),
    "itd_set_resident_window": (_INT, [_P, _I32]),
    "itd_dev_alloc": (_INT, [_INT, _I64, ctypes.POINTER(_P)]),
    "itd_dev_free": (_INT, [_INT, _P]),
    "itd_dev_copy": (_INT, [_INT, _P, _P, _I64, _I32]),
    "itd_set_kernel_timing": (_INT, [_P, _INT]),
    "itd_set_kernel_timing_stride": (_INT, [_P, _INT]),
    "itd_get_kernel_timing": (_INT, [_P, _I32, _P, _P]),
}

_lib = None


def hipcc_command(out=LIB_PATH, tile=None):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-Wall", "-o", out]
    if tile:
        cmd.append("-DITD_TILE=%d" % tile)
    return cmd + SOURCES + ["-ldl"]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force=False):
    """Compile libpyitd_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force or needs_build():
        subprocess.check_call(hipcc_command())
    return LIB_PATH


def load():
    """dlopen the HIP library and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "pyitd_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        if L.itd_abi_version() != ABI_VERSION:
            raise RuntimeError("pyitd_amd: ABI version mismatch")
        _lib = L
    return _lib


class ITDError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = load().itd_status_string(status).decode()
        super().__init__("pyitd_hip status %d (%s)%s" % (status, msg, (": " + detail) if detail else ""))
