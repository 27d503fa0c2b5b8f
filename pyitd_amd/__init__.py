"""pyitd_amd — MI355X (gfx950) engine for the ITD hot path of falseywinchnet/PyITD.

    from pyitd_amd import ITD, itd_baseline_extract, detect_peaks
    rows = ITD().itd(x, max_iteration=7)

Only the data-parallel path  itd(x, n) -> rotations, baseline  lives here: hand-written HIP kernels
behind a C ABI (include/pyitd_hip.h), and this host-side mirror of the reference's call surface.
"""
from . import _lib
from ._lib import ITDError, build
from .engine import Engine
from .itd import (ITD, baseline_knot_estimation, detect_knots, detect_peaks, isin, itd, itd_baseline_extract,
                  itd_batch, itd_levels, matlab_detect_peaks)

__all__ = ["ITD", "ITDError", "Engine", "build", "itd", "itd_levels", "itd_batch", "itd_baseline_extract", "detect_peaks",
           "matlab_detect_peaks", "detect_knots", "baseline_knot_estimation", "isin"]
