"""pyitd_amd — MI355X (gfx950) engine for the ITD hot path of falseywinchnet/PyITD.

    from pyitd_amd import ITD, itd_baseline_extract, detect_peaks
    rows = ITD().itd(x, max_iteration=7)

Only the data-parallel path  itd(x, n) -> rotations, baseline  lives here: hand-written HIP kernels
behind a C ABI (include/pyitd_hip.h), and this host-side mirror of the reference's call surface.
"""
from . import _lib
from ._lib import ITDError, build
from .engine import Engine
from .itd import (ITD, baseline_knot_estimation, detect_knots, detect_peaks, find_extrema, generate_sine_wave, instantaneous, isin, itd,
                  itd_baseline_extract, itd_baseline_extract_cubic, itd_baseline_extract_fast, itd_baseline_extract_iq, itd_batch, itd_levels,
                  itd_sine_wrapper, matlab_detect_peaks, release_engines)

from .spline import (crossways_itd_baseline_extract, itd_baseline_extract_modified, itd_baseline_extract_rows,
                     itd_baseline_extract_spline, mad, retrieve_statistical_image_component, totalextract2d)

__all__ = ["itd_baseline_extract_modified", "itd_baseline_extract_spline", "itd_baseline_extract_rows", "mad",
           "crossways_itd_baseline_extract", "retrieve_statistical_image_component", "totalextract2d", "ITD", "ITDError", "Engine", "build", "itd", "itd_levels", "itd_batch", "itd_baseline_extract", "detect_peaks",
           "matlab_detect_peaks", "detect_knots", "baseline_knot_estimation", "isin", "find_extrema", "generate_sine_wave",
           "itd_baseline_extract_fast", "itd_baseline_extract_cubic", "itd_baseline_extract_iq", "itd_sine_wrapper", "release_engines", "instantaneous"]
