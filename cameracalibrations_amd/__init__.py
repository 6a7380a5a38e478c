"""MI355X-native camera-model fit: drop-in for the `fit_model` / point-map path of
yakir12/CameraCalibrations.  HIP kernels live in csrc/, the C ABI in
include/ccal.h; this package is the Python mirror of the reference's Julia interface.
Importing it needs no GPU; every compute call does (there is no CPU fallback)."""
from .calibration import (Calibration, RowCol, XYZ, board_template, calculate_errors, fit, fit_corners,  # noqa: F401
                          obj2img, rectification)
from .detect import _detect_corners, detect_corners, detect_corners_many  # noqa: F401
from .io import load, save  # noqa: F401
from .solver import CRITERIA, Solver, device_count, fit_many, fit_model, fit_models, reference_flags  # noqa: F401

__all__ = ["fit", "fit_corners", "fit_model", "fit_models", "fit_many", "Calibration", "RowCol", "XYZ", "rectification", "calculate_errors",
           "obj2img", "board_template", "save", "load", "detect_corners", "detect_corners_many", "Solver", "device_count", "reference_flags", "CRITERIA"]
