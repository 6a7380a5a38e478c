"""ctypes binding of the C ABI in include/ccal.h (libccal_hip.so).

This mirrors 1:1 what the Julia wrapper (julia/CameraCalibrationsAMD.jl) does with `ccall`.
There is no fallback: if the HIP library is missing or no gfx950 device is visible, every
compute call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCAL_LIB") or os.path.join(_HERE, "libccal_hip.so")   # CCAL_LIB: variant builds (tools/)
NINTR = 9

# status codes (include/ccal.h)
OK, W_MAX_ITER, W_RANK_DEFICIENT = 0, 1, 2      # (the two warnings are bits: 3 = both)
E_ARG, E_NONPLANAR, E_DEGENERATE, E_NONFINITE, E_DEVICE, E_COMM, E_STATE = -1, -2, -3, -4, -5, -6, -7

# flag word (OpenCV values, reference src/detect_fit.jl:40)
USE_INTRINSIC_GUESS = 0x1
FIX_ASPECT_RATIO = 0x2
FIX_PRINCIPAL_POINT = 0x4
ZERO_TANGENT_DIST = 0x8
FIX_FOCAL_LENGTH = 0x10
FIX_K1 = 0x20
FIX_K2 = 0x40
FIX_K3 = 0x80
FIX_TANGENT_DIST = 0x200000


class CcalError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ccal status {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("rms", C.c_double), ("final_err", C.c_double), ("rel_step", C.c_double),
                ("t_init_intr_s", C.c_double), ("t_init_extr_s", C.c_double), ("t_lm_s", C.c_double),
                ("iters", C.c_int32), ("sweeps", C.c_int32), ("reschur", C.c_int32), ("rejected", C.c_int32),
                ("lambda_lg10", C.c_int32), ("n_bad", C.c_int32), ("timed_sweeps", C.c_int32), ("sweep_form", C.c_int32),
                ("sweep_us", C.c_double), ("solve_kernels_us", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Trace(C.Structure):
    _fields_ = [("err", C.c_double), ("rel_step", C.c_double), ("accepted", C.c_int32),
                ("lambda_lg10", C.c_int32), ("iters", C.c_int32), ("pad", C.c_int32)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); also the list of symbols the CPU test checks for
SIGNATURES = {
    "ccal_version": (C.c_int, []),
    "ccal_last_error": (C.c_char_p, []),
    "ccal_device_count": (C.c_int, []),
    "ccal_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "ccal_destroy": (None, [C.c_void_p]),
    "ccal_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccal_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p, C.c_int32, C.c_int32, C.c_int64]),
    "ccal_comm_unique_id": (C.c_int, [C.c_char_p]),
    "ccal_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]),
    "ccal_p2p_export": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    "ccal_p2p_connect": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "ccal_p2p_selftest": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccal_p2p_set_exchange": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccal_p2p_export2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]),
    "ccal_p2p_connect2": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "ccal_p2p_shared_device": (C.c_int, [C.c_void_p]),
    "ccal_device_uuid": (C.c_int, [C.c_int32, C.c_char_p]),
    "ccal_set_problem": (C.c_int, [C.c_void_p, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                   C.c_uint32, C.c_int32, C.c_double]),
    "ccal_set_image_points": (C.c_int, [C.c_void_p, _fp]),
    "ccal_set_image_points_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ccal_init_intrinsics": (C.c_int, [C.c_void_p]),
    "ccal_set_intrinsics": (C.c_int, [C.c_void_p, _dp]),
    "ccal_init_extrinsics": (C.c_int, [C.c_void_p]),
    "ccal_set_extrinsics": (C.c_int, [C.c_void_p, _dp]),
    "ccal_lm_run": (C.c_int, [C.c_void_p]),
    "ccal_fit": (C.c_int, [C.c_void_p]),
    "ccal_fit_many": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_int32)]),
    "ccal_calibrate_many": (C.c_int, [_fp, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_double,
                                      C.c_uint32, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "ccal_set_sweep_timing": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccal_set_graph": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccal_set_pose_init": (C.c_int, [C.c_void_p, C.c_int32]),
    "ccal_get_result": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.POINTER(Stats)]),
    "ccal_get_trace": (C.c_int, [C.c_void_p, C.POINTER(Trace), C.c_int32, C.POINTER(C.c_int32)]),
    "ccal_calibrate": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_uint32,
                                 C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int32),
                                 C.POINTER(Stats)]),
    "ccal_calibrate_multi": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_uint32,
                                       C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_int32), _dp, _dp, _dp, _dp, _dp,
                                       C.POINTER(C.c_int32), C.POINTER(Stats)]),
    "ccal_thread_cleanup": (None, []),
    "ccal_project": (C.c_int, [C.c_void_p, _dp, C.c_int32, _dp, _dp, C.c_double, _dp, _dp, C.c_double, _dp]),
    "ccal_unproject": (C.c_int, [C.c_void_p, _dp, C.c_int32, _dp, _dp, C.c_double, _dp, _dp, C.c_double, _dp]),
    "ccal_project_many": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, C.c_double, _dp, _dp, C.c_double, _dp]),
    "ccal_unproject_many": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, _dp, _dp, C.c_double, _dp, _dp, C.c_double, _dp]),
    "ccal_errors": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp,
                              C.c_double, C.c_int32, _dp, _dp]),
    "ccal_rectify_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                     _dp, _dp, C.c_double, _dp, _dp, C.c_double, C.c_double, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_double, C.c_void_p]),
    "ccal_detect_corners": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_int32, _dp, C.POINTER(C.c_int32), _dp]),
    "ccal_detect_corners_many": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
                                           C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "ccal_debug_image_grams": (C.c_int, [C.c_void_p, _dp, _dp, _dp]),
    "ccal_debug_acc_sum": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32, _dp]),
    "ccal_debug_get_homographies": (C.c_int, [C.c_void_p, _dp]),
    "ccal_debug_p2p_world": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint32, _dp, _dp]),
    "ccal_debug_multi_set_epoch": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_uint32]),
    "ccal_get_init": (C.c_int, [C.c_void_p, _dp, _dp]),
    "ccal_time_sweep": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
}

_LIB = None


def lib():
    """Load libccal_hip.so (built by cameracalibrations_amd.build).  Raises if it is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CcalError(E_DEVICE, f"{LIB_PATH} not built: run `python -m cameracalibrations_amd.build` "
                                      "(there is no CPU fallback)")
        try:
            import torch  # noqa: F401  -- bind to torch's HIP runtime when it is around (see build.py)
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


_RAW = {}


def raw(name):
    """The same entry point with every pointer argument declared `void *`: callers pass plain integer addresses (numpy's
    `arr.ctypes.data`) -- a few microseconds less per call than building typed ctypes pointers, which is what a drop-in call of a
    hundred microseconds is made of.  Same symbol, same argument order and widths as SIGNATURES[name]."""
    fn = _RAW.get(name)
    if fn is None:
        res, args = SIGNATURES[name]
        fn = lib()[name]                                     # (a new function object: its prototype is its own)
        fn.restype = res
        fn.argtypes = [a if a in (C.c_int, C.c_int32, C.c_uint32, C.c_int64, C.c_double) else C.c_void_p for a in args]
        _RAW[name] = fn
    return fn


def check(rc):
    if rc < 0:
        raise CcalError(rc, lib().ccal_last_error().decode("utf-8", "replace"))
    return rc


def f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_fp)


def f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def out64(*shape):
    a = np.zeros(shape, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)
