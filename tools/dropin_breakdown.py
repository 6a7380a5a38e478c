"""Where the drop-in call's time goes on the reference's example (6 images x 5x8 corners): fit_model (Python marshalling + ccal_calibrate), the
ccal_calibrate call alone on pre-marshalled arrays, and ccal_fit on resident corners.  Medians of 200."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cameracalibrations_amd as cc  # noqa: E402
from cameracalibrations_amd import _ffi as F  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver, reference_flags  # noqa: E402

ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
img = [np.array(m, dtype=np.float64) for m in ex["corners"]]
obj = cc.board_template(ex["n_corners"])


def med(fn, n=200):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


print("fit_model (lists of float64 arrays in): %.1f us" % med(lambda: cc.fit_model(ex["image_size_wh"], obj, img, ex["n_corners"], True, 1)))
arr = np.ascontiguousarray(np.stack(img), dtype=np.float32)
print("fit_model (one float32 array in):       %.1f us" % med(lambda: cc.fit_model(ex["image_size_wh"], obj, arr, ex["n_corners"], True, 1)))
P, N = 40, len(img)
o32 = np.ascontiguousarray(obj, dtype=np.float32)
K4, d5, rv, tv = np.zeros(4), np.zeros(5), np.zeros((N, 3)), np.zeros((N, 3))
rms, it, st = C.c_double(0), C.c_int32(0), F.Stats()
dp = F._dp
args = (F.f32(o32)[1], F.f32(arr)[1], P, N, 375, 500, 1.0, reference_flags(True), CRITERIA[0], CRITERIA[1], 0, K4.ctypes.data_as(dp),
        d5.ctypes.data_as(dp), rv.ctypes.data_as(dp), tv.ctypes.data_as(dp), C.byref(rms), C.byref(it), C.byref(st))
call = F.lib().ccal_calibrate
print("ccal_calibrate alone (ctypes, arrays ready): %.1f us" % med(lambda: call(*args)))
with Solver(0) as s:
    s.set_problem(obj, N, ex["image_size_wh"], aspect=1.0, flags=reference_flags(True), max_iter=30, eps=1e-3)
    s.set_image_points(arr)
    print("ccal_fit, corners resident:               %.1f us" % med(lambda: s.fit()))
    print("   ccal_set_problem + ccal_set_image_points (host arrays): %.1f us" % med(lambda: (s.set_problem(obj, N, ex["image_size_wh"], aspect=1.0, flags=reference_flags(True), max_iter=30, eps=1e-3), s.set_image_points(arr))))
    s.fit()
    print("   ccal_get_result incl. poses: %.1f us" % med(lambda: s.result()))
