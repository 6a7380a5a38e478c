"""Latency of the drop-in entry point (ccal_calibrate through fit_model / fit_corners) on the
reference's own example size (6 images x 40 corners) and on C3, host buffers in and out."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cameracalibrations_amd as cc  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402

ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
img = [np.array(m, dtype=np.float64) for m in ex["corners"]]
obj = cc.board_template(ex["n_corners"])


def t(fn, n):
    """median over n calls (a mean is at the mercy of the rare tens-of-milliseconds host stall of these boxes)"""
    fn()
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)), r


ms, m = t(lambda: cc.fit_model(ex["image_size_wh"], obj, img, ex["n_corners"], True, 1), 50)
print(f"fit_model, reference example (6 x 40): {ms:.3f} ms per call   stats {m.stats}")
ms, (c, eps) = t(lambda: cc.fit_corners(img, ex["n_corners"], 1, ex["image_size_wh"], rng=np.random.default_rng(1)), 20)
print(f"fit_corners (fit + 4 error metrics), reference example: {ms:.3f} ms per call   eps {eps}")
wl = syn.named_workload("C3")
imgs = [wl.img[i] for i in range(wl.n_images)]
ms, m = t(lambda: cc.fit_model(wl.image_size, wl.obj, imgs, wl.n_corners, True, 1), 10)
print(f"fit_model, 1000 x 54 (reference flags), host arrays in/out: {ms:.3f} ms per call   stats {m.stats}")
arr = np.ascontiguousarray(wl.img, dtype=np.float32)          # one [N,P,2] float32 array: no per-image marshalling on the host
ms, m = t(lambda: cc.fit_model(wl.image_size, wl.obj, arr, wl.n_corners, True, 1), 30)
print(f"fit_model, 1000 x 54 (reference flags), one float32 array in, host arrays out: {ms:.3f} ms per call")
