"""fit_model (host arrays in and out) on small fits: the reference's example (6 x 40) and synthetic reference-flag sets of N images x 54
corners; median microseconds per call.  CCAL_LIB selects a library variant.   python tools/small_dropin_timing.py [--tag name]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cameracalibrations_amd as cc  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="")
a = ap.parse_args()


def med(fn, n=200):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))


ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
img = [np.array(m, dtype=np.float64) for m in ex["corners"]]
obj = cc.board_template(ex["n_corners"])
out = [f"example 6x40 {med(lambda: cc.fit_model(ex['image_size_wh'], obj, img, ex['n_corners'], True, 1)):.1f}"]
for k in (5, 4, 3):
    out.append(f"example {k}x40 {med(lambda: cc.fit_model(ex['image_size_wh'], obj, img[:k], ex['n_corners'], True, 1)):.1f}")
for n in (4, 5, 6, 7, 8, 12):
    wl = syn.named_workload("REF", n_images=n)
    pts = [wl.img[i] for i in range(n)]
    out.append(f"REF {n}x54 {med(lambda: cc.fit_model(wl.image_size, wl.obj, pts, wl.n_corners, True, 1)):.1f}")
print(f"[dropin {a.tag}] " + " | ".join(out))
