"""Diagnostic: per-call latency of the corner detector on the reference's example images (host image in, corners out)."""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd.detect import detect_corners  # noqa: E402

ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
imgs = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "example_png", f)))) for f in ex["files"]]
nc = tuple(ex["n_corners"])
for im in imgs:
    detect_corners(im, nc)
ts = []
for _ in range(30):
    for im in imgs:
        t0 = time.perf_counter()
        detect_corners(im, nc)
        ts.append(1e6 * (time.perf_counter() - t0))
print("detect_corners: median %.1f us, min %.1f, p90 %.1f  (%d x %d px)" % (np.median(ts), np.min(ts), np.percentile(ts, 90), *imgs[0].shape))
