"""Corner detection of a calibration's images: image-by-image calls against one ccal_detect_corners_many (three in flight); median
microseconds per image on the reference's example PNGs (375 x 500) and on rendered perspective boards (~0.5 Mpx)."""
import json
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.detect import detect_corners, detect_corners_many  # noqa: E402

ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
imgs = [np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "example_png", f)))) for f in ex["files"]]
nc = tuple(ex["n_corners"])
sets = {"example PNGs 375x500 (6)": (imgs, nc), "example PNGs x4 (24)": (imgs * 4, nc)}
big = [syn.generate_checkerboard((13, 14), 97)[1].astype(np.uint8)] * 12      # (the reference's test generator, test/runtests.jl:6-18)
sets[f"synthetic board {big[0].shape[0]}x{big[0].shape[1]} (12)"] = (big, (13, 14))
for name, (batch, n_c) in sets.items():
    for _ in range(3):
        detect_corners_many(batch, n_c)
        [detect_corners(im, n_c) for im in batch]
    t_one, t_many = [], []
    for _ in range(20):
        t0 = time.perf_counter()
        a = [detect_corners(im, n_c) for im in batch]
        t1 = time.perf_counter()
        b = detect_corners_many(batch, n_c)
        t2 = time.perf_counter()
        t_one.append((t1 - t0) / len(batch))
        t_many.append((t2 - t1) / len(batch))
    same = all((x is None) == (y is None) and (x is None or np.array_equal(x, y)) for x, y in zip(a, b))
    print(f"[detect] {name}: image by image {1e6 * np.median(t_one):.1f} us per image | three in flight {1e6 * np.median(t_many):.1f} us per image "
          f"({np.median(t_one) / np.median(t_many):.2f}x), identical {same}", flush=True)
