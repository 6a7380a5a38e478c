"""Diagnostic: repeatability of the `maps` block of bench.py (same process, several repetitions)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver  # noqa: E402

wl = syn.named_workload("C3")
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points(wl.img)
    s.calibrate()
    rf = s.result()
for rep in range(6):
    m = bench.maps_block(wl, rf, with_cpu=False)
    print(rep, "forward %.3g inverse %.3g errors %.3g single %.1f us" % (m["forward_points_per_s"], m["inverse_points_per_s"], m["errors_corners_per_s"],
                                                                  m["single_point_call_us"]), flush=True)
    time.sleep(0.5 * rep)
