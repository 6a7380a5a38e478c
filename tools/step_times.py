"""Diagnostic: duration of consecutive lm_run() calls right after set-up (does the first handful run slower?)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver  # noqa: E402

wl = syn.named_workload("C3")
img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
s = Solver(0)
s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
s.set_image_points(img)
s.init_intrinsics()
s.init_extrinsics()
ts = []
for _ in range(300):
    t0 = time.perf_counter()
    s.lm_run()
    ts.append(1e6 * (time.perf_counter() - t0))
ts = np.array(ts)
print("first 12:", [int(t) for t in ts[:12]])
for a, b in ((0, 5), (5, 25), (25, 50), (50, 100), (100, 200), (200, 300)):
    print("steps %3d..%3d: mean %.1f us median %.1f" % (a, b, ts[a:b].mean(), np.median(ts[a:b])))
time.sleep(0.5)
ts2 = []
for _ in range(40):
    t0 = time.perf_counter()
    s.lm_run()
    ts2.append(1e6 * (time.perf_counter() - t0))
print("after 0.5 s idle, first 12:", [int(t) for t in ts2[:12]], "mean of 5..25: %.1f" % np.mean(ts2[5:25]))
