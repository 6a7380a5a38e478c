"""Median wall time of the two initialisation stages (host call to completion) on a workload; CCAL_LIB selects a library variant.
   python tools/init_timing.py [--workload C3] [--images N] [--tag name]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--images", type=int, default=None)
ap.add_argument("--tag", default="")
a = ap.parse_args()
wl = syn.named_workload(a.workload, n_images=a.images)
img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points(img)
    ti, te, tf = [], [], []
    for rep in range(60):
        t0 = time.perf_counter()
        s.init_intrinsics()
        t1 = time.perf_counter()
        s.init_extrinsics()
        t2 = time.perf_counter()
        ti.append(t1 - t0)
        te.append(t2 - t1)
    for rep in range(30):
        t0 = time.perf_counter()
        s.fit()
        tf.append(time.perf_counter() - t0)
    r = s.result()
print(f"[init {a.tag or os.environ.get('CCAL_LIB', 'default')}] {wl.name} N={wl.n_images}: intrinsics {1e6 * np.median(ti[10:]):.1f} us, "
      f"extrinsics {1e6 * np.median(te[10:]):.1f} us, one-pipeline fit {1e6 * np.median(tf[5:]):.1f} us, rms {r['rms']:.9f} iters {r['iters']}")
