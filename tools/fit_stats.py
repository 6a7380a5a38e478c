"""Distribution of the one-pipeline fit's wall time (ccal_fit, corners resident) over many calls: median, p99, max -- the overlapped first LM launch
(round 5) must not produce outliers (CCAL_NO_OVERLAP=1: the serial form).   python tools/fit_stats.py [--workload C3] [--images N] [--calls 2000]"""
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
ap.add_argument("--calls", type=int, default=2000)
ap.add_argument("--tag", default="")
a = ap.parse_args()
wl = syn.named_workload(a.workload, n_images=a.images)
img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points(img)
    for _ in range(20):
        s.fit()
    ts = []
    for _ in range(a.calls):
        t0 = time.perf_counter()
        s.fit()
        ts.append(time.perf_counter() - t0)
    r = s.result()
ts = 1e6 * np.array(ts)
print(f"[fit {a.tag or ('no-overlap' if os.environ.get('CCAL_NO_OVERLAP') else 'overlap')}] {wl.name} N={wl.n_images}: median {np.median(ts):.1f} us, p90 {np.percentile(ts, 90):.1f}, "
      f"p99 {np.percentile(ts, 99):.1f}, max {ts.max():.1f}, over 200 us: {(ts > 200).sum()} of {len(ts)}; rms {r['rms']:.9f} iters {r['iters']} sweeps {r['sweeps']}")
