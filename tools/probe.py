"""Quick timing probe of one library build (CCAL_LIB selects a variant): C3 (or --workload) LM solve time, in-solve mean launch
duration of k_lm_trial (HIP events bound to the dispatches) and back-to-back first-trial launches."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C3")
ap.add_argument("--images", type=int, default=None)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--tag", default="")
ap.add_argument("--b2b-only", action="store_true", help="back-to-back first-trial launches only (timing-experiment builds whose solves do not converge)")
a = ap.parse_args()
wl = syn.named_workload(a.workload, n_images=a.images)
img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
s = Solver(0)
s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
s.set_image_points(img)
s.init_intrinsics()
s.init_extrinsics()
if a.b2b_only:
    s.time_sweep(50)
    print(f"[probe {a.tag}] {wl.name} N={wl.n_images}: b2b first-trial {s.time_sweep(200):.2f} us {s.time_sweep(200):.2f} us")
    sys.exit(0)
for _ in range(20):
    s.lm_run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    s.lm_run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
r = s.result()
s.set_sweep_timing(True)
us = n = 0
for _ in range(50):
    s.lm_run()
    rr = s.result()
    us += rr["sweep_us"] * rr["timed_sweeps"]
    n += rr["timed_sweeps"]
s.set_sweep_timing(False)
b2b = s.time_sweep(200)
print(f"[probe {a.tag or os.environ.get('CCAL_LIB', 'default')} R={os.environ.get('CCAL_ACC_REPLICAS', '-')}] {wl.name} N={wl.n_images}: "
      f"{1e6 * dt:.2f} us/solve, in-solve launch {us / max(n, 1):.2f} us, b2b first-trial {b2b:.2f} us, "
      f"iters {r['iters']} sweeps {r['sweeps']}+{r['reschur']} rms {r['rms']:.9f}")
