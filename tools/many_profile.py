"""For rocprofv3 --kernel-trace --stats: 32 resident fits of the reference's example shape, 30 rounds of ccal_fit_many (merged launches).
   rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o prof -- python3 tools/many_profile.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import board_template, fit_many  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver, reference_flags  # noqa: E402

ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
obj = board_template(ex["n_corners"])
img = np.stack([np.array(m, dtype=np.float32).reshape(-1, 2) for m in ex["corners"]])
rng = np.random.default_rng(7)
solvers = []
for _ in range(32):
    s = Solver(0)
    s.set_problem(obj, img.shape[0], ex["image_size_wh"], aspect=1.0, flags=reference_flags(True), max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points((img + rng.normal(0, 0.05, img.shape)).astype(np.float32))
    solvers.append(s)
for _ in range(30):
    fit_many(solvers)
print("rounds 30 contexts 32 merged", solvers[0].result()["sweep_form"] >> 16)
for s in solvers:
    s.close()
