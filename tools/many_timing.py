"""Throughput of independent fits on one GPU: K contexts fitted one after the other (ccal_fit each) against ccal_fit_many (every
pipeline enqueued before any wait).  Per workload and K: microseconds per fit, median over rounds.
   python tools/many_timing.py [--contexts 1,2,4,8,16,32]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from cameracalibrations_amd import board_template, fit_many, fit_model, fit_models  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver, reference_flags  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--contexts", default="1,2,4,8,16,32")
ap.add_argument("--rounds", type=int, default=40)
a = ap.parse_args()
ks = [int(v) for v in a.contexts.split(",")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def problems(name, k):
    """k different problems of one shape"""
    if name == "ref_example":
        with open(os.path.join(ROOT, "tests", "golden", "example_corners.json")) as f:
            ex = json.load(f)
        obj = board_template(ex["n_corners"])
        img = np.stack([np.array(m, dtype=np.float32).reshape(-1, 2) for m in ex["corners"]])
        rng = np.random.default_rng(7)
        return [(obj, (img + rng.normal(0, 0.05, img.shape)).astype(np.float32), ex["image_size_wh"], 1.0, reference_flags(True)) for _ in range(k)]
    n = {"C1": 100, "C3": 1000, "C1x25": 25}[name]
    out = []
    for i in range(k):
        wl = syn.named_workload("C1" if name.startswith("C1") else "C3", n_images=n, seed_offset=i)
        out.append((wl.obj, wl.img, wl.image_size, wl.aspect, wl.flags))
    return out


for name in ("ref_example", "C1x25", "C1", "C3"):
    kmax = max(ks)
    probs = problems(name, kmax)
    solvers = []
    for obj, img, size, aspect, flags in probs:
        s = Solver(0)
        s.set_problem(obj, img.shape[0], size, aspect=aspect, flags=flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
        s.set_image_points(torch.from_numpy(np.ascontiguousarray(img)).to("cuda:0"))
        solvers.append(s)
    for k in ks:
        use = solvers[:k]
        # each kind of call in its steady state; then the many-call right after the contexts' own fits (every context whose last
        # launches ran on its own stream is joined to the merged launches' stream once: an event record + wait each)
        t_seq, t_many, t_after = [], [], []
        for rep in range(a.rounds):
            t0 = time.perf_counter()
            for s in use:
                s.fit()
            t_seq.append((time.perf_counter() - t0) / k)
        for rep in range(a.rounds):
            t0 = time.perf_counter()
            fit_many(use)
            t_many.append((time.perf_counter() - t0) / k)
        for rep in range(12):
            for s in use:
                s.fit()
            t0 = time.perf_counter()
            fit_many(use)
            t_after.append((time.perf_counter() - t0) / k)
        print(f"[many] {name:12s} contexts {k:3d}: one at a time {1e6 * np.median(t_seq[5:]):8.1f} us per fit | ccal_fit_many {1e6 * np.median(t_many[5:]):8.1f} us per fit "
              f"({np.median(t_seq[5:]) / np.median(t_many[5:]):.2f}x) | right after the contexts' own fits {1e6 * np.median(t_after[2:]):8.1f}", flush=True)
    for s in solvers:
        s.close()

# the same through the drop-in seam, host arrays in and out: fit_model per calibration against one fit_models call
with open(os.path.join(ROOT, "tests", "golden", "example_corners.json")) as f:
    ex = json.load(f)
obj = board_template(ex["n_corners"])
rng = np.random.default_rng(11)
base = np.stack([np.array(m, dtype=np.float64).reshape(-1, 2) for m in ex["corners"]])
for k in ks:
    sets = [list(base + rng.normal(0, 0.05, base.shape)) for _ in range(k)]
    t_seq, t_many = [], []
    for rep in range(a.rounds):
        t0 = time.perf_counter()
        for pts in sets:
            fit_model(ex["image_size_wh"], obj, pts, ex["n_corners"], True, 1)
        t1 = time.perf_counter()
        fit_models(ex["image_size_wh"], obj, sets, ex["n_corners"], True, 1)
        t2 = time.perf_counter()
        t_seq.append((t1 - t0) / k)
        t_many.append((t2 - t1) / k)
    arr = np.stack([np.stack(st) for st in sets])
    t_arr = []
    for rep in range(a.rounds):
        t0 = time.perf_counter()
        fit_models(ex["image_size_wh"], obj, arr, ex["n_corners"], True, 1)
        t_arr.append((time.perf_counter() - t0) / k)
    print(f"[many] fit_models, one array in  calibrations {k:3d}: {1e6 * np.median(t_arr[5:]):8.1f} us per calibration", flush=True)
    print(f"[many] fit_model(s)  calibrations {k:3d}: fit_model each {1e6 * np.median(t_seq[5:]):8.1f} us per calibration | one fit_models {1e6 * np.median(t_many[5:]):8.1f} us per calibration "
          f"({np.median(t_seq[5:]) / np.median(t_many[5:]):.2f}x)", flush=True)
