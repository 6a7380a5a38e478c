"""Soak of fit_models (= ccal_calibrate_many): calls with 1..40 calibrations of mixed sizes -- a few images staged through the pinned blocks,
hundreds uploaded the ordinary way, now and then a degenerate one (all corners in one point) that must fail alone -- every result compared
with fit_model on the same set (same values, bit for bit).   python tools/models_soak.py [--rounds 150]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import _ffi as F  # noqa: E402
from cameracalibrations_amd import fit_model, fit_models  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=150)
a = ap.parse_args()
rng = np.random.default_rng(77)
wl = syn.named_workload("REF", n_images=700)
size, nc = wl.image_size, wl.n_corners
want = {}


def ref(key, pts):
    if key not in want:
        m = fit_model(size, wl.obj, pts, nc, True, 1.0)
        want[key] = (m.frow, m.fcol, m.crow, m.ccol, m.k, m.rms, m.iters, np.array(m.Rs), np.array(m.ts))
    return want[key]


bad = calls = cals = failed_alone = 0
for rnd in range(a.rounds):
    k = int(rng.integers(1, 41))
    keys, sets = [], []
    for _ in range(k):
        n = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 30, 50, 120, 300, 700], p=[.08, .08, .1, .1, .1, .1, .1, .1, .08, .06, .05, .03, .02]))
        lo = int(rng.integers(0, 700 - n + 1))
        keys.append((lo, n))
        sets.append(wl.img[lo:lo + n])
    degenerate = rng.random() < 0.15
    if degenerate:
        j = int(rng.integers(k))
        sets[j] = np.full_like(sets[j], 100.0)
        keys[j] = None
    calls += 1
    try:
        out = fit_models(size, wl.obj, sets, nc, True, 1.0)
        if degenerate:
            bad += 1
            print(f"MISSING FAILURE round {rnd}")
            continue
    except F.CcalError as e:
        if not degenerate:
            bad += 1
            print(f"UNEXPECTED FAILURE round {rnd}: {e}")
            continue
        failed_alone += 1
        # the others of that call: again without the bad one
        keep = [i for i in range(k) if keys[i] is not None]
        keys = [keys[i] for i in keep]
        sets = [sets[i] for i in keep]
        out = fit_models(size, wl.obj, sets, nc, True, 1.0) if sets else []
    for key, pts, m in zip(keys, sets, out):
        w = ref(key, pts)
        cals += 1
        if (m.frow, m.fcol, m.crow, m.ccol, m.k, m.rms, m.iters) != w[:7] or not np.array_equal(np.array(m.Rs), w[7]) or not np.array_equal(np.array(m.ts), w[8]):
            bad += 1
            print(f"MISMATCH round {rnd} set {key}", flush=True)
    if rnd % 25 == 24:
        print(f"round {rnd + 1}: {calls} calls, {cals} calibrations compared, {failed_alone} calls with a degenerate calibration, {bad} mismatches", flush=True)
print(f"models_soak: {calls} fit_models calls, {cals} calibrations compared with fit_model, {failed_alone} calls with a degenerate calibration reported, {bad} mismatches")
sys.exit(1 if bad else 0)
