"""Soak of ccal_fit_many: a pool of contexts with random small and medium problems (1..300 images, assorted boards / models), thousands
of rounds; between rounds some contexts get new corners, some a new problem, some run their own ccal_fit.  Every result must be
bit-identical to what a fresh context's ccal_fit gives for the same problem.   python tools/many_soak.py [--rounds 400] [--contexts 24]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import fit_many  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import CRITERIA, Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=400)
ap.add_argument("--contexts", type=int, default=24)
a = ap.parse_args()
rng = np.random.default_rng(2025)
NAMES = ["C1", "C2", "REF", "C5"]


def draw(k):
    name = NAMES[int(rng.integers(len(NAMES)))]
    n = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 12, 25, 60, 130, 300])) if name != "C5" else int(rng.choice([2, 3, 5, 9, 40]))
    if name == "C1" and n < 2:
        n = 2
    return syn.named_workload(name, n_images=n, seed_offset=int(rng.integers(1 << 20)))


def setup(s, wl):
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points(wl.img)


def alone(wl):
    with Solver(0) as s:
        setup(s, wl)
        st = s.fit()
        return st, s.result()


def same(r, r0):
    return r["iters"] == r0["iters"] and r["sweeps"] == r0["sweeps"] and all(np.array_equal(r[k], r0[k]) for k in ("intr", "rvecs", "tvecs"))


solvers = [Solver(0) for _ in range(a.contexts)]
wls = [draw(i) for i in range(a.contexts)]
want = [alone(w) for w in wls]
for s, w in zip(solvers, wls):
    setup(s, w)
bad = fits = changes = 0
for rnd in range(a.rounds):
    for i in rng.choice(a.contexts, size=3, replace=False):
        kind = int(rng.integers(3))
        if kind == 0:                               # a new problem (another shape)
            wls[i] = draw(i)
            setup(solvers[i], wls[i])
            want[i] = alone(wls[i])
        elif kind == 1:                             # new corners of the same shape
            w = syn.named_workload(wls[i].name, n_images=wls[i].n_images, seed_offset=int(rng.integers(1 << 20)))
            wls[i] = w
            solvers[i].set_image_points(w.img)
            want[i] = alone(w)
        else:                                       # its own fit in between
            solvers[i].fit()
        changes += 1
    k = int(rng.integers(2, a.contexts + 1))
    use = sorted(rng.choice(a.contexts, size=k, replace=False))
    sts = fit_many([solvers[i] for i in use])
    for i, st in zip(use, sts):
        fits += 1
        if st != want[i][0] or not same(solvers[i].result(), want[i][1]):
            bad += 1
            print(f"MISMATCH round {rnd} context {i}: {wls[i].name} N={wls[i].n_images}", flush=True)
    if rnd % 50 == 49:
        print(f"round {rnd + 1}: {fits} fits through ccal_fit_many, {changes} context changes, {bad} mismatches", flush=True)
print(f"many_soak: {a.rounds} rounds, {a.contexts} contexts, {fits} fits through ccal_fit_many, {changes} context changes in between, {bad} mismatches")
for s in solvers:
    s.close()
sys.exit(1 if bad else 0)
