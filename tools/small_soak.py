"""Soak of the round-5 small-problem paths: (1) fit_model on the reference's example, thousands of calls through the small-call path (inputs
staged in the pinned block, poses from the launch that ends the solve): every result bit-identical to the first; (2) random small problems
(1..8 images, assorted boards / models / flag words): the one-launch solve (<= 4 images) and the split one-pass sweep against
CCAL_NO_SOLO=1 CCAL_WPI=1 -- same iteration count, RMS to 1e-12 -- and bit-identical reruns.   python tools/small_soak.py [--calls 5000] [--cases 300]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cameracalibrations_amd as cc  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=5000)
ap.add_argument("--cases", type=int, default=300)
a = ap.parse_args()
ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
img = [np.array(m, dtype=np.float64) for m in ex["corners"]]
obj = cc.board_template(ex["n_corners"])
first = None
for k in range(a.calls):
    m = cc.fit_model(ex["image_size_wh"], obj, img[: 1 + k % 6] if k % 7 == 3 else img, ex["n_corners"], True, 1)
    if k % 7 == 3:
        continue                                   # (a call of another size in between: the cached context re-sizes)
    sig = (m.frow, m.k, m.rms, m.iters, tuple(np.concatenate(m.Rs + m.ts)))
    if first is None:
        first = sig
    assert sig == first, (k, sig[:4], first[:4])
print(f"fit_model x {a.calls}: bit-identical (rms {first[2]:.12f}, {first[3]} iterations)")

rng = np.random.default_rng(5)
FLAGS = [0, syn.FIX_K3, syn.FIX_K2 | syn.FIX_K3 | syn.ZERO_TANGENT_DIST, syn.FIX_ASPECT_RATIO | syn.ZERO_TANGENT_DIST | syn.FIX_K2 | syn.FIX_K3]
bad = 0
for case in range(a.cases):
    n1, n2 = int(rng.integers(4, 12)), int(rng.integers(3, 9))
    N = int(rng.integers(1, 9))
    flags = int(FLAGS[int(rng.integers(0, len(FLAGS)))])
    model = "ref" if (flags & syn.FIX_ASPECT_RATIO) else str(rng.choice(["pinhole", "k1k2p1p2"]))
    try:
        wl = syn.make_workload("soak", (n1, n2), N, model, seed=int(rng.integers(1, 1 << 30)), sigma=0.1, depth=(8.0, 16.0))
    except Exception:  # noqa: BLE001
        continue
    out = []
    for env in ({}, {}, {"CCAL_NO_SOLO": "1", "CCAL_WPI": "1"}):
        for k_ in ("CCAL_NO_SOLO", "CCAL_WPI"):
            os.environ.pop(k_, None)
        os.environ.update(env)
        with Solver(0) as s:
            s.set_problem(wl.obj, N, wl.image_size, aspect=wl.aspect, flags=flags, max_iter=30, eps=1e-6)
            s.set_image_points(wl.img)
            try:
                st, r = s.calibrate()
            except Exception as exc:  # noqa: BLE001
                st, r = -99, dict(rms=float("nan"), iters=-1, intr=np.zeros(9), form=str(exc)[:40])
        out.append((st, r))
    (s0, r0), (s1, r1), (s2, r2) = out
    same_rerun = s0 == s1 and r0["iters"] == r1["iters"] and np.array_equal(r0["intr"], r1["intr"])
    agree = s0 == s2 and r0["iters"] == r2["iters"] and (abs(r0["rms"] - r2["rms"]) < 1e-12 or (np.isnan(r0["rms"]) and np.isnan(r2["rms"])))
    if not (same_rerun and agree):
        bad += 1
        print(f"case {case}: {n1}x{n2} N={N} flags=0x{flags:x}: rerun {same_rerun} status {s0}/{s2} iters {r0['iters']}/{r2['iters']} rms {r0['rms']!r} {r2['rms']!r}")
print(f"{a.cases} small problems: {bad} disagreements between the shipped paths and CCAL_NO_SOLO=1 CCAL_WPI=1")
sys.exit(1 if bad else 0)
