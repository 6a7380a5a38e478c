"""Randomised device-vs-oracle comparison (run on the GPU box): board shapes, image counts, models, flag words,
noise levels and aspect ratios drawn from a seeded generator; every case runs the full fit (device init + LM,
OpenCV criteria) on both sides.  Prints every disagreement and a summary; exit code 1 if any."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import _ffi as F  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=150)
ap.add_argument("--seed", type=int, default=7)
args = ap.parse_args()
rng = np.random.default_rng(args.seed)
FLAG_BITS = [syn.FIX_ASPECT_RATIO, syn.FIX_PRINCIPAL_POINT, syn.ZERO_TANGENT_DIST, syn.FIX_K1, syn.FIX_K2, syn.FIX_K3,
             syn.FIX_TANGENT_DIST]
bad, skipped, done, both_failed = 0, 0, 0, 0
for case in range(args.cases):
    n1, n2 = int(rng.integers(3, 17)), int(rng.integers(3, 13))
    if n1 * n2 > 192 or n1 * n2 < 6:
        continue
    N = int(rng.choice([1, 2, 3, 4, 5, 7, 12, 20, 33, 64, 130, 257]))
    model = str(rng.choice(["ref", "pinhole", "k1k2p1p2", "full"]))
    flags = 0
    for b in FLAG_BITS:
        if rng.random() < 0.35:
            flags |= int(b)
    sigma = float(rng.choice([0.0, 0.05, 0.3]))
    aspect = float(rng.choice([1.0, 0.8, 1.3])) if (flags & syn.FIX_ASPECT_RATIO) else 1.0
    depth = (14.0, 28.0) if max(n1, n2) > 11 else (8.0, 16.0)
    eps = float(rng.choice([1e-3, 1e-6]))
    try:
        wl = syn.make_workload("fuzz", (n1, n2), N, "ref" if (flags & syn.FIX_ASPECT_RATIO) else model,
                               seed=int(rng.integers(1, 1 << 30)), sigma=sigma, depth=depth, aspect=aspect)
    except Exception as exc:  # noqa: BLE001
        skipped += 1
        continue
    desc = f"case {case}: {n1}x{n2} N={N} model={model} flags=0x{flags:x} sigma={sigma} aspect={aspect} eps={eps}"
    try:
        ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=eps, solver="schur")
    except Exception as exc:  # noqa: BLE001
        ro = None
        o_err = str(exc)
    try:
        with Solver(0) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=eps)
            s.set_image_points(wl.img)
            st, r = s.calibrate()
    except F.CcalError as exc:
        r, st = None, exc.code
    done += 1
    if ro is None and r is None:
        both_failed += 1
    if ro is None or r is None:
        if (ro is None) != (r is None):
            bad += 1
            print("ONE SIDE FAILED", desc, "oracle:", "error" if ro is None else "ok", "device status:", st)
        continue
    if not np.isfinite(ro["rms"]) or ro["rms"] > 5.0:
        skipped += 1                       # ill-posed draw (both sides wander): not a parity statement
        continue
    tol_rms = 1e-7
    ok = (r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < tol_rms
          and np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-5
          and np.abs(r["intr"][4:] - ro["intr"][4:]).max() < 1e-4
          and np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-4 and np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-3)
    if not ok:
        bad += 1
        print("MISMATCH", desc, "iters", r["iters"], ro["iters"], "rms", r["rms"], ro["rms"],
              "d_intr", np.abs(r["intr"] - ro["intr"]).max(), "d_rt", np.abs(r["rvecs"] - ro["rvecs"]).max(),
              np.abs(r["tvecs"] - ro["tvecs"]).max())
print(f"seed {args.seed}: {done} cases compared, {skipped} skipped, {both_failed} refused by both sides, {bad} disagreements")
sys.exit(1 if bad else 0)
