"""Soak of the single-host-thread multi-shard fit (ccal_calibrate_multi with one device listed several times: shards on one
stream, exchange as send / receive kernels): the same sharded fit repeated K times per shard count, every result required
to be bit-identical to the first (a lost, stale or torn row would change the sums) and equal to the one-context fit to
rounding.   python tools/multi_soak.py --fits 2000"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import _ffi  # noqa: E402
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver, calibrate_multi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--fits", type=int, default=2000)
ap.add_argument("--images", type=int, default=333)
args = ap.parse_args()
wl = syn.named_workload("C2", n_images=args.images)
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
    s.set_image_points(wl.img)
    _, base = s.calibrate()
bad_total = 0
for shards in (2, 3, 8, 16):
    devs = [0] * shards
    st, first = calibrate_multi(wl.obj, wl.img, wl.image_size, devs, aspect=wl.aspect, flags=wl.flags)
    assert first["iters"] == base["iters"] and abs(first["rms"] - base["rms"]) < 1e-12
    bad, t0 = 0, time.time()
    for i in range(args.fits):
        st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, devs, aspect=wl.aspect, flags=wl.flags)
        if r["rms"] != first["rms"] or not np.array_equal(r["intr"], first["intr"]) or not np.array_equal(r["rvecs"], first["rvecs"]):
            bad += 1
    dt = time.time() - t0
    print(f"{shards:2d} shards on one GPU: {args.fits} complete fits of {args.images} images ({1e3 * dt / args.fits:.2f} ms each, host buffers in and "
          f"out), rms {first['rms']:.12f}, {first['iters']} iterations, deviating results: {bad}", flush=True)
    bad_total += bad
    _ffi.lib().ccal_thread_cleanup()
sys.exit(1 if bad_total else 0)
