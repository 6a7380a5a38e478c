"""k_init_extrinsics with the normalised homography from (0) OpenCV's own sequence, DLT + polish on normalised points,
(1) the polish started from K^-1 H_pix T_c, (2) K^-1 H_pix T_c decomposed directly: stage time, distance of the poses
from the oracle's (cvFindExtrinsicCameraParams2 restated), and the full fit that follows (iterations, RMS).
   python tools/hmode_probe.py [--workloads C1,C3] > profiles/r04_init_hmode_probe.txt"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402
from oracle import oracle as orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="C1,C3")
ap.add_argument("--images", type=int, default=None)
a = ap.parse_args()
for name in a.workloads.split(","):
    wl = syn.named_workload(name, n_images=a.images)
    W, H = wl.image_size
    K4, _ = orc.init_intrinsics(wl.obj, wl.img, W, H, aspect=wl.aspect if (wl.flags & 2) else 0.0)
    intr0 = np.zeros(9)
    intr0[:4] = K4
    n_or = min(wl.n_images, 1000)
    rt0 = np.array([np.concatenate(orc.find_extrinsics(wl.obj, wl.img[i], intr0)) for i in range(n_or)])
    ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=30, eps=1e-3, solver="schur")
    img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
    for hmode in (0, 1, 2):  # 1 = from the pixel-space DLT result since the second half of round 4
        os.environ["CCAL_INIT_HMODE"] = str(hmode)
        with Solver(0) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=30, eps=1e-3)
            s.set_image_points(img)
            te, tf = [], []
            for rep in range(60):
                s.init_intrinsics()
                t1 = time.perf_counter()
                s.init_extrinsics()
                te.append(time.perf_counter() - t1)
            gi, grt = s.get_init()
            for rep in range(30):
                t0 = time.perf_counter()
                s.fit()
                tf.append(time.perf_counter() - t0)
            r = s.result()
        dr = np.abs(grt[:n_or, :3] - rt0[:, :3]).max()
        dt = np.abs(grt[:n_or, 3:] - rt0[:, 3:]).max()
        print(f"[hmode {hmode}] {wl.name} N={wl.n_images} P={wl.obj.shape[0]}: extrinsics stage {1e6 * np.median(te[10:]):.1f} us, "
              f"one-pipeline fit {1e6 * np.median(tf[5:]):.1f} us | init poses vs oracle: max|d rvec| {dr:.2e} max|d tvec| {dt:.2e} "
              f"(df/f {np.abs(gi[:2] - intr0[:2]).max() / intr0[1]:.1e}) | fit: iters {r['iters']} (oracle {ro['iters']}), "
              f"rms - oracle {r['rms'] - ro['rms']:+.2e}, max|d rvec| {np.abs(r['rvecs'] - ro['rvecs']).max():.1e}", flush=True)
