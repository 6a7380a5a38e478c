"""Fits and detections from 4 and 8 host threads at once (each thread its own contexts, created and destroyed as it goes; problems of
changing sizes force graph re-builds while other threads allocate, copy and launch): every result equal to the single-threaded one.
   python tools/thread_stress.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from cameracalibrations_amd import fit_model, synthetic as syn
from cameracalibrations_amd.solver import Solver, CRITERIA
from cameracalibrations_amd.detect import detect_corners
jobs = [("REF", 6), ("C1", 25), ("C2", 40), ("REF", 3), ("C5", 5), ("C2", 130), ("C1", 9), ("REF", 12), ("C2", 300), ("C5", 40)]
wls = [syn.named_workload(n, n_images=k, seed_offset=7 * i) for i, (n, k) in enumerate(jobs)]
xys, board = syn.generate_checkerboard((13, 14), 97)
board = board.astype(np.uint8)
def one(i):
    wl = wls[i % len(wls)]
    out = []
    for rep in range(3):
        m = fit_model(wl.image_size, wl.obj, list(wl.img), wl.n_corners, True, wl.aspect)
        with Solver(0) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
            s.set_image_points(wl.img)
            st = s.fit(); r = s.result()
        if i % 5 == 0:
            detect_corners(board, (13, 14))
        out.append((m.frow, m.k, m.rms, m.iters, st, r["iters"], r["rms"]))
    assert out[0] == out[1] == out[2]
    return out[0]
serial = [one(i) for i in range(len(wls))]
errs = 0
for workers in (4, 8):
    with ThreadPoolExecutor(max_workers=workers) as pool:
        res = list(pool.map(one, range(12 * len(wls))))
    for i, r in enumerate(res):
        if r != serial[i % len(wls)]:
            errs += 1
print("thread stress: 2 x", 12 * len(wls), "jobs (fit_model + Solver.fit x3 each, detection mixed in), mismatches", errs)
sys.exit(1 if errs else 0)
