"""Fits and detections from 4 and 8 host threads at once (each thread its own contexts, created and destroyed as it goes; problems of
changing sizes force graph re-builds while other threads allocate, copy and launch): every result equal to the single-threaded one.
   python tools/thread_stress.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from cameracalibrations_amd import fit_model, fit_models, synthetic as syn
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
# many-calls from several threads at once (round 6): every thread its own pool of contexts and its own merge arenas; mixed sizes, so that
# some calibrations are staged through pinned blocks on streams of their own and have to be joined to the merged launches' stream
ref_sets = [syn.named_workload("REF", n_images=k, seed_offset=300 + j) for j, k in enumerate((6, 80, 5, 80, 80, 3, 12, 4))]
ref_alone = [fit_model(w.image_size, w.obj, list(w.img), w.n_corners, True, w.aspect) for w in ref_sets]
def many(i):
    order = [(i + j) % len(ref_sets) for j in range(5 + i % 3)]
    res = fit_models(ref_sets[0].image_size, ref_sets[0].obj, [list(ref_sets[q].img) for q in order], ref_sets[0].n_corners, True, ref_sets[0].aspect)
    bad = 0
    for q, a in zip(order, res):
        b = ref_alone[q]
        if (a.frow, a.fcol, a.crow, a.ccol, a.k, a.rms, a.iters) != (b.frow, b.fcol, b.crow, b.ccol, b.k, b.rms, b.iters) or \
                not np.array_equal(np.array(a.Rs), np.array(b.Rs)) or not np.array_equal(np.array(a.ts), np.array(b.ts)):
            bad += 1
    return bad
serial = [one(i) for i in range(len(wls))]
errs = 0
for workers in (4, 8):
    with ThreadPoolExecutor(max_workers=workers) as pool:
        res = list(pool.map(one, range(12 * len(wls))))
    for i, r in enumerate(res):
        if r != serial[i % len(wls)]:
            errs += 1
many_errs = 0
for workers in (4, 8):
    with ThreadPoolExecutor(max_workers=workers) as pool:
        many_errs += sum(pool.map(many, range(80)))
        # ... and both kinds of work side by side
        f1 = [pool.submit(many, i) for i in range(40)]
        f2 = [pool.submit(one, i) for i in range(2 * len(wls))]
        many_errs += sum(f.result() for f in f1)
        errs += sum(1 for i, f in enumerate(f2) if f.result() != serial[i % len(wls)])
print("thread stress: 2 x", 12 * len(wls), "jobs (fit_model + Solver.fit x3 each, detection mixed in), mismatches", errs)
print("thread stress: 2 x (80 + 40) fit_models calls of 5-7 mixed-size calibrations from 4 and 8 threads, calibrations deviating from fit_model:", many_errs)
errs += many_errs
sys.exit(1 if errs else 0)
