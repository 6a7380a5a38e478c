#!/usr/bin/env python3
"""bench.py -- corner-residuals/sec of the device-side calibrateCamera LM loop (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline workload (config.workload)
  N = 1: BASELINE.json configs[2] -- synthetic pinhole + k1,k2,p1,p2, 1000 images x 9x6 corners on one MI355X.
  N > 1: BASELINE.json configs[3] -- the SAME 1000-image set sharded per image over the N ranks (strong
         scaling), one exchange of the 68-double reduced intrinsic system per LM trial.
Extra blocks of the same JSON line (never `value`):
  "weak_scaling"  N > 1: every rank holds its own 1000 images of one joint fit (1000 N images).
  "c5"            BASELINE.json configs[4] shape, 10000 x 15x10 corners, full k1..k3,p1,p2 model: whole on the
                  one GPU at N = 1, sharded 10000 / N images per rank at N > 1 (1250 per rank at 8).
  "transports"    N > 1: the headline's sharded fit once more over ncclAllReduce called by the library (RCCL) and over
                  the torch.distributed hook, beside the direct row exchange the headline uses.
  "maps"          RowCol <-> XYZ maps and the four RMS error metrics on C3's 54 000 corners (N = 1).
  "detect"        corner detection (the step before the fit, reference src/detect_fit.jl:5-21) on the reference's six
                  example images and on two synthetic boards of its test generator (N = 1).

One step = one complete global LM solve from the closed-form initial values (rows A4/A5
run once on the device before timing and are reported separately, as SURVEY.md 8d defines the metric), observed
corners resident in HBM before the timed region.  value = corner-residual evaluations (each with its 2x(6+K)
Jacobian block accumulated into the normal equations) per second over the whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6    # FP64 vector (= FP64 matrix) peak: 256 CUs x 128 flop/clk x 2.4 GHz
FP64_MEASURED_TFLOPS = 60.0   # dense v_fma_f64 in every wave, measured on MI355X (profiles/r06_lane_group_probe.txt)
# the clock these launches actually hold (s_memtime / s_memrealtime stamps, tools/stamps.py): 2.32 GHz with one wave per SIMD
# (C3, profiles/r04_wave_stamps.json: 2.30-2.32), 2.1 GHz under sustained FP64 load (C5, profiles/r04_c5_pmc_utilisation_summary.json: SQ_WAVE_CYCLES over the waves' lifetime)
CLOCK_GHZ_LATENCY, CLOCK_GHZ_FP64_LOAD, CLOCK_GHZ_PEAK = 2.32, 2.1, 2.4
KERNEL_BOUNDARY_US = 1.45  # dependent kernel boundary, same stream (guide, "boundary" row of the price list)
METRIC = "corner-residuals/sec + LM iters-to-tol, 1000 imgs x 9x6 corners, 1/2/4/8 GPU"

MODEL_TEXT = {
    "pinhole": "synthetic pinhole (fx,fy,cx,cy only, no distortion)",
    "k1k2p1p2": "synthetic pinhole + k1,k2,p1,p2",
    "full": "synthetic pinhole + k1,k2,k3,p1,p2 (full 5-coefficient model)",
    "ref": "synthetic data under the reference's own flag word (f,cx,cy,k1; fixed aspect)",
}


def free_intrinsics(flags):
    """K of SURVEY.md 8: free intrinsic parameters under an OpenCV flag word."""
    from cameracalibrations_amd import synthetic as syn
    free = [True] * 9                                   # fx fy cx cy k1 k2 p1 p2 k3
    if flags & syn.FIX_ASPECT_RATIO:
        free[0] = False
    if flags & syn.FIX_FOCAL_LENGTH:
        free[0] = free[1] = False
    if flags & syn.FIX_PRINCIPAL_POINT:
        free[2] = free[3] = False
    if flags & (syn.ZERO_TANGENT_DIST | syn.FIX_TANGENT_DIST):
        free[6] = free[7] = False
    if flags & syn.FIX_K1:
        free[4] = False
    if flags & syn.FIX_K2:
        free[5] = False
    if flags & syn.FIX_K3:
        free[8] = False
    return sum(free)


def algorithmic_bytes_per_sweep(N, P):
    """SURVEY.md 8d contract figure: 8 B corner read + 96/P B pose read+write per corner-residual."""
    return N * P * (8.0 + 96.0 / P)


def flops_per_corner_residual(K):
    """projection + Jacobian ~250, symmetric JtJ rank-2 update, Jt e."""
    return 250 + 2 * (6 + K) * (7 + K) + 4 * (6 + K)


def committed_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:      # noqa: BLE001
        return None


def cpu_baseline(wl, criteria, budget_s=12.0):
    """The oracle (plain-C port of the reference algorithm, Schur mode, OpenMP over images)
    timed on the host cores on the same workload; repeats until ~budget_s of CPU work."""
    from oracle import oracle as orc
    # the GPU box exposes every host core but grants a 16-core share per GPU
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    units, t_lm, reps = 0, 0.0, 0
    orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=criteria[0],
                  eps=criteria[1], solver="schur", nthreads=cores)       # warm the OpenMP team
    t0 = time.time()
    while time.time() - t0 < budget_s and reps < 200:
        r = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=criteria[0],
                          eps=criteria[1], solver="schur", nthreads=cores)
        units += wl.n_images * wl.n_points * (r["j_sweeps"] + r["e_sweeps"])
        t_lm += r["t_lm_s"]
        reps += 1
    out = {"value": units / t_lm, "unit": "corner-residuals/s", "cores": cores, "kind": "port",
           "sample": f"{reps} x full {wl.name} fit ({wl.n_images} x {wl.n_points}), criteria {criteria}, "
                     f"oracle Schur mode, LM loop only; rms {r['rms']:.6f} px, iters {r['iters']}",
           "build": "oracle/ccal_oracle.c, " + getattr(orc, "BUILD_FLAGS", "gcc -O2 -ffp-contract=off -fopenmp (no -march=native)"),
           "counts": "Jacobian sweeps + error-only sweeps, as OpenCV's loop makes them"}
    # the solve as OpenCV stores it (dense (18+6N)^2 normal equations, one thread) is cubic in N: a 250-image
    # sample of the same workload (the full 1000 images need minutes per fit)
    n = min(wl.n_images, 250)
    rd = orc.calibrate(wl.obj, wl.img[:n], wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=criteria[0],
                       eps=criteria[1], solver="dense", nthreads=1)
    out["dense_solve"] = {"value": n * wl.n_points * (rd["j_sweeps"] + rd["e_sweeps"]) / rd["t_lm_s"],
                          "unit": "corner-residuals/s", "cores": 1, "kind": "port",
                          "sample": f"first {n} images of {wl.name}, OpenCV-faithful dense solve, LM loop only; "
                                    f"{rd['t_lm_s']:.2f} s, iters {rd['iters']}"}
    return out


def opencv_baseline(wl, criteria, budget_s=20.0):
    """if the box happens to have OpenCV's Python binding, time the real thing --
    cv2.calibrateCamera with the reference's inputs (Float32 points, flag word, criteria) -- on a bounded
    sample (its dense (18+6N)^2 solve is cubic in N).  Absent in this image; returns None then (or on any
    error), and the port above stays the only baseline."""
    try:
        import cv2
        n = min(wl.n_images, 60)
        obj = [wl.obj.astype(np.float32)] * n
        img = [wl.img[i].astype(np.float32) for i in range(n)]
        crit = (cv2.TERM_CRITERIA_COUNT + cv2.TERM_CRITERIA_EPS, int(criteria[0]), float(criteria[1]))
        t0 = time.time()
        reps, rms = 0, float("nan")
        while time.time() - t0 < budget_s and reps < 20:
            rms = cv2.calibrateCamera(obj, img, tuple(int(v) for v in wl.image_size), None, None, flags=int(wl.flags),
                                      criteria=crit)[0]
            reps += 1
        dt = (time.time() - t0) / max(reps, 1)
        return {"seconds_per_fit": dt, "images": n, "corners_per_image": wl.n_points, "rms_px": float(rms),
                "what": "cv2.calibrateCamera incl. its own initialisation, single process", "version": cv2.__version__}
    except Exception:      # noqa: BLE001
        return None


def maps_block(wl, rf, with_cpu=True, reps=20):
    """RowCol <-> XYZ maps and the four RMS metrics (reference src/meta.jl:82,88, src/buildcalibrations.jl:37-67) on the
    headline workload's corners, through the same C ABI a Julia caller would use: host arrays in, host arrays out
    (PCIe-inclusive by construction -- these entry points take caller-owned host buffers), one call for all images."""
    from cameracalibrations_amd.solver import Solver
    N, P = wl.n_images, wl.n_points
    f, c, k = rf["intr"][:2].copy(), rf["intr"][2:4].copy(), float(rf["intr"][4])   # the Julia maps carry one radial coefficient
    rv, tv = rf["rvecs"], rf["tvecs"]
    obj = wl.obj.astype(np.float64)
    img = wl.img.astype(np.float64)
    rng = np.random.default_rng(0)
    inv_pts = rng.random((N, 100, 2)) * (np.asarray(wl.image_size, dtype=np.float64) - 1) + 1     # inverse_samples = 100
    out = {"points": N * P, "stat": "median over %d calls per quantity (200 for the single-point call)" % reps,
           "what": "one C-ABI call per quantity over all images, host buffers in and out (PCIe inclusive); camera = "
                                    "this fit seen through the reference's single-k map model (k2, p1, p2, k3 dropped)"}
    with Solver(0) as s:
        def rate(fn, units):
            # median over the calls: one call in a few thousand on these boxes stalls for tens of milliseconds
            # (host side, seen with every entry point), which a mean over 20 calls would report as the rate
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = fn()
                ts.append(time.perf_counter() - t0)
            return units / float(np.median(ts)), r
        out["forward_points_per_s"], rc = rate(lambda: s.project_many(obj, f, c, k, rv, tv, 1.0), N * P)
        out["inverse_points_per_s"], _ = rate(lambda: s.unproject_many(rc, f, c, k, rv, tv, 1.0), N * P)
        out["errors_corners_per_s"], e = rate(lambda: s.errors(wl.n_corners, obj, img, f, c, k, rv, tv, 1.0, inv_pts=inv_pts), N * P)
        out["errors"] = {kk: float(v) for kk, v in e.items()}
        one = np.array([[1.0, 2.0, 0.0]])
        for _ in range(20):                      # (the first call loads the code object of the completion kernel)
            s.project(one, f, c, k, rv[0], tv[0], 1.0)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter()
            s.project(one, f, c, k, rv[0], tv[0], 1.0)
            ts.append(time.perf_counter() - t0)
        out["single_point_call_us"] = 1e6 * float(np.median(ts))
        out["single_point_call_us_mean"] = 1e6 * float(np.mean(ts))
    # a caller that loops over points (the reference's own round-trip metric does, src/buildcalibrations.jl:53-58): 100
    # point-wise calls one by one, the same loop through Calibration.batch() (queued, evaluated in one device call), and one
    # call with the 100 points as an array
    from cameracalibrations_amd.calibration import Calibration
    cal = Calibration(f, c, rv, tv, 1.0, k, [])
    p100 = [np.r_[rng.uniform(0, 8, 2), 0.0] for _ in range(100)]
    arr100 = np.array(p100)

    def med(fn, n=30):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return 1e6 * float(np.median(ts))

    def buffered():
        with cal.batch() as b:
            for x in p100:
                b(x, 1)
    out["loop_of_100_points_us"] = med(buffered)
    out["loop_of_100_points_unbuffered_us"] = med(lambda: [cal(x, 1) for x in p100], n=5)
    out["one_call_100_points_us"] = med(lambda: cal(arr100, 1))
    if with_cpu:
        from oracle import oracle as orc
        n = min(N, 200)
        t0 = time.perf_counter()
        for i in range(n):
            orc.map_forward(obj, f, c, k, rv[i], tv[i], 1.0)
        out["cpu_forward_points_per_s"] = n * P / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for i in range(n):
            orc.map_inverse(rc[i], f, c, k, rv[i], tv[i], 1.0)
        out["cpu_inverse_points_per_s"] = n * P / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        eo = orc.errors(wl.n_corners, obj, img, f, c, k, rv, tv, 1.0, inv_pts=inv_pts)
        out["cpu_errors_corners_per_s"] = N * P / (time.perf_counter() - t0)
        out["cpu"] = f"oracle (plain C, one thread), {n} images for the maps, all {N} for the metrics"
        out["errors_vs_cpu_max_abs"] = float(max(abs(eo[kk] - e[kk]) for kk in eo))
    return out


def detect_block(with_cpu=True, reps=10):
    """Corner detection through the C ABI (host image in, RowCol corners out): the reference's six example images
    (tests/golden/example_png, expected corners from its test/example/corners.json) and two boards of its synthetic
    generator (test/runtests.jl:6-43, restated in oracle/detect_oracle.py)."""
    from PIL import Image

    from cameracalibrations_amd.detect import detect_corners
    ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
    nc = tuple(ex["n_corners"])
    cases = [(np.ascontiguousarray(np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", "example_png", f)))), nc,
              np.array(t, dtype=np.float64)) for f, t in zip(ex["files"], ex["corners"])]
    from cameracalibrations_amd.synthetic import generate_checkerboard
    for w, h, ratio in ((13, 14, 95), (7, 10, 60)):
        xys, img = generate_checkerboard((w, h), ratio)
        cases.append((np.ascontiguousarray(img.astype(np.uint8)), (w, h), xys))
    out = {"images": len(cases), "pixels": int(sum(im.size for im, _, _ in cases)), "stat": "median over %d passes over the images" % reps,
           "what": "ccal_detect_corners per image, host buffer in, corners out (PCIe inclusive)"}
    worst = 0.0
    for im, n, truth in cases:                    # warm-up + the reference's bar (every corner within 1 px)
        res = detect_corners(im, n)
        if res is None:
            raise RuntimeError("board not found")
        worst = max(worst, float(np.linalg.norm(res.transpose(1, 0, 2).reshape(-1, 2) - truth, axis=1).max()))
    out["max_corner_error_px"] = worst
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for im, n, _ in cases:
            detect_corners(im, n)
        ts.append(time.perf_counter() - t0)
    out["images_per_s"] = len(cases) / float(np.median(ts))
    out["pixels_per_s"] = out["pixels"] / float(np.median(ts))
    # the six example images as one calibration's images: image-by-image calls against one ccal_detect_corners_many (three in flight)
    from cameracalibrations_amd.detect import detect_corners_many
    six = [im for im, _, _ in cases[:6]]
    a = [detect_corners(im, nc) for im in six]
    b = detect_corners_many(six, nc)
    t1, t2 = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        for im in six:
            detect_corners(im, nc)
        t1.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        detect_corners_many(six, nc)
        t2.append(time.perf_counter() - t0)
    out["example_images"] = {"images": 6, "us_per_image_one_at_a_time": 1e6 * float(np.median(t1)) / 6, "us_per_image_three_in_flight": 1e6 * float(np.median(t2)) / 6,
                             "identical": bool(all(np.array_equal(x, y) for x, y in zip(a, b)))}
    if with_cpu:
        from oracle import detect_oracle as D      # (the CPU comparison only)
        t0 = time.perf_counter()
        ro = [D.detect_corners(im, n) for im, n, _ in cases[:3]]
        out["cpu_images_per_s"] = 3 / (time.perf_counter() - t0)
        out["cpu"] = "oracle/detect_oracle.py (numpy, one thread), first 3 images"
        out["vs_cpu_max_abs_px"] = max(float(np.abs(detect_corners(im, n).transpose(1, 0, 2).reshape(-1, 2) - r).max())
                                       for (im, n, _), r in zip(cases[:3], ro))
    return out


def small_block(job, name, n_images, criteria, with_cpu=True):
    """a small fit on one GPU (BASELINE.json configs[1]: 100 x 9x6 pinhole) with the same quantities as the headline"""
    from cameracalibrations_amd import synthetic as syn
    wl = syn.named_workload(name, n_images=n_images)
    s, _, _ = job.make_solver(wl, wl.img, wl.n_images, criteria)
    steps = 100
    el, r = job.timed_solves(s, steps, 10)
    us, n, b2b, _ = job.launch_timing(s, 30)
    s.fit()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        s.fit()
        ts.append(time.perf_counter() - t0)
    out = {"workload": f"{wl.name}: {wl.n_images} images x {wl.n_corners[0]}x{wl.n_corners[1]} corners, {free_intrinsics(wl.flags)} free intrinsics",
           "value": float(wl.n_images) * wl.n_points * r["sweeps"] * steps / el, "unit": "corner-residuals/s",
           "ms_per_step": 1e3 * el / steps, "steps": steps, "full_fit_ms": 1e3 * float(np.median(ts)),
           "avg_launch_us": us, "median_launch_us": job.median_launch_us, "launches_timed": n, "avg_launch_us_back_to_back": b2b,
           "kernel": kernel_name(job.sweep_form), "lm": lm_block(r)}
    s.close()
    if with_cpu:
        from oracle import oracle as orc
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
        ro = [orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=criteria[0], eps=criteria[1],
                            solver="schur", nthreads=cores) for _ in range(5)]
        out["cpu_oracle"] = {"lm_ms": 1e3 * float(np.median([q["t_lm_s"] for q in ro])), "cores": cores, "kind": "port",
                             "what": "oracle Schur mode, LM loop only, median of 5", "rms_px": ro[-1]["rms"], "iters": int(ro[-1]["iters"])}
    return out


def ref_example_block(with_cpu=True):
    """The reference's own regime: its example's six images x 5x8 corners (test/example/corners.json -> tests/golden) through the
    drop-in seam fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect) (src/detect_fit.jl:27-61), host arrays in
    and out like the Julia caller's; and the same corners resident, as one pipeline."""
    import cameracalibrations_amd as cc
    from cameracalibrations_amd.solver import Solver, reference_flags
    ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
    img = [np.array(m, dtype=np.float64) for m in ex["corners"]]
    obj = cc.board_template(ex["n_corners"])

    def med(fn, n):
        fn()
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts)), r
    ms, m = med(lambda: cc.fit_model(ex["image_size_wh"], obj, img, ex["n_corners"], True, 1), 50)
    out = {"workload": "test/example/corners.json: 6 images x 5x8 corners, reference flag word (K = 4: f, cx, cy, k1), criteria (30, 1e-3)",
           "full_fit_ms": ms, "what": "fit_model(...) per call, host lists in / NamedTuple-like out, median of 50 (PCIe and Python inclusive)",
           "rms_px": float(m.stats["rms"]), "iters": int(m.stats["iters"]), "sweeps": int(m.stats["sweeps"])}
    arr = np.ascontiguousarray(np.stack(img), dtype=np.float32)
    with Solver(0) as s:
        s.set_problem(obj, len(img), ex["image_size_wh"], aspect=1.0, flags=reference_flags(True), max_iter=30, eps=1e-3)
        s.set_image_points(arr)
        ms_res, _ = med(lambda: s.fit(), 50)
        s.init_intrinsics()
        s.init_extrinsics()
        ms_lm, _ = med(lambda: s.lm_run(), 100)
        rr = s.result()
        out.update({"full_fit_resident_ms": ms_res, "ms_per_step": ms_lm, "kernel": kernel_name(rr["sweep_form"]),
                    "resident": "the same corners resident in HBM: ccal_fit (A4-A9 as one pipeline) per call | ccal_lm_run per call"})
    if with_cpu:
        from oracle import oracle as orc
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ro = orc.calibrate(obj, arr, ex["image_size_wh"], aspect=1.0, flags=reference_flags(True), max_iter=30, eps=1e-3,
                               solver="dense", nthreads=1)
            ts.append(time.perf_counter() - t0)
        out["cpu_oracle"] = {"full_fit_ms": 1e3 * float(np.median(ts)), "lm_ms": 1e3 * ro["t_lm_s"], "cores": 1, "kind": "port",
                             "what": "oracle, OpenCV-faithful dense mode incl. its initialisation, one thread, median of 5",
                             "rms_px": ro["rms"], "iters": int(ro["iters"])}
    return out


def many_block(contexts=32, rounds=25):
    """Independent fits in flight together (ccal_fit_many: every context's A4-A9 pipeline enqueued before any wait) against the same
    contexts fitted one after the other, corners resident: microseconds per fit.  The caller this stands for calibrates many cameras
    (the loop around src/buildcalibrations.jl:12-26)."""
    import cameracalibrations_amd as cc
    from cameracalibrations_amd import synthetic as syn
    from cameracalibrations_amd.solver import CRITERIA, Solver, reference_flags
    ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
    ex_img = np.stack([np.array(m, dtype=np.float32).reshape(-1, 2) for m in ex["corners"]])
    rng = np.random.default_rng(7)

    def problem(name, i):
        if name == "ref_example":
            return cc.board_template(ex["n_corners"]), (ex_img + rng.normal(0, 0.05, ex_img.shape)).astype(np.float32), ex["image_size_wh"], 1.0, reference_flags(True)
        wl = syn.named_workload(name, seed_offset=i)
        return wl.obj, wl.img, wl.image_size, wl.aspect, wl.flags
    out = {"contexts": contexts, "what": "us per fit, median over rounds, each kind of call in its steady state: the contexts' ccal_fit one after the other | "
                                         "ccal_fit_many over all of them, call after call | the same right after the contexts' own fits (one stream join "
                                         "per context); different problems of one shape, corners resident, results bit-equal (tests)"}
    for name in ("ref_example", "C1", "C3"):
        solvers = []
        for i in range(contexts):
            obj, img, size, aspect, flags = problem(name, i)
            s = Solver(0)
            s.set_problem(obj, img.shape[0], size, aspect=aspect, flags=flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
            s.set_image_points(img)
            solvers.append(s)
        # each way of calling in ITS steady state (calls of one kind in a row); and the many-call right after the contexts' own fits,
        # where every context whose last launches ran on its own stream is joined to the merged launches' stream once (an event each)
        t_seq, t_many, t_after = [], [], []
        for rep in range(rounds + 3):
            t0 = time.perf_counter()
            for s in solvers:
                s.fit()
            t_seq.append((time.perf_counter() - t0) / contexts)
        for rep in range(rounds + 3):
            t0 = time.perf_counter()
            cc.fit_many(solvers)
            t_many.append((time.perf_counter() - t0) / contexts)
        for rep in range(8):
            for s in solvers:
                s.fit()
            t0 = time.perf_counter()
            cc.fit_many(solvers)
            t_after.append((time.perf_counter() - t0) / contexts)
        a, b = float(np.median(t_seq[3:])), float(np.median(t_many[3:]))
        out[name.lower()] = {"us_per_fit_one_at_a_time": 1e6 * a, "us_per_fit_many": 1e6 * b, "speedup": a / b,
                             "us_per_fit_many_right_after_own_fits": 1e6 * float(np.median(t_after[2:])),
                             "images": int(img.shape[0]), "corners_per_image": int(img.shape[1])}
        for s in solvers:
            s.close()
    # the same through the drop-in seam, host arrays in and out (PCIe and Python inclusive): fit_model per calibration | one fit_models
    obj = cc.board_template(ex["n_corners"])
    base = np.stack([np.array(m, dtype=np.float64).reshape(-1, 2) for m in ex["corners"]])
    sets = [list(base + rng.normal(0, 0.05, base.shape)) for _ in range(contexts)]
    t_seq, t_many = [], []
    for rep in range(rounds + 3):
        t0 = time.perf_counter()
        for pts in sets:
            cc.fit_model(ex["image_size_wh"], obj, pts, ex["n_corners"], True, 1)
        t1 = time.perf_counter()
        cc.fit_models(ex["image_size_wh"], obj, sets, ex["n_corners"], True, 1)
        t2 = time.perf_counter()
        t_seq.append((t1 - t0) / contexts)
        t_many.append((t2 - t1) / contexts)
    a, b = float(np.median(t_seq[3:])), float(np.median(t_many[3:]))
    out["ref_example_drop_in"] = {"us_per_calibration_fit_model": 1e6 * a, "us_per_calibration_fit_models": 1e6 * b, "speedup": a / b,
                                  "what": "host lists in / namespaces out: fit_model called per calibration | one fit_models(...) = ccal_calibrate_many"}
    return out


class BlockFailed(RuntimeError):
    """Raised by EVERY rank at the same point of a block (after an all_agree), so that a per-block try / except never
    lets one rank skip a collective the others enter."""


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as plain child processes of THIS process --
    which has made no HIP call and never will -- with the environment torch.distributed.run would give them, relay
    their output (rank 0 prints the JSON line), and exit non-zero if any of them does."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CCAL_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: the inbox handles cross processes
        env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and deadline is None:
                # a rank died: give the others a while to notice (bounded waits in the library), then end them
                rc, deadline = p.returncode, time.time() + float(os.environ.get("CCAL_BENCH_GRACE_S", "30"))
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                      # (this exact child, by its handle)
        time.sleep(0.05)
    for p in procs:
        rc = rc or (p.returncode or 0)
    return rc


def dry_run(args):
    """The launch plumbing without a GPU: every rank joins a gloo group over the environment it was started with,
    the ranks all-reduce their numbers and rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    world, rank = args.gpus, int(os.environ.get("RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) != world:
        raise SystemExit(f"bench.py: --gpus {world} but WORLD_SIZE={os.environ.get('WORLD_SIZE')} in the environment")
    if os.environ.get("CCAL_BENCH_FAIL_RANK") == str(rank):        # (tests: a rank that dies before the rendezvous)
        return 7
    t = torch.tensor([rank + 1], dtype=torch.int64)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "sum_of_rank_numbers": int(t.item()),
                          "launcher": "own child processes" if os.environ.get("CCAL_BENCH_CHILD") else "external"}))
    return 0


DEADLINE_EXIT = 3      # exit status of a run whose optional last block hung (the JSON line has been printed)


class Deadline:
    """`with Deadline(seconds, rank, line)`: if the body is still running after `seconds` (a transport stuck in a call that
    never returns), rank 0 prints the line it has (`line()` -> dict, marked `"incomplete": true`) and the process ends with
    status DEADLINE_EXIT (non-zero: the measured line is out, but a hung transport is a finding, not a clean run); every rank
    holds the same deadline and ends by itself.  Only for optional blocks after everything that counts has been measured."""

    def __init__(self, seconds, rank, line):
        self.seconds, self.rank, self.line, self.timer = seconds, rank, line, None

    def _fire(self):
        try:
            if self.rank == 0:
                print(json.dumps(dict(self.line(), incomplete=True)), flush=True)
            print(f"[bench] rank {self.rank}: optional block abandoned after {self.seconds:.0f} s", file=sys.stderr, flush=True)
        finally:
            os._exit(DEADLINE_EXIT)

    def __enter__(self):
        import threading
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


class Job:
    """One rank's side of the bench: process group, transports, timing helpers."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.args = torch, dist, args
        self.world = args.gpus
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            if int(os.environ.get("WORLD_SIZE", "1")) != self.world:
                raise SystemExit(f"bench.py: --gpus {self.world} but WORLD_SIZE={os.environ.get('WORLD_SIZE')} in the environment")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.share_gpu:
                self.local_rank = 0
                torch.cuda.set_device(0)
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
        else:
            torch.cuda.set_device(0)
        self.shared_device = False
        self.dev = torch.device("cuda", self.local_rank)
        self.cdev = "cpu" if (self.world > 1 and args.share_gpu) else self.dev   # where torch.distributed tensors live

    # ---- collectives over the bench's own process group (never inside the timed data path) ----
    def all_agree(self, ok):
        if self.world == 1:
            return bool(ok)
        t = self.torch.tensor([1 if ok else 0], dtype=self.torch.int32, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def together(self, fn, what):
        """Run the rank-local `fn`; all ranks then agree on the outcome and either all go on or all raise BlockFailed
        here.  (fn may make library calls that exchange rows with the peers: those waits are bounded, a peer's failure
        comes back as an error on every rank.)"""
        ok, why, val = True, "", None
        try:
            val = fn()
        except Exception as exc:      # noqa: BLE001
            ok, why = False, f"{what}: {exc}"
        if not self.all_agree(ok):
            raise BlockFailed(why or f"{what}: a peer rank failed")
        return val

    def max_over_ranks(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()

    # ---- transports: every stage is decided by ALL ranks together (a rank never changes transport alone) ----
    def attach_comm(self, sol, n_total, want=None):
        from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, broadcast_id_via_torch
        if self.world == 1:
            return "none"
        world, rank, want = self.world, self.rank, want or self.args.allreduce
        if want == "p2p":
            why = ""
            try:
                # (the all-gather inside is always joined; ranks that find a device id twice -- a rehearsal on one GPU --
                # all take the one-workgroup exchange, so that a waiting rank holds one compute unit, not a grid)
                self.shared_device = sol.init_p2p(world, rank, n_total, allgather_bytes_via_torch(),
                                                  shared_device=True if self.args.share_gpu else None)
                ok = True
            except Exception as exc:      # noqa: BLE001
                ok, why = False, str(exc)
            if self.all_agree(ok):
                self.dist.barrier()       # every inbox is mapped before the first row is sent
                try:
                    sol.p2p_selftest(16)
                    # ... and the real thing: init (8-double exchange) + two sharded solves (fused exchange, graph replay)
                    sol.init_intrinsics()
                    sol.init_extrinsics()
                    sol.lm_run()
                    sol.lm_run()
                except Exception as exc:  # noqa: BLE001
                    ok, why = False, str(exc)
                if self.all_agree(ok):
                    return ("p2p (direct xGMI row exchange kernels, self-tested" +
                            ("; one exchange workgroup per rank: the ranks share a GPU)" if self.shared_device else ")"))
            print(f"[bench] rank {rank}: direct exchange unavailable ({why or 'a peer failed'}); all ranks use RCCL",
                  file=sys.stderr)
        if want in ("p2p", "rccl") and not self.args.share_gpu:   # (RCCL refuses two ranks on one device: rehearsals skip it)
            # the broadcast of the id is collective: every rank joins it whether or not its own part worked
            ok, why = True, ""
            ident = None
            try:
                from cameracalibrations_amd import _ffi as F
                import ctypes as C
                buf = C.create_string_buffer(128)
                if rank == 0:
                    F.check(F.lib().ccal_comm_unique_id(buf))
                ident = buf.raw
            except Exception as exc:      # noqa: BLE001
                ok, why = False, str(exc)
            ident = broadcast_id_via_torch()(ident if (rank == 0 and ok) else (bytes(128) if rank == 0 else None))
            ok = ok and ident != bytes(128)
            if self.all_agree(ok):
                try:
                    sol.init_rccl(world, rank, n_total, lambda _b: ident)
                except Exception as exc:  # noqa: BLE001
                    ok, why = False, str(exc)
                if self.all_agree(ok):
                    return "rccl (ncclAllReduce inside the library)"
            # decided by all ranks together: a rank whose own RCCL set-up worked drops it again (set_allreduce below)
            print(f"[bench] rank {rank}: native RCCL unavailable ({why or 'a peer failed'}); all ranks use the torch hook",
                  file=sys.stderr)
        sol.set_allreduce(TorchAllReduce(), world, rank, n_total)
        return "torch.distributed all_reduce hook"

    # ---- one workload: solver set-up, timed solves, launch timing ----
    def make_solver(self, wl, my_img, n_total, criteria, want=None):
        from cameracalibrations_amd.solver import Solver

        def local():
            img_dev = self.torch.from_numpy(np.ascontiguousarray(my_img)).to(self.dev)   # inputs resident in HBM
            self.torch.cuda.synchronize()
            # the context's own non-blocking stream (not torch's legacy default stream, which serialises with every
            # blocking stream of the process): the timed region is bracketed by device-wide synchronisation anyway
            s = Solver(self.local_rank)
            s.set_problem(wl.obj, my_img.shape[0], wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=criteria[0],
                          eps=criteria[1])
            s.set_image_points(img_dev)
            return s, img_dev
        s, img_dev = self.together(local, "solver set-up")
        comm = self.attach_comm(s, n_total, want)              # (decides every stage with all ranks itself)

        def init():
            s.init_intrinsics()
            s.init_extrinsics()
        self.together(init, "closed-form initialisation")
        return s, img_dev, comm

    def timed_solves(self, s, steps, warmup):
        self.together(lambda: [s.lm_run() for _ in range(warmup)], "warm-up solves")
        self.barrier()
        t0 = time.perf_counter()
        ok, why = True, ""
        try:
            for _ in range(steps):
                s.lm_run()
        except Exception as exc:      # noqa: BLE001
            ok, why = False, str(exc)
        self.barrier()                                         # (reached by every rank whatever happened above)
        dt = time.perf_counter() - t0
        if not self.all_agree(ok):
            raise BlockFailed(why or "timed solves: a peer rank failed")
        return self.max_over_ranks(dt), s.result()

    def launch_timing(self, s, steps):
        """mean device time of the sweeping k_lm_trial launches inside solves (HIP events bound to the dispatches on
        the launch stream) and of back-to-back first-trial launches"""
        def timed():
            s.set_sweep_timing(True)
            us, n_timed, t1, self.solve_kernels_us, per_solve = 0.0, 0, time.perf_counter(), 0.0, []
            for _ in range(steps):
                s.lm_run()
                rr = s.result()
                us += rr["sweep_us"] * rr["timed_sweeps"]
                n_timed += rr["timed_sweeps"]
                if rr["timed_sweeps"]:
                    per_solve.append(rr["sweep_us"])
                self.solve_kernels_us += rr["solve_kernels_us"] / max(steps, 1)   # all launches of a solve, summed
                self.sweep_form = rr["sweep_form"]
            self.torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            s.set_sweep_timing(False)
            # (surplus launches that return at once and re-damping sweeps are not among the timed ones: ccal_lm_run counts only
            #  the launches whose trial evaluated a point; the median is over the solves' own means)
            self.median_launch_us = float(np.median(per_solve)) if per_solve else 0.0
            return us / max(n_timed, 1), n_timed, s.time_sweep(200), dt / max(steps, 1)
        return self.together(timed, "launch timing")

    def eager_solves(self, s, steps, warmup):
        """the same solves launched kernel by kernel instead of replayed as one hipGraph (host cost only differs)"""
        s.set_graph(False)
        try:
            return self.timed_solves(s, steps, warmup)[0]
        finally:
            s.set_graph(True)


def shard(wl, world, rank):
    from cameracalibrations_amd.distributed import shard_range
    lo, hi = shard_range(wl.n_images, world, rank)
    return wl.img[lo:hi]


# Latency floor of ONE LM trial, from quantities the kernel does not choose (measured constants: the guide's price list
# and profiles/r04_latency_probe.txt, tools/lat_probe.hip): what any one-launch-per-trial design pays when every image
# has a SIMD to itself (the C3 regime).  Not this kernel's own instruction stream: the kernel sits well above it.
CLOCK_GHZ = CLOCK_GHZ_LATENCY   # measured in-kernel clock of the one-wave-per-SIMD launches
LAT = {"fma": 4.2, "rcp": 16.2, "rsq": 16.2, "dpp": 12.2, "readlane": 28.2, "lds_round_trip": 128.0,   # cycles (profiles/r04_latency_probe.txt)
       "global_state": 740.0, "global_atomics": 480.0}      # dependent global round trips at kernel start / end
LATENCY_FLOOR_SOURCE = (
    "boundary: dependent kernel boundary, guide price list; global: the two true dependent global round trips of a trial "
    "(previous state + accumulators in, 740 cycles; the memory-side atomics out, 480 cycles; DESIGN.md section 4); chain: the "
    "minimum dependent FP64 chain of one image's trial at the measured per-step latencies of profiles/r04_latency_probe.txt "
    "(v_fma_f64 4.2, v_rcp/v_rsq_f64 16.2, v_mov_b64_dpp 12.2, readlane + use 28.2 cycles; LDS write -> read 128) -- limbs -> "
    "doubles 10 fma; 9 Gauss-Jordan pivots x (dpp + rcp + 3 fma) + quotient; state hand-off 2 LDS round trips; pose 1 mul + "
    "4 dpp adds + readlane + 1 add; rotation series 3 + 11 + 4 fma; projection + Jacobian row 3 + 1 + rcp + 4 + 18 fma; "
    "rows -> tile -> operands 1 LDS round trip; Gram -> LDS -> factor 1 round trip; 6 Cholesky columns x (fma + rsq + 6 "
    "refinement steps + mul); 6 forward-substitution steps x 2; Y -> LDS, 6 fma dots, workgroup sum through LDS, 3 adds, "
    "limb split ~ 15 integer steps; gram: the image's 2P x 16 Gram (symmetric half, 2 x 136 flop per row) at the FP64 matrix "
    "rate of ONE SIMD, 32 flop per cycle -- a throughput term, one image per SIMD")


def latency_floor(P):
    """-> (floor in us, its terms in us): boundary + dependent global round trips + dependent FP64 chain + Gram"""
    L = LAT
    chain = (10 * L["fma"]                                                  # limbs -> doubles
             + 9 * (L["dpp"] + L["rcp"] + 3 * L["fma"]) + L["rcp"] + 6 * L["fma"] + L["dpp"]   # damped 9x9 solve, candidate
             + 2 * L["lds_round_trip"]                                      # state: wave 0 -> LDS -> every wave
             + L["fma"] + 4 * (L["dpp"] + L["fma"]) + L["readlane"] + L["fma"]                 # candidate pose
             + 18 * L["fma"]                                                # R, K: |r|^2, 11-term Horner, entries
             + 4 * L["fma"] + L["rcp"] + 4 * L["fma"] + 18 * L["fma"]       # X, 1/Z (+2 Newton), x .. residual / Jacobian row
             + 2 * L["lds_round_trip"]                                      # rows -> tile -> operands; Gram -> LDS -> factor
             + 6 * (L["fma"] + L["rsq"] + 7 * L["fma"])                     # Cholesky: pivot, rsqrt + refinement, scale
             + 12 * L["fma"]                                                # forward substitution
             + 2 * L["lds_round_trip"] + 6 * L["fma"] + 3 * L["fma"] + 15 * L["fma"])          # Y, dots, workgroup sum, limb split
    gram = 2.0 * P * 136 * 2 / 32.0                                         # cycles on one SIMD's matrix pipe
    terms = {"boundary": KERNEL_BOUNDARY_US,
             "global_round_trips": (L["global_state"] + L["global_atomics"]) / (CLOCK_GHZ * 1e3),
             "dependent_fp64_chain": chain / (CLOCK_GHZ * 1e3),
             "gram_at_matrix_peak_one_simd": gram / (CLOCK_GHZ * 1e3)}
    return sum(terms.values()), terms


def kernel_name(form, kind=1):
    """the k_lm_trial instantiation as rocprofv3 prints it, from ccal_stats.sweep_form (FLOW | LOOP << 4 | WPI << 8 | SOLO << 12 | MULTI << 16)"""
    tf = lambda b: "true" if b else "false"  # noqa: E731
    return "ccal::k_lm_trial<%d, %s, %d, %d, %s, %s>" % (form & 15, tf((form >> 4) & 1), kind, (form >> 8) & 15, tf((form >> 12) & 1), tf((form >> 16) & 1))


def roofline_block(N_rank, P, K, sweep_us, n_timed, b2b_us, events_ms_per_step, kernel, median_us=None):
    """the dominant kernel against its ceilings: HBM (the contract's classification), FP64 and the latency floor"""
    alg_bytes = algorithmic_bytes_per_sweep(N_rank, P)
    achieved = alg_bytes / (sweep_us * 1e-6) / 1e9 if sweep_us > 0 else 0.0
    flops = N_rank * P * flops_per_corner_residual(K)
    tflops = flops / (sweep_us * 1e-6) / 1e12 if sweep_us > 0 else 0.0
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "traffic": None, "traffic_source": None, "kernel": kernel,
           "algorithmic_bytes_per_launch": alg_bytes,
           # SURVEY.md 8d diagnostic: the same sweep with the per-image blocks kept in HBM (this design: one
           # 1-KB block read for both parities and written once per image and sweep)
           "bytes_per_launch_with_blocks_in_hbm": alg_bytes + N_rank * 3 * 1024,
           "avg_launch_us": sweep_us, "median_launch_us": median_us, "launches_timed": n_timed, "avg_launch_us_back_to_back": b2b_us,
           "timed_region_with_events_ms_per_step": events_ms_per_step,
           "fp64_flop_per_launch": flops, "fp64_tflops": tflops, "fp64_peak_tflops": FP64_PEAK_TFLOPS,
           "fp64_frac": tflops / FP64_PEAK_TFLOPS}
    # traffic: HBM-side bytes per launch from the committed rocprofv3 PMC passes of this same command (tools/pmc.sh:
    # FETCH_SIZE and WRITE_SIZE in separate passes, both calibrated on a known byte count with this access width)
    for shape, names in (((1000, 54), ("r05_pmc_traffic_summary.json", "r04_pmc_traffic_summary.json", "r03_pmc_traffic_summary.json")),
                         ((10000, 150), ("r05_c5_pmc_traffic_summary.json", "r04_c5_pmc_traffic_summary.json", "r03_c5_pmc_traffic_summary.json"))):
        for name in names:
            pmc = committed_profile(name)
            if pmc and (N_rank, P) == shape and "ccal::k_lm_trial" in pmc and out["traffic"] is None:
                out["traffic"] = float(pmc["ccal::k_lm_trial"]["hbm_bytes_per_launch"])
                out["traffic_source"] = f"committed profile profiles/{name} (rocprofv3 PMC passes of this command, not measured in this run)"
    floor_us, terms = latency_floor(P)
    if N_rank > 1024:      # more images than SIMDs: a launch is FP64 throughput, not one image's latency chain
        terms = {"boundary": KERNEL_BOUNDARY_US, "fp64_flop_at_vector_peak": flops / (FP64_PEAK_TFLOPS * 1e6)}
        floor_us = sum(terms.values())
        # the chip does not hold 2.4 GHz under this load: the same floor at the clock it holds, and what that makes of the launch
        held = FP64_PEAK_TFLOPS * CLOCK_GHZ_FP64_LOAD / CLOCK_GHZ_PEAK
        out["fp64_peak_tflops_at_held_clock"] = held
        out["fp64_frac_at_held_clock"] = tflops / held
        out["held_clock_ghz"] = CLOCK_GHZ_FP64_LOAD
        out["latency_floor_us_at_held_clock"] = KERNEL_BOUNDARY_US + flops / (held * 1e6)
        out["frac_of_floor_at_held_clock"] = out["latency_floor_us_at_held_clock"] / sweep_us if sweep_us > 0 else None
        # what the vector unit of this chip sustains on dense independent v_fma_f64 in every wave (profiles/r06_lane_group_probe.txt,
        # tools/lg_probe.hip: 57-62 TFLOP/s at the 2.0 GHz it then holds) -- a diagnostic beside the contract's peak, never instead of it
        out["fp64_vector_rate_measured_tflops"] = FP64_MEASURED_TFLOPS
        out["fp64_frac_of_measured_rate"] = tflops / FP64_MEASURED_TFLOPS
    else:
        out["held_clock_ghz"] = CLOCK_GHZ_LATENCY
    out["latency_floor_us"] = floor_us
    out["frac_of_floor"] = floor_us / sweep_us if sweep_us > 0 else None
    out["latency_floor_terms_us"] = terms
    out["latency_floor_source"] = LATENCY_FLOOR_SOURCE
    return out


def lm_block(r, rt=None, tight_ms=None):
    d = {"iters_to_tol_1e-3": r["iters"], "sweeps_per_solve": r["sweeps"], "redamping_sweeps_per_solve": r["reschur"],
         "rejected": r["rejected"], "rms_px": r["rms"]}
    if rt is not None:
        d.update({"iters_to_tol_1e-10": rt["iters"], "sweeps_1e-10": rt["sweeps"], "redamping_sweeps_1e-10": rt["reschur"],
                  "rejected_1e-10": rt["rejected"], "rms_px_1e-10": rt["rms"], "ms_per_solve_1e-10": tight_ms})
    return d


def config_block(wl, model, cfg_index, N_rank, n_total, world, comm, criteria, strong):
    from cameracalibrations_amd import synthetic as syn  # noqa: F401
    K = free_intrinsics(wl.flags)
    if world == 1:
        where = f"{n_total} images x {wl.n_corners[0]}x{wl.n_corners[1]} corners on 1 MI355X (BASELINE.json configs[{cfg_index}])"
    elif strong:
        where = (f"the same {n_total} images x {wl.n_corners[0]}x{wl.n_corners[1]} corners sharded per image over {world} "
                 f"MI355X, {N_rank} on rank 0 (BASELINE.json configs[{cfg_index}])")
    else:
        where = (f"{N_rank} images x {wl.n_corners[0]}x{wl.n_corners[1]} corners per GPU, one joint fit of {n_total} images "
                 f"over {world} MI355X (weak scaling of BASELINE.json configs[{cfg_index}])")
    return {"workload": f"{wl.name}: {MODEL_TEXT.get(model, model)}, {where}",
            "images_total": n_total, "images_rank0": N_rank, "corners_per_image": wl.n_points, "free_intrinsics": K,
            "flags": wl.flags, "criteria": list(criteria), "step": "global LM solve (A6-A9) from device-side init",
            "parallelism": f"image-sharded x{world}" if world > 1 else "single GPU", "allreduce": comm}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3", help="headline workload: C1, C3, C5, REF (cameracalibrations_amd/synthetic.py)")
    ap.add_argument("--images", type=int, default=None, help="override the workload's image count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the weak_scaling / c5 / maps / detect blocks")
    ap.add_argument("--allreduce", default=os.environ.get("CCAL_ALLREDUCE", "p2p"), choices=["p2p", "rccl", "torch"],
                    help="N > 1: 'p2p' = direct xGMI row exchange kernels (self-tested at start-up, else every rank falls back "
                         "together), 'rccl' = ncclAllReduce called by the library itself, 'torch' = torch.distributed hook")
    ap.add_argument("--weak", action="store_true",
                    help="N > 1: make the weak-scaling joint fit (every rank its own shard of the named size) the headline "
                         "instead of BASELINE.json configs[3]; the JSON then says scaling = weak")
    ap.add_argument("--strong", action="store_true", help="(default at N > 1; kept for older command lines)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: only the timed solves of the headline (no converged-criteria run, no extras), so "
                         "that counter means are means over the timed region's launches")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous of the ranks only (gloo, no GPU work): checks the launch plumbing, e.g. on a box without a GPU")
    ap.add_argument("--c5-images", type=int, default=None, help="image count of the c5 extra block (default: the named 10000)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: all ranks use cuda:0 and rendezvous over gloo (not a measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process starts the ranks and touches no GPU itself (torch is not even imported here)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if args.dry_run:
        sys.exit(dry_run(args))

    from cameracalibrations_amd import synthetic as syn
    from cameracalibrations_amd.solver import CRITERIA

    job = Job(args)
    world, rank = job.world, job.rank
    strong = world > 1 and not args.weak
    model = syn.CONFIGS[args.workload][2]
    cfg_index = syn.CONFIGS[args.workload][3] if world == 1 else (3 if args.workload in ("C2", "C3") and strong else syn.CONFIGS[args.workload][3])
    if args.workload == "C5":
        cfg_index = 4

    # ---- headline ---------------------------------------------------------------------------------------
    if strong or world == 1:
        wl = syn.named_workload(args.workload, n_images=args.images)
        my_img = shard(wl, world, rank)
        n_total = wl.n_images
    else:      # weak scaling: a different shard of the same camera on every rank
        wl = syn.named_workload(args.workload, n_images=args.images, seed_offset=1000 * rank)
        my_img, n_total = wl.img, wl.n_images * world
    N, P = my_img.shape[0], wl.n_points
    K = free_intrinsics(wl.flags)
    s, img_dev, comm_kind = job.make_solver(wl, my_img, n_total, CRITERIA)
    elapsed, r = job.timed_solves(s, args.steps, args.warmup)
    # like for like with OpenCV's loop (and the CPU baseline): sweeps at new candidate points count; the re-damping
    # sweeps a rejected step costs THIS design are not counted as work, only their time is
    sweeps = r["sweeps"] * args.steps
    eager_elapsed = job.eager_solves(s, args.steps, min(args.warmup, 5))
    sweep_us, n_timed, sweep_us_b2b, ev_s = job.launch_timing(s, args.steps)

    # closed-form init and a full fit, reported separately
    t2 = time.perf_counter()
    s.init_intrinsics()
    s.init_extrinsics()
    s.lm_run()
    job.torch.cuda.synchronize()
    staged_ms = 1e3 * (time.perf_counter() - t2)
    rf = s.result()
    # ... and as ONE pipeline (ccal_fit: every stage enqueued behind the previous one, one wait), median of 20
    s.fit()
    ts = []
    for _ in range(20):
        t2 = time.perf_counter()
        s.fit()
        ts.append(time.perf_counter() - t2)
    fit_ms = 1e3 * float(np.median(ts))
    rp = s.result()
    assert rp["rms"] == rf["rms"] and rp["iters"] == rf["iters"]
    # the opt-in pose initialisation (ccal_set_pose_init 2: the polished pixel-space homography decomposed directly), timed the same way
    s.set_pose_init(2)
    s.fit()
    ts2, te2 = [], []
    for _ in range(20):
        t2 = time.perf_counter()
        s.fit()
        ts2.append(time.perf_counter() - t2)
    r2 = s.result()
    for _ in range(10):
        s.init_intrinsics()
        t2 = time.perf_counter()
        s.init_extrinsics()
        te2.append(time.perf_counter() - t2)
    fast_init = {"full_fit_ms": 1e3 * float(np.median(ts2)), "extrinsics_ms": 1e3 * float(np.median(te2)),
                 "rms_minus_default": r2["rms"] - rp["rms"], "iters": int(r2["iters"]),
                 "what": "ccal_set_pose_init(2): no second homography in the pose stage; same optimum for views whose pose polish "
                         "converges, not OpenCV's trajectory for a view that exhausts its 20 iterations (DESIGN.md section 7) -- opt-in"}
    s.set_pose_init(1)

    # converged criteria (100, 1e-10): iters-to-tol
    rt, tight_ms = None, None
    if not args.timed_only:
        s.set_problem(wl.obj, N, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=100, eps=1e-10)
        s.set_image_points(img_dev)
        s.init_intrinsics()
        s.init_extrinsics()
        s.lm_run()
        job.torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(10):
            s.lm_run()
        job.torch.cuda.synchronize()
        tight_ms = 1e2 * (time.perf_counter() - t3)
        rt = s.result()
    job.barrier()

    out = None
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": float(n_total) * P * sweeps / elapsed,
            "unit": "corner-residuals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # what a step costs beyond its kernels: the step's wall time minus the summed device time of every launch of one
            # solve (HIP events bound to the dispatches, from the event-instrumented pass) = graph replay, inter-kernel
            # boundaries and the host's wait; and the same solves launched kernel by kernel instead of as one graph
            "host_overhead_us_per_step": 1e6 * elapsed / args.steps - job.solve_kernels_us,
            "solve_kernels_us_per_step": job.solve_kernels_us,
            "eager_ms_per_step": 1e3 * eager_elapsed / args.steps,
            "higher_is_better": True,
            # the 1/2/4/8-GPU series is ONE 1000-image problem (configs[2] on one GPU, configs[3] sharded): total work fixed
            "scaling": "strong" if (strong or world == 1) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.share_gpu else "synthetic (REHEARSAL: all ranks share one GPU, not a measurement)",
            "config": config_block(wl, model, cfg_index, N, n_total, world, comm_kind, CRITERIA, strong),
            "lm": lm_block(r, rt, tight_ms),
            "init": {"intrinsics_ms": 1e3 * rf["t_init_intr_s"], "extrinsics_ms": 1e3 * rf["t_init_extr_s"], "full_fit_ms": fit_ms,
                     "full_fit_stage_by_stage_ms": staged_ms, "pose_init_mode": 1, "pose_init_mode_2": fast_init},
            "roofline": roofline_block(N, P, K, sweep_us, n_timed, sweep_us_b2b, 1e3 * ev_s, kernel_name(job.sweep_form),
                                       job.median_launch_us),
            "parity": "device == oracle (CPU restatement of OpenCV 4.10 from the published algorithm) in tests/; no run of OpenCV "
                      "pins the trajectory (iters-to-tol) -- DESIGN.md section 2",
        }
    s.close()
    del s, img_dev

    # ---- extra blocks (never `value`) ---------------------------------------------------------------------
    extras = not (args.no_extras or args.timed_only)
    # Every stage of an extra block that can fail on one rank alone ends in an agreement of all ranks (Job.together):
    # either all of them go on or all of them raise BlockFailed at the same point and skip the block together.
    steps_x, warm_x = max(5, min(args.steps, 50)), 5
    if extras and world > 1 and strong and args.workload in ("C2", "C3"):
        try:
            wlw = job.together(lambda: syn.named_workload(args.workload, n_images=args.images, seed_offset=1000 * rank), "weak-scaling data")
            sw, _, comm_w = job.make_solver(wlw, wlw.img, wlw.n_images * world, CRITERIA)
            el, rw = job.timed_solves(sw, steps_x, warm_x)
            if rank == 0:
                out["weak_scaling"] = {"value": float(wlw.n_images * world) * wlw.n_points * rw["sweeps"] * steps_x / el,
                                       "unit": "corner-residuals/s", "ms_per_step": 1e3 * el / steps_x, "steps": steps_x,
                                       "config": config_block(wlw, model, 2, wlw.n_images, wlw.n_images * world, world, comm_w,
                                                              CRITERIA, False),
                                       "lm": lm_block(rw)}
            sw.close()
            del sw
        except BlockFailed as exc:    # (raised by every rank together)
            if rank == 0:
                out["weak_scaling"] = {"error": str(exc)}
    if extras and args.workload != "C5":
        try:
            wl5 = job.together(lambda: syn.named_workload("C5", n_images=args.c5_images), "C5 data")
            my5 = shard(wl5, world, rank)
            s5, _, comm5 = job.make_solver(wl5, my5, wl5.n_images, CRITERIA)
            st5 = max(5, min(args.steps, 20))
            el, r5 = job.timed_solves(s5, st5, 3)
            us5, n5, b2b5, ev5 = job.launch_timing(s5, min(st5, 10))
            if rank == 0:
                out["c5"] = {"value": float(wl5.n_images) * wl5.n_points * r5["sweeps"] * st5 / el, "unit": "corner-residuals/s",
                             "ms_per_step": 1e3 * el / st5, "steps": st5,
                             "config": config_block(wl5, "full", 4, my5.shape[0], wl5.n_images, world, comm5, CRITERIA, True),
                             "lm": lm_block(r5),
                             "roofline": roofline_block(my5.shape[0], wl5.n_points, free_intrinsics(wl5.flags), us5, n5, b2b5,
                                                        1e3 * ev5, kernel_name(job.sweep_form), job.median_launch_us)}
            s5.close()
            del s5
            if world == 1 and args.c5_images is None:
                # the per-GPU share of configs[4] at 8 GPUs (rank 0's 1250 images x 150 corners) on this one GPU, resident: what
                # an 8-GPU run can at best make of a trial at kernel level, reportable while no 8-GPU node is measured
                my8 = shard(wl5, 8, 0)
                s8, _, _ = job.make_solver(wl5, my8, my8.shape[0], CRITERIA)
                el8, r8 = job.timed_solves(s8, st5, 3)
                us8, n8, b2b8, _ = job.launch_timing(s8, min(st5, 10))
                out["c5"]["share_at_8"] = {
                    "images": int(my8.shape[0]), "corners_per_image": wl5.n_points, "avg_launch_us": us8, "median_launch_us": job.median_launch_us,
                    "kernel": kernel_name(job.sweep_form), "launches_timed": n8,
                    "avg_launch_us_back_to_back": b2b8, "ms_per_step": 1e3 * el8 / st5, "sweeps_per_solve": int(r8["sweeps"]),
                    "whole_avg_launch_us": us5, "predicted_kernel_speedup": us5 / us8 if us8 > 0 else None,
                    "what": "k_lm_trial on rank 0's 1250-image shard as its own fit on one GPU (no exchange): whole / share bounds the "
                            "8-GPU scaling of a trial before any exchange cost; north_star asks >= 6x"}
                s8.close()
                del s8
        except BlockFailed as exc:    # (raised by every rank together)
            if rank == 0:
                out["c5"] = {"error": str(exc)}
    if extras and world > 1:
        # Independent calibrations spread over the GPUs (a caller with many cameras: the loop around src/buildcalibrations.jl:12-26):
        # every rank fits its own 32 C1-shaped calibrations per round with one ccal_fit_many, no exchange at all -- the way this path
        # scales with GPUs when there are many small problems instead of one large one (hard part 3).
        try:
            def make_many():
                from cameracalibrations_amd.solver import Solver as _S
                ss = []
                for i in range(32):
                    wli = syn.named_workload("C1", seed_offset=100 * rank + i)
                    si = _S(job.local_rank)
                    si.set_problem(wli.obj, wli.n_images, wli.image_size, aspect=wli.aspect, flags=wli.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
                    si.set_image_points(wli.img)
                    ss.append(si)
                for _ in range(3):
                    cc_fit_many(ss)
                return ss, wli.n_images * wli.n_points
            from cameracalibrations_amd import fit_many as cc_fit_many
            ss, corners_per_fit = job.together(make_many, "independent fits: set-up")
            rounds_i = 20
            job.barrier()
            t0 = time.perf_counter()
            ok_i, why_i, sweeps_i = True, "", 0
            try:
                for _ in range(rounds_i):
                    cc_fit_many(ss)
                sweeps_i = sum(int(si.result()["sweeps"]) for si in ss)
            except Exception as exc:      # noqa: BLE001
                ok_i, why_i = False, str(exc)
            job.barrier()
            dt_i = job.max_over_ranks(time.perf_counter() - t0)
            if not job.all_agree(ok_i):
                raise BlockFailed(why_i or "independent fits: a peer rank failed")
            tot_sweeps = job.max_over_ranks(float(sweeps_i))      # (same shapes on every rank: the largest count stands for all)
            for si in ss:
                si.close()
            if rank == 0:
                out["independent_fits"] = {
                    "what": "every rank fits its own 32 calibrations of the C1 shape (100 x 9x6, pinhole) per round with one ccal_fit_many (full fit: "
                            "initialisation + LM, corners resident), no exchange between the ranks; barrier + max over ranks around 20 rounds",
                    "fits_per_s": world * 32 * rounds_i / dt_i, "us_per_fit_whole_job": 1e6 * dt_i / (world * 32 * rounds_i),
                    "corner_residuals_per_s": world * rounds_i * corners_per_fit * tot_sweeps / dt_i, "ranks": world, "scaling": "weak"}
        except BlockFailed as exc:    # (raised by every rank together)
            if rank == 0:
                out["independent_fits"] = {"error": str(exc)}
    if extras and world > 1 and strong:
        # The same sharded fit over the other two transports (the headline takes --allreduce, by default the direct row
        # exchange): ncclAllReduce called by the library (RCCL; refuses ranks that share a device, so a rehearsal skips it)
        # and the torch.distributed hook.  Last block, under a deadline: a transport that hangs costs this block only.
        kinds = ["torch"] if args.share_gpu else ["rccl", "torch"]
        tb = {}
        with Deadline(120.0, rank, lambda: dict(out, transports=dict(tb, error="abandoned at its deadline")) if rank == 0 else None):
            for kind in kinds:
                if kind == args.allreduce:
                    continue
                try:
                    st, _, comm_t = job.make_solver(wl, my_img, n_total, CRITERIA, want=kind)
                    el, rk = job.timed_solves(st, steps_x, warm_x)
                    tb[kind] = {"transport": comm_t, "ms_per_step": 1e3 * el / steps_x, "steps": steps_x,
                                "rms_px": float(rk["rms"]), "iters": int(rk["iters"]), "sweeps_per_solve": int(rk["sweeps"])}
                    st.close()
                    del st
                except BlockFailed as exc:    # (raised by every rank together)
                    tb[kind] = {"error": str(exc)}
        if rank == 0:
            out["transports"] = tb
    if extras and world == 1:
        try:
            out["c1"] = small_block(job, "C1", None, CRITERIA, with_cpu=not args.no_cpu_baseline)
        except Exception as exc:      # noqa: BLE001
            out["c1"] = {"error": str(exc)}
        try:
            out["ref_example"] = ref_example_block(with_cpu=not args.no_cpu_baseline)
        except Exception as exc:      # noqa: BLE001
            out["ref_example"] = {"error": str(exc)}
        try:
            out["many_fits"] = many_block()
        except Exception as exc:      # noqa: BLE001
            out["many_fits"] = {"error": str(exc)}
        try:
            out["maps"] = maps_block(wl, rf, with_cpu=not args.no_cpu_baseline)
        except Exception as exc:      # noqa: BLE001
            out["maps"] = {"error": str(exc)}
        try:
            out["detect"] = detect_block(with_cpu=not args.no_cpu_baseline)
        except Exception as exc:      # noqa: BLE001
            out["detect"] = {"error": str(exc)}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, CRITERIA)
            ocv = opencv_baseline(wl, CRITERIA)
            if ocv is not None:
                out["cpu_baseline"]["opencv"] = ocv
        print(json.dumps(out))
    job.close()


if __name__ == "__main__":
    main()
