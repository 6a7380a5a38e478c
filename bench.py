#!/usr/bin/env python3
"""bench.py -- corner-residuals/sec of the device-side calibrateCamera LM loop (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[2] -- synthetic pinhole + k1,k2,p1,p2,
1000 images x 9x6 corners per GPU (SURVEY.md 8d recipe, cameracalibrations_amd/synthetic.py),
observed corners resident in HBM before the timed region.  With N GPUs every rank holds its
own 1000-image shard of ONE joint fit (shared intrinsics): weak scaling, one RCCL all-reduce of
the 68-double reduced system per LM trial.

One step = one complete global LM solve from the closed-form initial
values (rows A4/A5 run once on the device before timing and are reported separately, as
SURVEY.md 8d defines the metric).  value = corner-residual evaluations (each with its
2x(6+K) Jacobian block accumulated) per second over the whole job.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def algorithmic_bytes_per_sweep(N, P):
    """SURVEY.md 8d contract figure: 8 B corner read + 96/P B pose read+write per corner-residual."""
    return N * P * (8.0 + 96.0 / P)


def pmc_traffic_bytes(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (tools/pmc.sh: FETCH_SIZE and WRITE_SIZE in separate passes, both calibrated on a
    known byte count with this code's access width).  None if the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic_summary.json")
    try:
        return float(json.load(open(path))[kernel]["hbm_bytes_per_launch"])
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
                     f"oracle Schur mode, LM loop only; rms {r['rms']:.6f} px, iters {r['iters']}"}
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
        import numpy as np
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--images", type=int, default=None, help="override images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allreduce", default=os.environ.get("CCAL_ALLREDUCE", "p2p"), choices=["p2p", "rccl", "torch"],
                    help="N > 1: 'p2p' = direct xGMI row exchange kernels (self-tested at start-up, else falls back), "
                         "'rccl' = ncclAllReduce called by the library itself, 'torch' = torch.distributed hook")
    ap.add_argument("--strong", action="store_true",
                    help="N > 1: shard ONE problem of the named size over the ranks (BASELINE.json configs[3]) instead of "
                         "giving every rank its own shard of that size; the JSON then says scaling = strong")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: only the timed solves (skip the converged-criteria run, so that counter means "
                         "are means over the timed region's launches)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: all ranks use cuda:0 and rendezvous over gloo (not a measurement)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from cameracalibrations_amd import synthetic as syn
    from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, broadcast_id_via_torch
    from cameracalibrations_amd.solver import CRITERIA, Solver

    world = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        assert int(os.environ.get("WORLD_SIZE", "1")) == world, "launch with torch.distributed.run --nproc-per-node N"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)
    cdev = "cpu" if (world > 1 and args.share_gpu) else dev          # where torch.distributed tensors live

    # per-rank shard of the joint problem (different images on every rank, same camera)
    strong = bool(args.strong and world > 1)
    if strong:
        from cameracalibrations_amd.distributed import shard_range
        wl = syn.named_workload(args.workload, n_images=args.images)
        lo, hi = shard_range(wl.n_images, world, rank)
        n_total, my_img = wl.n_images, wl.img[lo:hi]
    else:
        wl = syn.named_workload(args.workload, n_images=args.images, seed_offset=1000 * rank)
        n_total, my_img = wl.n_images * world, wl.img
    N, P = my_img.shape[0], wl.n_points
    img_dev = torch.from_numpy(np.ascontiguousarray(my_img)).to(dev)   # inputs resident in HBM
    torch.cuda.synchronize()
    # the context's own non-blocking stream (not torch's legacy default stream, which serialises with every
    # blocking stream of the process): the timed region is bracketed by device-wide synchronisation anyway
    s = Solver(local_rank)
    s.set_problem(wl.obj, N, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=CRITERIA[0], eps=CRITERIA[1])
    s.set_image_points(img_dev)

    def attach_comm(sol):
        if world == 1:
            return "none"

        def all_agree(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())

        if args.allreduce == "p2p":
            why = ""
            try:
                sol.init_p2p(world, rank, n_total, allgather_bytes_via_torch())
                ok = True
            except Exception as exc:      # noqa: BLE001
                ok, why = False, str(exc)
            if all_agree(ok):
                dist.barrier()            # every inbox is mapped before the first row is sent
                try:
                    sol.p2p_selftest(16)
                    # ... and the real thing: init (8-double exchange) + two sharded solves (fused exchange, graph replay)
                    sol.init_intrinsics()
                    sol.init_extrinsics()
                    sol.lm_run()
                    sol.lm_run()
                except Exception as exc:  # noqa: BLE001
                    ok, why = False, str(exc)
                if all_agree(ok):
                    return "p2p (direct xGMI row exchange kernels, self-tested)"
            print(f"[bench] rank {rank}: direct exchange unavailable ({why or 'a peer failed'}); using RCCL", file=sys.stderr)
        if args.allreduce in ("p2p", "rccl"):
            try:
                sol.init_rccl(world, rank, n_total, broadcast_id_via_torch())
                return "rccl (ncclAllReduce inside the library)"
            except Exception as exc:      # noqa: BLE001 - fall back to the torch hook on every rank
                print(f"[bench] rank {rank}: native RCCL unavailable ({exc}); using the torch hook", file=sys.stderr)
        sol.set_allreduce(TorchAllReduce(), world, rank, n_total)
        return "torch.distributed all_reduce hook"

    comm_kind = attach_comm(s)
    s.init_intrinsics()
    s.init_extrinsics()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        s.lm_run()
    # --- timed region: exactly K steps -------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s.lm_run()
    barrier()
    elapsed = time.perf_counter() - t0
    r = s.result()
    sweeps = (r["sweeps"] + r["reschur"]) * args.steps   # deterministic: every step repeats the same solve
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --- sweep-kernel duration, HIP events on the launch stream, same steps re-run -------
    s.set_sweep_timing(True)
    us, n_timed, t1 = 0.0, 0, time.perf_counter()
    for _ in range(args.steps):
        s.lm_run()
        rr = s.result()
        us += rr["sweep_us"] * rr["timed_sweeps"]
        n_timed += rr["timed_sweeps"]
    torch.cuda.synchronize()
    elapsed_events = time.perf_counter() - t1
    s.set_sweep_timing(False)
    sweep_us = us / max(n_timed, 1)
    sweep_us_b2b = s.time_sweep(200)

    # --- separately reported: closed-form init and a full fit ---------------------------
    t2 = time.perf_counter()
    s.init_intrinsics()
    s.init_extrinsics()
    s.lm_run()
    torch.cuda.synchronize()
    fit_ms = 1e3 * (time.perf_counter() - t2)
    rf = s.result()

    # --- converged criteria (100, 1e-10): iters-to-tol -----------------------------------
    if args.timed_only:
        rt, tight_ms = dict(iters=None, sweeps=0, reschur=0, rms=None), None
    else:
        s.set_problem(wl.obj, N, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=100, eps=1e-10)
        s.set_image_points(img_dev)
        s.init_intrinsics()
        s.init_extrinsics()
        s.lm_run()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(10):
            s.lm_run()
        torch.cuda.synchronize()
        tight_ms = 1e2 * (time.perf_counter() - t3)
        rt = s.result()

    if rank == 0:
        units = float(n_total) * P * sweeps
        alg_bytes = algorithmic_bytes_per_sweep(N, P)
        achieved = alg_bytes / (sweep_us * 1e-6) / 1e9 if sweep_us > 0 else 0.0
        out = {
            "metric": "corner-residuals/sec + LM iters-to-tol, 1000 imgs x 9x6 corners, 1/2/4/8 GPU",
            "value": units / elapsed,
            "unit": "corner-residuals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if not args.share_gpu else "synthetic (REHEARSAL: all ranks share one GPU, not a measurement)",
            "config": {"workload": f"{wl.name}: synthetic pinhole + k1,k2,p1,p2, {N} images x "
                                   f"{wl.n_corners[0]}x{wl.n_corners[1]} corners per GPU (BASELINE.json configs[2]), "
                                   f"one joint fit over {world} GPU(s)",
                       "images_per_gpu": N, "corners_per_image": P, "free_intrinsics": 8, "flags": wl.flags,
                       "criteria": list(CRITERIA), "step": "global LM solve (A6-A9) from device-side init",
                       "parallelism": f"image-sharded x{world}" if world > 1 else "single GPU",
                       "allreduce": comm_kind},
            "lm": {"iters_to_tol_1e-3": r["iters"], "sweeps_per_solve": r["sweeps"] + r["reschur"], "rejected": r["rejected"],
                   "rms_px": r["rms"], "iters_to_tol_1e-10": rt["iters"], "sweeps_1e-10": rt["sweeps"] + rt["reschur"],
                   "rms_px_1e-10": rt["rms"], "ms_per_solve_1e-10": tight_ms},
            "init": {"intrinsics_ms": 1e3 * rf["t_init_intr_s"], "extrinsics_ms": 1e3 * rf["t_init_extr_s"],
                     "full_fit_ms": fit_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_bytes("ccal::k_lm_trial") if (N, P) == (1000, 54) else None,
                         "kernel": "ccal::k_lm_trial<false>", "algorithmic_bytes_per_launch": alg_bytes,
                         # SURVEY.md 8d diagnostic: the same sweep with the per-image blocks kept in HBM (this design)
                         "bytes_per_launch_with_blocks_in_hbm": alg_bytes + N * 2 * 8 * (27 + 6 * 8) if P == 54 else None,
                         "avg_launch_us": sweep_us, "launches_timed": n_timed,
                         "avg_launch_us_back_to_back": sweep_us_b2b,
                         "timed_region_with_events_ms_per_step": 1e3 * elapsed_events / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, CRITERIA)
            ocv = opencv_baseline(wl, CRITERIA)
            if ocv is not None:
                out["cpu_baseline"]["opencv"] = ocv
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
