"""Soak of the direct row exchange with ranks sharing ONE GPU (rendezvous over gloo): every rank repeats the
same sharded fit K times and requires bit-identical results every time (a lost, stale or torn row would change
the sums).  Launch: python -m torch.distributed.run --nproc-per-node R tools/p2p_soak.py --solves K"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.distributed import allgather_bytes_via_torch, shard_range  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--solves", type=int, default=20000)
ap.add_argument("--images", type=int, default=333)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
wl = syn.named_workload("C2", n_images=args.images)
lo, hi = shard_range(args.images, world, rank)
with Solver(0) as s:
    s.set_problem(wl.obj, hi - lo, wl.image_size, aspect=wl.aspect, flags=wl.flags)
    s.set_image_points(wl.img[lo:hi])
    s.init_p2p(world, rank, args.images, allgather_bytes_via_torch())
    dist.barrier()
    s.init_intrinsics()
    s.init_extrinsics()
    s.lm_run()
    first = s.result()
    bad = 0
    for i in range(args.solves):
        if i % 1000 == 999:          # the init stages exchange rows too (8 doubles)
            s.init_intrinsics()
            s.init_extrinsics()
        s.lm_run()
        r = s.result()
        if r["rms"] != first["rms"] or not np.array_equal(r["intr"], first["intr"]) or not np.array_equal(r["rvecs"], first["rvecs"]):
            bad += 1
        if i % 10000 == 9999 and rank == 0:
            print(f"{i + 1} solves, {bad} deviating", flush=True)
    dist.barrier()
print(f"rank {rank}: {args.solves} solves of {args.images} images over {world} ranks, rms {first['rms']:.9f}, "
      f"{first['iters']} iterations, deviating results: {bad}", flush=True)
dist.destroy_process_group()
sys.exit(1 if bad else 0)
