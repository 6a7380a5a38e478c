"""Per-solve cost of the multi-GPU code path at world_size 1 (one GPU box): graph path vs
reduce -> all-reduce -> solve with the torch hook vs the library's own RCCL communicator vs the
direct exchange kernels (one launch between trials, graph-captured)."""
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, broadcast_id_via_torch  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
wl = syn.named_workload("C3")
img = torch.from_numpy(wl.img).cuda()


def run(kind):
    with Solver(0, stream=torch.cuda.current_stream().cuda_stream) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
        s.set_image_points(img)
        if kind == "torch":
            s.set_allreduce(TorchAllReduce(), 1, 0, wl.n_images)
        elif kind == "p2p":
            s.init_p2p(1, 0, wl.n_images, allgather_bytes_via_torch())
        elif kind == "rccl":
            s.init_rccl(1, 0, wl.n_images, broadcast_id_via_torch())
        s.init_intrinsics()
        s.init_extrinsics()
        for _ in range(20):
            s.lm_run()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(200):
            s.lm_run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 200
        r = s.result()
        print(f"{kind:6s}: {1e6 * dt:8.1f} us per solve ({r['sweeps']} sweeps, {1e6 * dt / r['sweeps']:.1f} us per trial)")


for k in ("graph", "p2p", "rccl", "torch"):
    run(k)
dist.destroy_process_group()
