"""Diagnostic: when every workgroup of one k_lm_trial launch starts, finishes its prologue and exits (chip-wide 100 MHz
counter, -DCCAL_STAMPS build), and on which XCD / CU it ran.  usage: wg_timeline.py [--images N]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "cameracalibrations_amd", "csrc")
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, nargs="*", default=[1100])
a = ap.parse_args()
OUT = "/tmp/libccal_hip_wgtimes.so"
srcs = [os.path.join(CSRC, f) for f in ("ccal_lm.hip", "ccal_init.hip", "ccal_maps.hip", "ccal_comm.hip", "ccal_p2p.hip", "ccal_detect.hip", "ccal_api.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                       "-DCCAL_STAMPS", "-o", OUT, *srcs, "-ldl"])
from cameracalibrations_amd import _ffi as F  # noqa: E402
F.LIB_PATH = OUT
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

L = F.lib()
L.ccal_debug_wg_times.restype = C.c_int
os.environ["CCAL_NO_GRAPH"] = "1"
res = {}
for n in a.images:
    wl = syn.named_workload("C3", n_images=n)
    with Solver(0) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
        s.set_image_points(wl.img)
        for rep in range(3):
            s.calibrate()
        buf = (C.c_ulonglong * 4096)()
        L.ccal_debug_wg_times(buf)
    g = (n + 3) // 4
    t = np.array(buf[:], dtype=np.uint64).reshape(1024, 4)[:min(g, 1024)]
    t0 = int(t[:, 0].min())
    start = (t[:, 0].astype(np.int64) - t0) * 10      # ns
    pro = (t[:, 1].astype(np.int64) - t0) * 10
    end = (t[:, 2].astype(np.int64) - t0) * 10
    hw = t[:, 3] & np.uint64(0xffffffff)
    xcc = (t[:, 3] >> np.uint64(32)) & np.uint64(0xf)
    cu = (hw >> np.uint64(8)) & np.uint64(0xf)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    place = xcc.astype(np.int64) * 4096 + se.astype(np.int64) * 256 + sh.astype(np.int64) * 16 + cu.astype(np.int64)
    uniq, cnt = np.unique(place, return_counts=True)
    dbl = np.isin(place, uniq[cnt > 1])
    r = {"workgroups": int(len(t)), "start_ns": {"p50": int(np.median(start)), "p90": int(np.percentile(start, 90)), "max": int(start.max())},
         "prologue_ns": {"p50": int(np.median(pro - start)), "max": int((pro - start).max())},
         "sweep_ns": {"p50": int(np.median(end - pro)), "p90": int(np.percentile(end - pro, 90)), "max": int((end - pro).max())},
         "life_ns": {"p50": int(np.median(end - start)), "max": int((end - start).max())},
         "last_exit_ns": int(end.max()), "distinct_cus": int(len(uniq)), "workgroups_sharing_a_cu": int(dbl.sum()),
         "sweep_ns_alone_p50": int(np.median((end - pro)[~dbl])) if (~dbl).any() else None,
         "sweep_ns_shared_p50": int(np.median((end - pro)[dbl])) if dbl.any() else None,
         "life_ns_shared_max": int((end - start)[dbl].max()) if dbl.any() else None,
         "workgroup0": {"start": int(start[0]), "prologue_done": int(pro[0]), "exit": int(end[0])}}
    oe = np.argsort(end)
    r["last_exits"] = [(int(i), int(end[i])) for i in oe[-6:]]          # (workgroup, exit time): who finishes last
    r["exit_ns"] = {"p50": int(np.median(end)), "p90": int(np.percentile(end, 90)), "p99": int(np.percentile(end, 99))}
    r["prologue_done_ns"] = {"p50": int(np.median(pro)), "max_other_than_0": int(pro[1:].max()) if len(pro) > 1 else None}
    by_xcc = {int(x): int((xcc == x).sum()) for x in np.unique(xcc)}
    r["workgroups_per_xcd"] = by_xcc
    o = np.argsort(start)
    r["start_ns_sorted_every_16th"] = [int(v) for v in start[o][::16]]
    res[str(n)] = r
    print(n, json.dumps(r))
json.dump({"what": "per-workgroup timeline of the last sweeping k_lm_trial launch of a C3-shaped solve (thread 0 of each workgroup, s_memrealtime = "
           "chip-wide 100 MHz counter, 10 ns resolution; diagnostic build with stamps)", "by_images": res},
          open(os.path.join(ROOT, "gpurun_out", "wg_timeline.json"), "w"), indent=1)
