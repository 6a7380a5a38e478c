"""Diagnostic: where one sweep wave spends its cycles (s_memtime stamps).  Builds a SEPARATE
library with -DCCAL_STAMPS (never the shipped one) and prints per-phase shader cycles."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "cameracalibrations_amd", "csrc")
OUT = "/tmp/libccal_hip_stamps.so"
srcs = [os.path.join(CSRC, f) for f in ("ccal_lm.hip", "ccal_init.hip", "ccal_maps.hip", "ccal_comm.hip", "ccal_api.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-DCCAL_STAMPS", "-o", OUT, *srcs, "-ldl"])
from cameracalibrations_amd import _ffi as F  # noqa: E402
F.LIB_PATH = OUT
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

L = F.lib()
L.ccal_debug_stamps.restype = C.c_int
wl = syn.named_workload("C3")
names = ["entry->done-check", "setup", "prologue (backsubst)", "rodrigues", "point_rows", "tile store+sync",
         "mfma gram+sync", "Gs sync", "chol+fwd+sync", "block store", "red + partial"]
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
    s.set_image_points(wl.img)
    os.environ["CCAL_NO_GRAPH"] = "1"
    for rep in range(3):
        s.calibrate()
        buf = (C.c_ulonglong * 32)()
        L.ccal_debug_stamps(buf)
        t = np.array(buf[:11], dtype=np.int64)
        d = np.diff(t)
        print("total cycles", t[10] - t[0], "(%.2f us at 2.4 GHz)" % ((t[10] - t[0]) / 2400.0))
        for n, v in zip(names[1:], d):
            print(f"   {n:24s} {v:7d}")
        t2 = np.array(buf[16:21], dtype=np.int64)
        print("solve kernel (thread 0):", dict(zip(["done-check", "state+partials+reduce", "lm_advance", "write-back"],
                                                   np.diff(t2).tolist())), "total", t2[4] - t2[0])
