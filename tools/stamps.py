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
BLOCK = int(os.environ.get("CCAL_STAMP_BLOCK", "7"))
MEM = os.environ.get("CCAL_STAMPS_MEM", "0") == "1"
OUT = "/tmp/libccal_hip_stamps_%d_%d.so" % (BLOCK, MEM)
srcs = [os.path.join(CSRC, f) for f in ("ccal_lm.hip", "ccal_init.hip", "ccal_maps.hip", "ccal_comm.hip", "ccal_p2p.hip", "ccal_detect.hip", "ccal_api.hip")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                       "-DCCAL_STAMPS", *(["-DCCAL_STAMPS_MEM"] if MEM else []), "-DCCAL_STAMP_BLOCK=%d" % BLOCK, "-o", OUT, *srcs, "-ldl"])
from cameracalibrations_amd import _ffi as F  # noqa: E402
F.LIB_PATH = OUT
from cameracalibrations_amd import synthetic as syn  # noqa: E402
from cameracalibrations_amd.solver import Solver  # noqa: E402

L = F.lib()
L.ccal_debug_stamps.restype = C.c_int
wl = syn.named_workload(os.environ.get("CCAL_STAMP_WL", "C3"))
names = ["entry", "state + accumulators + advance", "head -> pose (backsubst)", "rotation R,K", "point_rows",
         "tile store", "mfma gram", "Gram -> LDS", "chol6 + Y", "outputs + WG sum + atomics", "U + block store"]
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
    s.set_image_points(wl.img)
    os.environ["CCAL_NO_GRAPH"] = "1"
    s.calibrate()       # (the first solve's stamps mix launches of two solves' worth of set-up: negative differences; not printed)
    for rep in range(3):
        s.calibrate()
        buf = (C.c_ulonglong * 32)()
        L.ccal_debug_stamps(buf)
        t = np.array(buf[:11], dtype=np.int64)
        d = np.diff(t)
        rt = 0
        print("total cycles", t[10] - t[0], "(%.2f us at 2.4 GHz)" % ((t[10] - t[0]) / 2400.0))
        for n, v in zip(names[1:], d):
            print(f"   {n:24s} {v:7d}")
        rt = int(buf[31]) - int(buf[30])
        if rt > 0:
            print("   shader clock over this wave's lifetime: %.0f MHz (s_memtime / s_memrealtime x 100 MHz)" % (100.0 * (int(buf[10]) - int(buf[0])) / rt))
        print("   outputs phase: Schur dots", int(buf[21]) - int(buf[8]), "| sums to LDS", int(buf[22]) - int(buf[21]), "| barrier", int(buf[23]) - int(buf[22]),
              "| workgroup sum + split + atomics issued", int(buf[9]) - int(buf[23]))
        print("   entry -> kernel arguments in SGPRs:", int(buf[13]) - int(buf[0]), "cycles")
        if MEM:
            print("   dependent single-word round trips after that: corner stream (read-only)", int(buf[14]) - int(buf[13]),
                  "| accumulators (memory-side atomics)", int(buf[15]) - int(buf[14]), "| previous state (plain stores of WG 0)", int(buf[20]) - int(buf[15]))
        t = np.array(buf[:13], dtype=np.int64)
        print("   prologue: loads -> barrier", t[11] - t[0], "decode (+exchange)", t[12] - t[11], "advance+sync", t[1] - t[12])
        a = np.array(buf[16:20], dtype=np.int64)
        print("   advance: state machine", a[0] - t[12], "assemble S", a[1] - a[0], "chol9", a[2] - a[1], "solves", a[3] - a[2],
              "tail", t[1] - a[3])

    # machine-readable summary of the last repetition (copied to profiles/r02_wave_stamps.json for block 7)
    import json
    phases = dict(zip(names[1:], [int(v) for v in d]))
    clock = 100.0 * (int(buf[10]) - int(buf[0])) / rt if rt > 0 else None
    json.dump({"workload": wl.name, "N": wl.n_images, "P": wl.n_points, "block": BLOCK, "wave_cycles": int(buf[10]) - int(buf[0]),
               "clock_mhz": clock, "phases_cycles": phases,
               "prologue_cycles": {"loads_to_barrier": int(t[11] - t[0]), "decode": int(t[12] - t[11]), "advance_and_sync": int(t[1] - t[12])},
               "what": "s_memtime stamps of wave 0 of one workgroup in the last sweeping launch of a solve (diagnostic build, "
                       "-DCCAL_STAMPS: the stamps and their scheduling barriers add roughly 5-10 % to the wave's cycles)"},
              open(os.path.join(ROOT, "gpurun_out", "wave_stamps_block%d.json" % BLOCK), "w"), indent=1)
    ib = (C.c_ulonglong * 32)()
    L.ccal_debug_init_stamps.restype = C.c_int
    L.ccal_debug_init_stamps(ib)
    v = [int(x) for x in ib]
    print("k_init_extrinsics (block 7): homography", v[1] - v[0], "cycles; decomposition", v[2] - v[1], "; pose LM", v[3] - v[2],
          "cycles with", v[4], "Jacobian passes at accepted points,", v[5], "at trial points,", v[6], "iterations; homography polish iterations", v[7])
    print("   last Jacobian pass (stamps are approximate: the compiler schedules across them): rows", v[12] - v[11], "| 28 products per lane", v[13] - v[12],
          "| sums over the lanes through LDS (wave_sums_lds)", v[14] - v[13])
    h = v[16:28]
    print("   homography: centroids + scales", h[1] - h[0], "| DLT sums", h[2] - h[1], "| 9x9 factor + inverse iteration", h[3] - h[2], "| compose", h[4] - h[3],
          "| polish", h[5] - h[4], "= first normal equations", h[6] - h[4], "+ iterations; last iteration: 8x8 solve", h[8] - h[7],
          "| normal equations at the trial point", h[9] - h[8], "| gain ratio + lambda", h[10] - h[9], "| acceptance", h[11] - h[10])
    print("   pose LM split: steps (6x6 solve + rotation)", v[8], "| passes at accepted points", v[9], "| passes at trial points", v[10])

    bb = (C.c_uint * (4096 * 4))()
    L.ccal_debug_init_blocks.restype = C.c_int
    L.ccal_debug_init_blocks(bb)
    b = np.array(bb[:], dtype=np.int64).reshape(4096, 4)[:1000]
    print("per-image k_init_extrinsics cycles: min %d median %d p90 %d max %d" % (b[:, 0].min(), np.median(b[:, 0]), np.percentile(b[:, 0], 90), b[:, 0].max()))
    for j, n in ((1, "homography polish iters"), (2, "pose LM iters"), (3, "pose trial-point passes")):
        print("  ", n, np.bincount(b[:, j]).tolist())
    slow = np.argsort(b[:, 0])[-8:]
    print("   slowest images: (cycles, pose iters, trial-point passes)", [(int(b[i, 0]), int(b[i, 2]), int(b[i, 3])) for i in slow])
    print("   rejected pose steps (trial-point passes - iterations): total", int((b[:, 3] - b[:, 2]).sum()), "max per image", int((b[:, 3] - b[:, 2]).max()))
