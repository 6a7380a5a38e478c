#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_rectify.py tests/test_detect.py -m gpu -x -q -k "maps or errors or rectif or full_cal or many" 2>&1 | tail -3
python - <<'PY'
import sys, json, time
sys.path.insert(0, ".")
import bench, numpy as np
from cameracalibrations_amd import synthetic as syn
from cameracalibrations_amd.solver import Solver, CRITERIA
wl = syn.named_workload("C3")
with Solver(0) as s:
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
    s.set_image_points(wl.img)
    st, rf = s.calibrate()
for rep in range(3):
    m = bench.maps_block(wl, rf, with_cpu=False)
    print({k: (round(v) if isinstance(v, float) and v > 100 else v) for k, v in m.items() if k not in ("what", "errors")})
PY
