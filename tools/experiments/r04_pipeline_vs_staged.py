import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from cameracalibrations_amd import synthetic as syn
from cameracalibrations_amd.solver import Solver
for name, n in (("C3", None), ("C5", 2000), ("C5", None), ("C1", None)):
    wl = syn.named_workload(name, n_images=n)
    img = torch.from_numpy(np.ascontiguousarray(wl.img)).to("cuda:0")
    with Solver(0) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=30, eps=1e-3)
        s.set_image_points(img)
        s.init_intrinsics(); s.init_extrinsics()
        i_st, rt_st = s.get_init()
        s.lm_run(); r_st = s.result()
        s.fit()
        i_pl, rt_pl = s.get_init(); r_pl = s.result()
    print(name, wl.n_images, "intr0 equal", np.array_equal(i_st, i_pl), np.abs(i_st - i_pl).max(), "ext0 equal", np.array_equal(rt_st, rt_pl), np.abs(rt_st - rt_pl).max(),
          "n views differing", int((np.abs(rt_st - rt_pl).max(1) > 0).sum()), "rms", r_st["rms"] - r_pl["rms"])
