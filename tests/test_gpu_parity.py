"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes --
through size-independent properties.  Tolerances are stated per assertion.

Parity targets (BASELINE.json north_star): RMS reprojection error within 1e-4 px; intrinsics,
rvec/tvec within a stated float tolerance; identical accepted-iteration counts under the
reference's criteria (30, 1e-3) (reference src/CameraCalibrations.jl:16).
"""
import json
import os

import numpy as np
import pytest

from cameracalibrations_amd import synthetic as syn
from cameracalibrations_amd.solver import Solver, fit_model, reference_flags

from _spawn import free_port, spawn_ranks

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.oracle")

REF = (30, 1e-3)
TIGHT = (100, 1e-10)


def _solver(wl, crit, img=None):
    s = Solver(0)
    s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=crit[0], eps=crit[1])
    s.set_image_points(wl.img if img is None else img)
    return s


def _oracle(wl, crit, **kw):
    return orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=crit[0],
                         eps=crit[1], solver="schur", **kw)


def _oracle_init(wl):
    K4, _ = orc.init_intrinsics(wl.obj, wl.img, *wl.image_size, aspect=wl.aspect if wl.flags & 2 else 0.0)
    intr0 = np.r_[K4, np.zeros(5)]
    rt0 = np.array([np.r_[orc.find_extrinsics(wl.obj, wl.img[i], intr0)] for i in range(wl.n_images)])
    return intr0, rt0


CASES = [("C1", 16), ("C2", 37), ("REF", 9), ("C5", 7), ("C2", 1), ("C1", 3)]
FULL_CASES = CASES + [("C1", 100)]      # BASELINE.json configs[1] at its stated size: 100 images x 9x6, pinhole


@pytest.mark.parametrize("name,n", CASES)
def test_per_image_gram_blocks(name, n):
    """Row A6+A7: per-image [Je|Ji|e]^T[Je|Ji|e] (JtJ blocks, Jt e, ||e||^2), <= 1e-11 scaled."""
    wl = syn.named_workload(name, n_images=n)
    with _solver(wl, REF) as s:
        G = s.image_grams(wl.intr_true, np.c_[wl.rvecs_true, wl.tvecs_true])
    asp = wl.aspect if wl.flags & 2 else 0.0
    for i in range(n):
        Go = orc.image_gram(wl.obj, wl.img[i], wl.rvecs_true[i], wl.tvecs_true[i], wl.intr_true, asp)
        scale = np.sqrt(np.outer(np.diag(Go), np.diag(Go))) + 1e-300
        assert np.abs(G[i] - Go).max() / 1.0 >= 0          # finite
        assert (np.abs(G[i] - Go) / scale).max() < 1e-11
        assert np.array_equal(G[i], G[i].T)                  # MFMA Gram is exactly symmetric


@pytest.mark.parametrize("name,n", CASES)
@pytest.mark.parametrize("crit", [REF, TIGHT])
def test_lm_trajectory_from_oracle_init(name, n, crit):
    """Rows A8+A9: same start -> same CvLevMarq trajectory (accept/reject, lambda, iteration
    count) and parameters to 1e-9 relative; the Schur solve equals the oracle's step."""
    wl = syn.named_workload(name, n_images=n)
    intr0, rt0 = _oracle_init(wl)
    ro = _oracle(wl, crit, init_intr=intr0, init_rt=rt0)
    with _solver(wl, crit) as s:
        s.set_intrinsics(intr0)
        s.set_extrinsics(rt0)
        s.lm_run()
        r, tr = s.result(), [t for t in s.trace() if t["accepted"] != 2]
    if crit == REF:
        assert r["iters"] == ro["iters"]
        assert [(t["accepted"], t["lambda_lg10"]) for t in tr] == [(t["accepted"], t["lambda_lg10"]) for t in ro["trace"]]
        assert np.allclose([t["err"] for t in tr], [t["err"] for t in ro["trace"]], rtol=1e-10)
        tol = 1e-9
    else:
        tol = 1e-6      # tight runs end on a rounding-level plateau where accept/reject is noise
    assert abs(r["rms"] - ro["rms"]) < 1e-9
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < tol
    assert np.abs(r["intr"][4:] - ro["intr"][4:]).max() < tol * 10
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < tol * 10
    assert np.abs(r["tvecs"] - ro["tvecs"]).max() < tol * 100


def test_device_accumulators_are_exact():
    """The exact accumulators on their own (ccal_debug_acc_sum): rows added as workgroups add them (integer atomics,
    any order), read back and decoded as a trial's prologue does, against exact rational arithmetic: the result is within 4 ulp of the exact sum of the contributions (each truncated at
    2^-150), whatever their order, magnitudes and cancellation (exact zero when they cancel)."""
    from fractions import Fraction
    rng = np.random.default_rng(11)

    def exact(col):
        t = Fraction(0)
        for x in col:
            f = Fraction(float(x))
            q = f * 2 ** 150                       # truncation towards zero at the accumulators' resolution
            t += Fraction(int(q) if q >= 0 else -int(-q), 2 ** 150)
        return t

    with Solver(0) as s:
        for n_rows, R, lo_exp, hi_exp in ((250, 8, -30, 40), (250, 8, -80, 20), (1024, 8, -140, 57), (37, 2, -60, 0), (512, 64, -100, 50)):
            rows = rng.standard_normal((n_rows, 68)) * np.exp2(rng.integers(lo_exp, hi_exp, (n_rows, 68)))
            rows[:, 5] = 1.0 / 3.0                                           # plain sums
            rows[: n_rows // 2 * 2 : 2, 7] = -rows[1 : n_rows // 2 * 2 : 2, 7]   # cancels to (almost) nothing
            rows[:, 9] = 0.0
            rows[:, 67] = 0.0                                                # the pad column carries flags, no value
            got = s.debug_acc_sum(rows, R)
            for c in range(67):
                t = exact(rows[:, c])
                if t == 0:
                    assert got[c] == 0.0
                    continue
                want = float(t)
                assert abs(Fraction(got[c]) - t) <= 4 * abs(Fraction(np.spacing(want))), (n_rows, R, c, got[c], want)
        # flags: a NaN anywhere makes the error column NaN, an out-of-range value makes it +inf
        rows = rng.standard_normal((9, 68))
        rows[:, 67] = 0.0
        rows[3, 12] = np.nan
        got = s.debug_acc_sum(rows, 8)
        assert np.isnan(got[63])
        rows[3, 12] = 1e40
        got = s.debug_acc_sum(rows, 8)
        assert np.isposinf(got[63])


@pytest.mark.parametrize("name,n", CASES)
@pytest.mark.parametrize("pose_mode", [1, 0])
def test_closed_form_initialisation(name, n, pose_mode):
    """Rows A4+A5: focal lengths from homographies and per-view poses, vs the oracle.
    The device takes the smallest eigenvector by inverse iteration (oracle: Jacobi; they agree to 2e-16,
    profiles/r05_init_parity_probe.txt) and both polish to the same optimum: 1e-8 relative on f, 1e-7 on poses.
    pose_mode 1 = the shipped default (the pose stage's DLT reused from the intrinsics stage), 0 = OpenCV's own sequence."""
    wl = syn.named_workload(name, n_images=n)
    intr0, rt0 = _oracle_init(wl)
    with _solver(wl, REF) as s:
        s.set_pose_init(pose_mode)
        s.init_intrinsics()
        gi, _ = s.get_init(extrinsics=False)
        s.set_intrinsics(intr0)         # same intrinsics for the pose stage
        s.init_extrinsics()
        _, grt = s.get_init()
    assert np.abs(gi[:2] - intr0[:2]).max() / intr0[1] < 1e-8
    assert np.array_equal(gi[2:4], intr0[2:4]) and np.all(gi[4:] == 0)
    assert np.abs(grt[:, :3] - rt0[:, :3]).max() < 1e-7
    assert np.abs(grt[:, 3:] - rt0[:, 3:]).max() < 1e-6


@pytest.mark.parametrize("name,n", FULL_CASES)
@pytest.mark.parametrize("crit", [REF, TIGHT])
def test_full_fit_matches_oracle(name, n, crit):
    """The whole calibrateCamera replacement, device init + device LM, vs the oracle.
    north_star bar: RMS within 1e-4 px; here 1e-8.  Parameters: 1e-6 relative."""
    wl = syn.named_workload(name, n_images=n)
    ro = _oracle(wl, crit)
    with _solver(wl, crit) as s:
        status, r = s.calibrate()
    assert status in (0, 1)
    assert abs(r["rms"] - ro["rms"]) < 1e-8
    if crit == REF:
        assert r["iters"] == ro["iters"]
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6
    assert np.abs(r["intr"][4:] - ro["intr"][4:]).max() < 1e-5
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5
    assert np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-4


def test_golden_fixtures(golden_dir):
    """Committed golden vectors (tests/golden/oracle_golden.json) incl. the reference's own
    corner fixture test/example/corners.json under both reference flag words."""
    gold = json.load(open(os.path.join(golden_dir, "oracle_golden.json")))
    ex = json.load(open(os.path.join(golden_dir, "example_corners.json")))
    for key, g in gold.items():
        if key.startswith("example"):
            obj = syn.board_template(tuple(ex["n_corners"]))
            img = np.array(ex["corners"], dtype=np.float32)
            size, aspect = ex["image_size_wh"], 1.0
        else:
            wl = syn.named_workload(g["workload"], n_images=g["n_images"])
            obj, img, size, aspect = wl.obj, wl.img, wl.image_size, g["aspect"]
        with Solver(0) as s:
            s.set_problem(obj, img.shape[0], size, aspect=aspect, flags=g["flags"], max_iter=g["max_iter"], eps=g["eps"])
            s.set_image_points(img)
            _, r = s.calibrate()
        assert abs(r["rms"] - g["rms"]) < 1e-8, key
        tol = 1e-6 if key.endswith("ref") else 2e-5
        assert np.abs(r["intr"] - np.array(g["intr"])).max() < tol * 1600, key
        assert np.abs(r["rvecs"] - np.array(g["rvecs"])).max() < tol * 10, key
        assert np.abs(r["tvecs"] - np.array(g["tvecs"])).max() < tol * 100, key
        if key.endswith("ref"):
            assert r["iters"] == g["iters"], key


def test_reference_example_via_fit_model(golden_dir):
    """reference test/runtests.jl:68-85 through the drop-in `fit_model` + maps: all four RMS
    errors < 1 on the reference's fixture, rectification identity, result equals the oracle."""
    import cameracalibrations_amd as cc
    ex = json.load(open(os.path.join(golden_dir, "example_corners.json")))
    n_corners = tuple(ex["n_corners"])
    img = np.array(ex["corners"], dtype=np.float64)
    imgpointss = [m for m in img]
    c, eps = cc.fit_corners(imgpointss, n_corners, 1, ex["image_size_wh"], files=ex["files"],
                            rng=np.random.default_rng(11))
    assert eps["n"] == 6
    assert all(eps[k] < 1 for k in ("reprojection", "projection", "distance", "inverse")), eps
    ro = orc.calibrate(syn.board_template(n_corners), img, ex["image_size_wh"], flags=orc.REF_FLAGS_DISTORTION,
                       solver="dense")
    assert abs(eps["reprojection"] - ro["rms"]) < 1e-8
    assert abs(c.f[0] - ro["intr"][0]) < 1e-6 * ro["intr"][0] and c.f[0] == c.f[1]
    assert abs(c.k - ro["intr"][4]) < 1e-7
    # Rectification testset: f(i) == c(i, 1)[[1,2]]
    f = cc.rectification(c, 1)
    i = cc.RowCol(1, 2)
    assert np.array_equal(f(i), c(i, 1)[[0, 1]])
    assert abs(c(i, 1)[2]) < 1e-9
    # forward o inverse = id
    assert np.allclose(c(c(i, 1), 1), i, atol=1e-9)
    # `c(x)` without index needs a file matching r"extrinsic" (src/meta.jl:90-93)
    with pytest.raises(KeyError):
        c(i)
    c.files[2] = "my_extrinsic.png"
    assert np.array_equal(c(i), c(i, 3))


def test_maps_and_errors_match_oracle():
    """Rows A11-A13 on the device vs the oracle's restatement of src/meta.jl."""
    rng = np.random.default_rng(5)
    f, c = np.array([401.4, 399.2]), np.array([189.2, 254.7])
    with Solver(0) as s:
        for k in (0.0, 0.0496, -0.08, 0.4):
            r, t = rng.uniform(-0.5, 0.5, 3), np.array([-1.9, -1.1, 7.4])
            xyz = np.c_[rng.uniform(0, 8, (257, 2)), np.zeros(257)] * 2.5
            fw, fo = s.project(xyz, f, c, k, r, t, 2.5), orc.map_forward(xyz, f, c, k, r, t, 2.5)
            assert np.abs(fw - fo).max() < 1e-10
            rc = rng.uniform(1, 400, (300, 2))
            iv, io = s.unproject(rc, f, c, k, r, t, 2.5), orc.map_inverse(rc, f, c, k, r, t, 2.5)
            assert np.abs(iv - io).max() < 1e-9 * max(1.0, np.abs(io).max())
        wl = syn.named_workload("REF", n_images=5)
        ro = orc.calibrate(wl.obj, wl.img, wl.image_size, flags=wl.flags, solver="schur")
        pts = rng.random((5, 64, 2)) * (np.array(wl.image_size) - 1) + 1
        args = (wl.n_corners, wl.obj.astype(np.float64) * 3.0, wl.img.astype(np.float64), ro["intr"][:2],
                ro["intr"][2:4], ro["intr"][4], ro["rvecs"], ro["tvecs"], 3.0)
        eg, eo = s.errors(*args, inv_pts=pts), orc.errors(*args, inv_pts=pts)
        for key in eo:
            assert abs(eg[key] - eo[key]) < 1e-10 * max(1.0, eo[key]), key


def test_error_metrics_over_many_images_match_oracle():
    """`ccal_errors` sums its per-image partials in image order through LDS in chunks of 2048 images: 4500 images
    (three chunks, the last one partial) of a small board against the oracle -- the sums do not depend on N's
    relation to the chunk."""
    rng = np.random.default_rng(9)
    N, nc = 4500, (4, 3)
    P = nc[0] * nc[1]
    f, c, k, cs = np.array([820.0, 815.0]), np.array([301.5, 399.0]), -0.07, 2.0
    obj = np.stack([np.arange(P) % nc[0], np.arange(P) // nc[0], np.zeros(P)], axis=1).astype(np.float64) * cs
    rv = rng.uniform(-0.4, 0.4, (N, 3))
    tv = np.c_[rng.uniform(-2, 0, (N, 2)), rng.uniform(7, 12, N)]
    with Solver(0) as s:
        img = s.project_many(obj, f, c, k, rv, tv, cs) + rng.normal(0, 0.3, (N, P, 2))
        pts = rng.random((N, 10, 2)) * 500 + 50
        args = (nc, obj, img, f, c, k, rv, tv, cs)
        eg, eo = s.errors(*args, inv_pts=pts), orc.errors(*args, inv_pts=pts)
    for key in eo:
        assert abs(eg[key] - eo[key]) < 1e-10 * max(1.0, eo[key]), key
    assert 0.35 < eo["reprojection"] < 0.5            # the injected 0.3 px per coordinate


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_full_size_properties(name):
    """BASELINE.json full sizes (C3: 1000 x 54; C5: 10000 x 150 on one GPU), checked through
    size-independent properties: ground-truth recovery within the noise floor, noise-level RMS,
    determinism (bit-identical reruns), permutation of images permutes poses only, and an
    oracle spot-check on a subset evaluated at the device's solution."""
    n_images = None if name == "C3" else 10000
    wl = syn.named_workload(name, n_images=n_images)
    with _solver(wl, TIGHT) as s:
        st, r = s.calibrate()
        st2, r2 = s.calibrate()
    assert st in (0, 1)
    for k in ("intr", "rvecs", "tvecs"):
        assert np.array_equal(r[k], r2[k]), k                  # deterministic reductions
    assert r["rms"] == r2["rms"] and r["iters"] == r2["iters"]
    assert abs(r["rms"] - wl.sigma * np.sqrt(2)) < 0.01            # RMS at the injected noise level
    assert np.abs(r["intr"][:4] - wl.intr_true[:4]).max() < 0.5     # px; noise floor of 54k+ corners
    assert np.abs(r["intr"][4:8] - wl.intr_true[4:8]).max() < 5e-3
    assert np.abs(r["rvecs"] - wl.rvecs_true).max() < 5e-3
    # stationarity at the returned point: an oracle Gram on a subset gives J^T e ~ 0 for extrinsics
    asp = wl.aspect if wl.flags & 2 else 0.0
    for i in (0, wl.n_images // 2, wl.n_images - 1):
        G = orc.image_gram(wl.obj, wl.img[i], r["rvecs"][i], r["tvecs"][i], r["intr"], asp)
        grad = G[:6, 15] / np.sqrt(np.diag(G)[:6] * G[15, 15])
        assert np.abs(grad).max() < 1e-6
    # permutation property: reversing the image order reverses the poses, same intrinsics
    with _solver(wl, TIGHT, img=np.ascontiguousarray(wl.img[::-1])) as s:
        _, rp = s.calibrate()
    assert np.abs(rp["intr"][:4] - r["intr"][:4]).max() < 1e-6
    assert np.abs(rp["rvecs"][::-1] - r["rvecs"]).max() < 1e-7
    assert abs(rp["rms"] - r["rms"]) < 1e-10


def test_edge_cases():
    """Error behaviour at the boundary: bad arguments, unsupported flags, non-planar rig,
    call order; minimal sizes (1 image, 4 corners would be rank-deficient -> 2x2 board + more)."""
    from cameracalibrations_amd import _ffi as F
    wl = syn.named_workload("C1", n_images=2)
    with Solver(0) as s:
        with pytest.raises(F.CcalError) as e:
            s.set_problem(wl.obj, 2, wl.image_size, flags=0x4000)          # CALIB_RATIONAL_MODEL
        assert e.value.code == F.E_ARG
        with pytest.raises(F.CcalError) as e:
            s.lm_run()
        assert e.value.code == F.E_STATE
        bent = wl.obj.copy()
        bent[3, 2] = 0.5
        s.set_problem(bent, 2, wl.image_size, flags=wl.flags)
        s.set_image_points(wl.img)
        with pytest.raises(F.CcalError) as e:
            s.init_intrinsics()
        assert e.value.code == F.E_NONPLANAR
        with pytest.raises(F.CcalError) as e:
            s.set_problem(wl.obj, 2, wl.image_size, aspect=1000.0, flags=reference_flags())
        assert e.value.code == F.E_ARG
    # single image, reference flags (focal length observable from one tilted view)
    wl1 = syn.named_workload("REF", n_images=1)
    ro = orc.calibrate(wl1.obj, wl1.img, wl1.image_size, flags=wl1.flags, solver="dense")
    m = fit_model(wl1.image_size, wl1.obj, [wl1.img[0]], wl1.n_corners, True, 1.0)
    assert abs(m.frow - ro["intr"][0]) < 1e-6 * ro["intr"][0] and m.frow == m.fcol
    assert abs(m.k - ro["intr"][4]) < 1e-6
    assert np.abs(m.Rs[0] - ro["rvecs"][0]).max() < 1e-6


def test_allreduce_paths_world1(monkeypatch):
    """The multi-GPU code path (reduce kernel -> all-reduce -> solve kernel, no graph) with one
    rank: (a) torch.distributed hook (nccl == RCCL), (b) the library's own RCCL communicator,
    (c) the direct xGMI exchange kernels.  All must reproduce the graph path bit for bit (the single-GPU fit with one wave per
    image, as image-sharded fits sweep one-pass boards: a split image's Gram is summed in another order)."""
    import torch
    import torch.distributed as dist
    from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, broadcast_id_via_torch
    wl = syn.named_workload("C2", n_images=37)
    monkeypatch.setenv("CCAL_WPI", "1")
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    monkeypatch.delenv("CCAL_WPI")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        hook = TorchAllReduce()
        with Solver(0, stream=torch.cuda.current_stream().cuda_stream) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
            s.set_image_points(torch.from_numpy(wl.img).cuda())
            s.set_allreduce(hook, 1, 0, wl.n_images)
            _, r1 = s.calibrate()
        assert hook.calls >= r1["sweeps"] + 1            # focal least squares + one per LM trial
        with Solver(0, stream=torch.cuda.current_stream().cuda_stream) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
            s.set_image_points(wl.img)
            s.init_rccl(1, 0, wl.n_images, broadcast_id_via_torch())
            _, r2 = s.calibrate()
        with Solver(0) as s:                               # (c) direct exchange kernels, graph-captured
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags)
            s.set_image_points(wl.img)
            s.init_p2p(1, 0, wl.n_images, allgather_bytes_via_torch())
            _, r3 = s.calibrate()
            _, r3b = s.calibrate()                         # graph replay: the epoch advances on the device
            assert np.array_equal(r3["intr"], r3b["intr"]) and r3["rms"] == r3b["rms"]
    finally:
        dist.destroy_process_group()
    for r in (r1, r2):
        assert r["iters"] == base["iters"] and r["rms"] == base["rms"]
        for k in ("intr", "rvecs", "tvecs"):
            assert np.array_equal(r[k], base[k]), k
    # the exchange kernel condenses the partial rows in another (fixed) order: same trajectory, last-bit sums
    assert r3["iters"] == base["iters"] and r3["sweeps"] == base["sweeps"] and abs(r3["rms"] - base["rms"]) < 1e-12
    for k in ("intr", "rvecs", "tvecs"):
        assert np.abs(r3[k] - base[k]).max() < 1e-9 * max(1.0, np.abs(base[k]).max()), k


FLAG_CASES = [
    # (flags, aspect, use_guess)
    (syn.FIX_ASPECT_RATIO | syn.ZERO_TANGENT_DIST | syn.FIX_K2 | syn.FIX_K3, 1.25, False),   # reference word, aspect != 1
    (syn.FIX_PRINCIPAL_POINT | syn.FIX_K3, 1.0, False),
    (syn.FIX_FOCAL_LENGTH | syn.FIX_K2 | syn.FIX_K3 | syn.ZERO_TANGENT_DIST, 1.0, True),
    (syn.FIX_TANGENT_DIST | syn.FIX_K1, 1.0, True),
    (syn.USE_INTRINSIC_GUESS | syn.FIX_ASPECT_RATIO | syn.FIX_K3, 1.0, True),
    (0, 1.0, False),                                                                            # full 9-parameter model
]


@pytest.mark.parametrize("flags,aspect,use_guess", FLAG_CASES)
def test_flag_words_match_oracle(flags, aspect, use_guess):
    """Parameter masks of SURVEY.md A.1 (every supported flag bit), incl. intrinsic guesses
    with non-zero distortion (undistortion in the pose initialisation) -- vs the oracle."""
    wl = syn.make_workload("flags", (9, 6), 14, "full", seed=99, aspect=1.0)
    if flags & syn.FIX_ASPECT_RATIO and not use_guess:
        wl = syn.make_workload("flags", (9, 6), 14, "ref", seed=99, aspect=aspect)
    guess = None
    if use_guess:
        flags |= syn.USE_INTRINSIC_GUESS
        guess = wl.intr_true * np.array([1.01, 0.99, 1.002, 0.998, 0.9, 1.1, 1.0, 1.0, 1.0])
    ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=1e-6,
                       solver="schur", init_intr=guess)
    with Solver(0) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=1e-6)
        s.set_image_points(wl.img)
        st, r = s.calibrate(init_intr=guess)
    assert st in (0, 1)
    assert r["iters"] == ro["iters"]
    assert abs(r["rms"] - ro["rms"]) < 1e-8
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6
    assert np.abs(r["intr"][4:] - ro["intr"][4:]).max() < 1e-5
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5 and np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-4
    # masked parameters did not move
    fixed = []
    if flags & syn.FIX_K1: fixed.append(4)
    if flags & syn.FIX_K2: fixed.append(5)
    if flags & syn.FIX_K3: fixed.append(8)
    if flags & (syn.ZERO_TANGENT_DIST | syn.FIX_TANGENT_DIST): fixed += [6, 7]
    if flags & syn.FIX_PRINCIPAL_POINT: fixed += [2, 3]
    if flags & syn.FIX_FOCAL_LENGTH: fixed += [0, 1]
    start = guess if guess is not None else np.r_[ro["init_intr"]]
    for a in fixed:
        want = 0.0 if (a in (6, 7) and flags & syn.ZERO_TANGENT_DIST) else start[a]
        assert r["intr"][a] == want, a
    if flags & syn.FIX_ASPECT_RATIO:
        asp = (guess[0] / guess[1]) if use_guess else aspect
        assert abs(r["intr"][0] / r["intr"][1] - asp) < 1e-12


def test_non_finite_input_is_reported():
    from cameracalibrations_amd import _ffi as F
    wl = syn.named_workload("C1", n_images=5)
    bad = wl.img.copy()
    bad[2, 7, 0] = np.nan
    with Solver(0) as s:
        s.set_problem(wl.obj, 5, wl.image_size, flags=wl.flags)
        s.set_image_points(bad)
        with pytest.raises(F.CcalError) as e:
            s.calibrate()
        assert e.value.code in (F.E_NONFINITE, F.E_DEGENERATE)


def _two_rank_worker(rank, world, port, n_total, out, transport="torch"):
    import torch
    import torch.distributed as dist
    from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                               # both ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = syn.named_workload("C2", n_images=n_total)
        lo, hi = shard_range(n_total, world, rank)
        with Solver(0, stream=torch.cuda.current_stream().cuda_stream) as s:
            s.set_problem(wl.obj, hi - lo, wl.image_size, aspect=wl.aspect, flags=wl.flags)
            s.set_image_points(torch.from_numpy(wl.img[lo:hi]).cuda())
            if transport == "p2p":                         # ranks sharing a GPU find out and take the one-workgroup exchange
                assert s.init_p2p(world, rank, n_total, allgather_bytes_via_torch()) is True
            elif transport == "p2p-fused":                 # the form real multi-GPU runs take: exchange fused into the prologue
                assert s.init_p2p(world, rank, n_total, allgather_bytes_via_torch(), shared_device=False) is False
            else:
                s.set_allreduce(TorchAllReduce(), world, rank, n_total)
            st, r = s.calibrate()
            if transport.startswith("p2p"):                # graph replay with the peers
                st2, r2 = s.calibrate()
                assert np.array_equal(r["intr"], r2["intr"]) and r["rms"] == r2["rms"]
            dist.barrier()                                 # no rank unmaps its inbox while a peer still sends
        out[rank] = dict(status=st, lo=lo, hi=hi, intr=r["intr"].tolist(), rvecs=r["rvecs"].tolist(),
                         tvecs=r["tvecs"].tolist(), rms=r["rms"], iters=r["iters"], sweeps=r["sweeps"])
    finally:
        dist.destroy_process_group()


def test_sharded_fit_two_ranks_on_one_gpu():
    """The multi-GPU algorithm end to end with REAL kernels: two processes share the one GPU, each
    holds half of the images, the reduced system is all-reduced through the torch.distributed hook
    (gloo here; the driver's multi-GPU run uses RCCL).  The joint fit must equal the single-process
    fit of all images: same LM trajectory, parameters to 1e-9 (only the summation order differs)."""
    import socket

    import torch.multiprocessing as mp
    n_total = 41
    wl = syn.named_workload("C2", n_images=n_total)
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    mgr = mp.Manager()
    out = mgr.dict()
    spawn_ranks(_two_rank_worker, lambda port: (2, port, n_total, out), 2, out)
    assert set(out.keys()) == {0, 1}
    a, b = out[0], out[1]
    assert a["intr"] == b["intr"] and a["rms"] == b["rms"] and a["iters"] == b["iters"]     # replicated state
    assert a["iters"] == base["iters"] and a["sweeps"] == base["sweeps"]
    assert abs(a["rms"] - base["rms"]) < 1e-12
    assert np.abs(np.array(a["intr"]) - base["intr"]).max() < 1e-9 * base["intr"][1]
    rv = np.vstack([np.array(a["rvecs"]), np.array(b["rvecs"])])
    tv = np.vstack([np.array(a["tvecs"]), np.array(b["tvecs"])])
    assert (a["lo"], a["hi"], b["lo"], b["hi"]) == (0, 21, 21, 41)
    assert np.abs(rv - base["rvecs"]).max() < 1e-10 and np.abs(tv - base["tvecs"]).max() < 1e-9


@pytest.mark.parametrize("world,n_total,transport", [(2, 41, "p2p"), (3, 41, "p2p"), (2, 2650, "p2p"), (2, 4400, "p2p"),
                                                     (2, 41, "p2p-fused"), (3, 41, "p2p-fused"), (4, 500, "p2p-fused")])
def test_sharded_fit_p2p_exchange_on_one_gpu(world, n_total, transport):
    """Same joint fit through the direct exchange (ccal_p2p_export / ccal_p2p_connect): the ranks'
    inboxes are fine-grained device allocations mapped into each other's process through HIP IPC,
    rows travel as {epoch | payload} words written by the peers' kernels, every solve is one
    hipGraph per rank.  Ranks share the test box's single GPU (the fabric hop is the driver's
    multi-GPU run); 3 ranks cover uneven shards and the rank-ordered sum; 2650 images put more than
    320 workgroups on each rank, i.e. the separate exchange kernel instead of the fused prologue; 4400 images
    (550 image groups per rank) exceed the resident grid, i.e. the looping sweep instantiation.  Ranks that share a GPU
    switch themselves to the one-workgroup exchange ("p2p"); "p2p-fused" forces the form ranks with a GPU each take
    (every workgroup of the trial kernel polls) on grids small enough to be resident together on the one GPU."""
    import socket

    import torch.multiprocessing as mp
    wl = syn.named_workload("C2", n_images=n_total)
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    mgr = mp.Manager()
    out = mgr.dict()
    spawn_ranks(_two_rank_worker, lambda port: (world, port, n_total, out, transport), world, out)
    assert set(out.keys()) == set(range(world))
    a = out[0]
    for r in range(1, world):
        assert out[r]["intr"] == a["intr"] and out[r]["rms"] == a["rms"] and out[r]["iters"] == a["iters"]
    assert a["iters"] == base["iters"] and a["sweeps"] == base["sweeps"]
    assert abs(a["rms"] - base["rms"]) < 1e-12
    assert np.abs(np.array(a["intr"]) - base["intr"]).max() < 1e-9 * base["intr"][1]
    rv = np.vstack([np.array(out[r]["rvecs"]) for r in range(world)])
    tv = np.vstack([np.array(out[r]["tvecs"]) for r in range(world)])
    assert np.abs(rv - base["rvecs"]).max() < 1e-10 and np.abs(tv - base["tvecs"]).max() < 1e-9


def _silent_peer_worker(rank, world, port, out):
    import time

    import torch
    import torch.distributed as dist
    from cameracalibrations_amd._ffi import CcalError
    from cameracalibrations_amd.distributed import allgather_bytes_via_torch, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CCAL_P2P_TIMEOUT_MS"] = "300"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = syn.named_workload("C2", n_images=24)
        lo, hi = shard_range(24, world, rank)
        with Solver(0) as s:
            s.set_problem(wl.obj, hi - lo, wl.image_size, aspect=wl.aspect, flags=wl.flags)
            s.set_image_points(wl.img[lo:hi])
            s.init_p2p(world, rank, 24, allgather_bytes_via_torch())
            if rank == 0:                                  # rank 1 never sends a row
                t0 = time.time()
                try:
                    s.calibrate()
                    out[0] = dict(raised=False)
                except CcalError as exc:
                    out[0] = dict(raised=True, status=exc.code, seconds=time.time() - t0, msg=str(exc))
            dist.barrier()
    finally:
        dist.destroy_process_group()


def test_p2p_silent_peer_times_out_instead_of_hanging():
    """A rank whose peer never shows up must come back with CCAL_E_COMM after CCAL_P2P_TIMEOUT_MS: every
    wait of the exchange kernels is bounded, nothing spins for good on the GPU."""
    import socket

    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    spawn_ranks(_silent_peer_worker, lambda port: (2, port, out), 2, out)
    r = out[0]
    assert r["raised"] and r["status"] == -6, r            # CCAL_E_COMM
    assert r["seconds"] < 5.0 and "timed out" in r["msg"], r


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8, [0] * 16])
def test_calibrate_multi_single_host_thread(devices):
    """ccal_calibrate_multi: one host thread drives one context per listed device (here the test box's
    one GPU, listed up to sixteen times = up to sixteen shards -- the most the exchange carries -- whose rows travel
    through in-process peer pointers; shards of one device share one stream and exchange by send / receive kernels,
    every send enqueued before any receive, so nothing waits for work queued behind it).  The joint fit equals the
    single-context fit of all images, twice in a row (cached contexts), and the per-image poses come back in
    image order."""
    from cameracalibrations_amd.solver import calibrate_multi
    wl = syn.named_workload("C2", n_images=41)
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    for rep in range(2):
        st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, devices, aspect=wl.aspect, flags=wl.flags)
        assert st == 0 and r["iters"] == base["iters"] and r["sweeps"] == base["sweeps"]
        assert abs(r["rms"] - base["rms"]) < 1e-12
        assert np.abs(r["intr"] - base["intr"]).max() < 1e-9 * base["intr"][1]
        assert np.abs(r["rvecs"] - base["rvecs"]).max() < 1e-10 and np.abs(r["tvecs"] - base["tvecs"]).max() < 1e-9
    from cameracalibrations_amd import _ffi
    _ffi.lib().ccal_thread_cleanup()


def test_c_abi_from_plain_c(golden_dir, tmp_path):
    """The boundary is a C ABI, not a Python extension: examples/fit_from_c.c (gcc, include/ccal.h, linked
    against libccal_hip.so only -- no Python, no torch in that process) fits the reference's example corners
    and must print exactly what the Python mirror's `fit_model` returns from the same library."""
    import shutil
    import struct
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "cameracalibrations_amd")
    exe = str(tmp_path / "fit_from_c")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "fit_from_c.c"), "-o", exe, "-L" + libdir, "-lccal_hip",
                           "-Wl,-rpath," + libdir, "-lm"])
    ex = json.load(open(os.path.join(golden_dir, "example_corners.json")))
    n_corners = tuple(ex["n_corners"])
    P = n_corners[0] * n_corners[1]
    img = np.array(ex["corners"], dtype=np.float32).reshape(-1, P, 2)
    obj = syn.board_template(n_corners).astype(np.float32)
    w, h = ex["image_size_wh"]
    path = tmp_path / "corners.bin"
    with open(path, "wb") as fh:
        fh.write(struct.pack("<4i", P, img.shape[0], int(w), int(h)))
        fh.write(obj.tobytes())
        fh.write(img.tobytes())
    out = subprocess.run([exe, str(path), "1", "1.0"], check=True, capture_output=True, text=True, timeout=120).stdout.split("\n")
    head = out[0].split()
    vals = dict(zip(out[1].split()[0::2], map(float, out[1].split()[1::2])))
    m = fit_model((w, h), obj, list(img), n_corners, True, 1.0)
    assert int(head[1]) == 0 and int(head[3]) == m.iters and float(head[5]) == m.rms
    assert (vals["frow"], vals["fcol"], vals["crow"], vals["ccol"], vals["k"]) == (m.frow, m.fcol, m.crow, m.ccol, m.k)
    poses = np.array([[float(x) for x in ln.split()[1:4] + ln.split()[5:8]] for ln in out[2:2 + img.shape[0]]])
    assert np.array_equal(poses[:, :3], np.array(m.Rs)) and np.array_equal(poses[:, 3:], np.array(m.ts))
    # ... and the batched call from C (ccal_calibrate_many, five calibrations: merged launches) returns the same five times
    assert out[2 + img.shape[0]].strip() == "calibrate_many 5 calibrations identical 1"


RANDOM_CASES = [
    # (seed, n_corners, n_images, model, flags, sigma, depth)
    (1, (5, 8), 5, "ref", syn.FIX_ASPECT_RATIO | syn.ZERO_TANGENT_DIST | syn.FIX_K2 | syn.FIX_K3, 0.05, (8.0, 16.0)),
    (2, (9, 6), 23, "k1k2p1p2", syn.FIX_K3, 0.2, (8.0, 16.0)),
    (3, (7, 7), 9, "pinhole", syn.ZERO_TANGENT_DIST | syn.FIX_K1 | syn.FIX_K2 | syn.FIX_K3, 0.3, (8.0, 16.0)),
    (4, (15, 10), 7, "full", 0, 0.1, (14.0, 28.0)),                      # 150 corners: three 64-corner passes per image
    (5, (11, 4), 31, "k1k2p1p2", syn.FIX_K3 | syn.FIX_K2, 0.0, (8.0, 16.0)),  # noise-free (only Float32 rounding)
    (6, (9, 6), 3, "ref", syn.FIX_ASPECT_RATIO | syn.ZERO_TANGENT_DIST | syn.FIX_K1 | syn.FIX_K2 | syn.FIX_K3, 0.1, (8.0, 16.0)),
    (7, (13, 5), 12, "k1k2p1p2", syn.FIX_K3 | syn.FIX_PRINCIPAL_POINT, 0.15, (10.0, 20.0)),
    (8, (6, 5), 64, "full", syn.FIX_K3, 0.1, (8.0, 16.0)),
]


@pytest.mark.parametrize("seed,n_corners,n_images,model,flags,sigma,depth", RANDOM_CASES)
def test_random_problems_match_oracle(seed, n_corners, n_images, model, flags, sigma, depth):
    """Seeded problems of assorted board shapes (incl. multi-pass 150-corner images), image counts, models,
    flag words and noise levels: full fit (device init + LM, OpenCV's criteria) against the oracle --
    same accepted-iteration count, RMS to 1e-8 px, parameters to 1e-6 relative / 1e-5."""
    wl = syn.make_workload("rnd", n_corners, n_images, model, seed=1000 + seed, sigma=sigma, depth=depth, aspect=1.0)
    aspect = 1.0
    ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=1e-3, solver="schur")
    with Solver(0) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=aspect, flags=flags, max_iter=30, eps=1e-3)
        s.set_image_points(wl.img)
        st, r = s.calibrate()
    assert st in (0, 1)
    assert r["iters"] == ro["iters"], (r["iters"], ro["iters"])
    assert abs(r["rms"] - ro["rms"]) < 1e-8
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6
    assert np.abs(r["intr"][4:] - ro["intr"][4:]).max() < 1e-5
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5 and np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-4


def test_p2p_selftest_world1():
    """ccal_p2p_selftest on a one-rank exchange (own inbox only): the stand-alone all-reduce kernel, sequence
    numbers advancing on the device, known sums back on the host."""
    with Solver(0) as s:
        s.init_p2p(1, 0, 8, lambda b: [b])
        s.p2p_selftest(12)
        s.p2p_selftest(3)


# ---- BASELINE.json configs[3]: the real 1000-image C3 set sharded over ranks, against the ORACLE -------------------
_C3_ORACLE = {}


def _c3_oracle():
    if not _C3_ORACLE:
        wl = syn.named_workload("C3")
        _C3_ORACLE.update(wl=wl, ro=_oracle(wl, REF, nthreads=8))
    return _C3_ORACLE["wl"], _C3_ORACLE["ro"]


def _c3_shard_worker(rank, world, port, out, transport, explicit_stream):
    import torch
    import torch.distributed as dist
    from cameracalibrations_amd.distributed import TorchAllReduce, allgather_bytes_via_torch, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CCAL_P2P_TIMEOUT_MS"] = "20000"
    torch.cuda.set_device(0)                               # the ranks share the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = syn.named_workload("C3")
        lo, hi = shard_range(wl.n_images, world, rank)
        kw = dict(stream=torch.cuda.current_stream().cuda_stream) if explicit_stream else {}
        with Solver(0, **kw) as s:                         # no stream=: the context's own NON-BLOCKING stream
            s.set_problem(wl.obj, hi - lo, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
            s.set_image_points(torch.from_numpy(wl.img[lo:hi]).cuda())
            if transport == "p2p":
                s.init_p2p(world, rank, wl.n_images, allgather_bytes_via_torch())
            elif transport == "p2p-fused":
                s.init_p2p(world, rank, wl.n_images, allgather_bytes_via_torch(), shared_device=False)
            else:
                s.set_allreduce(TorchAllReduce(), world, rank, wl.n_images)
            st, r = s.calibrate()
            st2, r2 = s.calibrate()                        # again (graph replay with the peers for p2p)
            assert np.array_equal(r["intr"], r2["intr"]) and r["rms"] == r2["rms"] and r["iters"] == r2["iters"]
            dist.barrier()
        out[rank] = dict(status=st, lo=lo, hi=hi, intr=r["intr"].tolist(), rvecs=r["rvecs"].tolist(), tvecs=r["tvecs"].tolist(),
                         rms=r["rms"], iters=r["iters"], sweeps=r["sweeps"], rejected=r["rejected"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "p2p"), (4, "p2p"), (2, "p2p-fused"), (2, "torch-hook")])
def test_c3_sharded_over_ranks_matches_oracle(world, transport):
    """BASELINE.json configs[3]: the SAME 1000-image x 9x6 set of configs[2], sharded per image over 2 / 4 ranks (sharing
    the test box's one GPU; at most 6 processes may use it, so 8 ranks are covered by the CPU world-size tests), through
    the direct exchange and through the torch.distributed hook on the context's own non-blocking stream (the hook must
    order itself on the stream it is handed).  Checked against the ORACLE, not against the single-GPU device fit:
    RMS to 1e-8 px, same accepted-iteration count, parameters to the full-fit tolerances."""
    import socket

    import torch.multiprocessing as mp
    wl, ro = _c3_oracle()
    mgr = mp.Manager()
    out = mgr.dict()
    spawn_ranks(_c3_shard_worker, lambda port: (world, port, out, transport if transport.startswith("p2p") else "torch", False), world, out)
    assert set(out.keys()) == set(range(world))
    a = out[0]
    for r in range(1, world):
        assert out[r]["intr"] == a["intr"] and out[r]["rms"] == a["rms"] and out[r]["iters"] == a["iters"]   # replicated state
    assert a["iters"] == ro["iters"] and a["rejected"] == 0
    assert abs(a["rms"] - ro["rms"]) < 1e-8
    intr = np.array(a["intr"])
    assert np.abs(intr[:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6 and np.abs(intr[4:] - ro["intr"][4:]).max() < 1e-5
    rv = np.vstack([np.array(out[r]["rvecs"]) for r in range(world)])
    tv = np.vstack([np.array(out[r]["tvecs"]) for r in range(world)])
    assert rv.shape == (1000, 3)
    assert np.abs(rv - ro["rvecs"]).max() < 1e-5 and np.abs(tv - ro["tvecs"]).max() < 1e-4


def test_calibrate_multi_reports_a_degenerate_shard_and_recovers():
    """One shard of a multi-shard fit holds an image without a homography (all corners in one spot): the call comes back
    with the initialisation's error (no hang: every stage of every shard was enqueued before any wait), and the cached
    contexts serve the next, healthy call."""
    from cameracalibrations_amd import _ffi as F
    from cameracalibrations_amd.solver import calibrate_multi
    wl = syn.named_workload("C2", n_images=24)
    bad = wl.img.copy()
    bad[17] = bad[17, 0]
    for devices in ([0, 0, 0], [0] * 8):
        with pytest.raises(F.CcalError) as e:
            calibrate_multi(wl.obj, bad, wl.image_size, devices, aspect=wl.aspect, flags=wl.flags)
        assert e.value.code in (F.E_DEGENERATE, F.E_NONFINITE), e.value
        st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, devices, aspect=wl.aspect, flags=wl.flags)
        assert st == 0 and 0.05 < r["rms"] < 0.3
    F.lib().ccal_thread_cleanup()


def _check_against_oracle(r, ro):
    assert r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < 1e-8
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6 and np.abs(r["intr"][4:] - ro["intr"][4:]).max() < 1e-5
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5 and np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-4


def test_c3_eight_shards_match_oracle():
    """BASELINE.json configs[3] at its largest world size: the 1000-image x 9x6 set in EIGHT shards of 125 images
    (ccal_calibrate_multi with the test box's one GPU listed eight times: the rank-ordered sum over eight rows, the
    four-slot tag protocol and the 544-byte row at world = 8), against the ORACLE -- RMS to 1e-8 px, same
    accepted-iteration count, parameters to the full-fit tolerances -- and against the one-GPU device fit."""
    from cameracalibrations_amd.solver import calibrate_multi
    wl, ro = _c3_oracle()
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    for rep in range(2):
        st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, [0] * 8, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
        assert st == 0 and r["rejected"] == 0
        _check_against_oracle(r, ro)
        assert r["sweeps"] == base["sweeps"] and abs(r["rms"] - base["rms"]) < 1e-12
        assert np.abs(r["intr"] - base["intr"]).max() < 1e-9 * base["intr"][1]
        assert np.abs(r["rvecs"] - base["rvecs"]).max() < 1e-10 and np.abs(r["tvecs"] - base["tvecs"]).max() < 1e-9
    from cameracalibrations_amd import _ffi
    _ffi.lib().ccal_thread_cleanup()


_C5_ORACLE = {}


def _c5_oracle():
    if not _C5_ORACLE:
        wl = syn.named_workload("C5")
        _C5_ORACLE.update(wl=wl, ro=_oracle(wl, REF, nthreads=8))
    return _C5_ORACLE["wl"], _C5_ORACLE["ro"]


def test_c5_single_gpu_matches_oracle_full_size():
    """BASELINE.json configs[4]'s problem (10000 images x 15x10 corners, full k1..k3,p1,p2 model) whole on one GPU
    against the ORACLE itself (its Schur mode fits the 1.5 M corners in seconds): RMS to 1e-8 px, same
    accepted-iteration count, parameters to the full-fit tolerances."""
    wl, ro = _c5_oracle()
    with _solver(wl, REF) as s:
        st, r = s.calibrate()
    assert st in (0, 1)
    _check_against_oracle(r, ro)


def test_exact_focal_sums_at_ten_thousand_images():
    """Row A4 at BASELINE.json configs[4]'s size: the focal least squares summed in the exact accumulators over 10000 images
    (up to 1250 adds per address; the total over the eight replicas is past an int64 from ~4096 images on, which the reader of
    round 4 did not survive) against the oracle's initial intrinsics -- an independent reference for the totals -- and the
    fit's one-pipeline form (solve in the pose kernel's prologue) against the stage-by-stage form, bit for bit."""
    wl = syn.named_workload("C5")
    K4, _ = orc.init_intrinsics(wl.obj, wl.img, *wl.image_size, aspect=wl.aspect if wl.flags & 2 else 0.0)
    with _solver(wl, REF) as s:
        s.init_intrinsics()
        gi, _ = s.get_init(extrinsics=False)
        assert np.abs(gi[:2] - K4[:2]).max() / K4[1] < 1e-8, (gi[:4], K4)
        assert np.array_equal(gi[2:4], K4[2:4])
        s.fit()
        gp, _ = s.get_init(extrinsics=False)
        assert np.array_equal(gp, gi)


def test_c5_eight_shards_match_oracle():
    """BASELINE.json configs[4] as named: the 10000 x 15x10 problem in EIGHT shards of 1250 images (the per-GPU share
    at 8 MI355X; here eight shards on the test box's one GPU driven by one host thread), against the ORACLE and
    against the one-GPU device fit, plus the oracle's stationarity spot-check at the returned point."""
    from cameracalibrations_amd.solver import calibrate_multi
    wl, ro = _c5_oracle()
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, [0] * 8, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
    assert st in (0, 1)
    _check_against_oracle(r, ro)
    assert r["iters"] == base["iters"] and r["sweeps"] == base["sweeps"] and abs(r["rms"] - base["rms"]) < 1e-11
    assert np.abs(r["intr"] - base["intr"]).max() < 1e-8 * base["intr"][1]
    assert np.abs(r["rvecs"] - base["rvecs"]).max() < 1e-9 and np.abs(r["tvecs"] - base["tvecs"]).max() < 1e-8
    asp = wl.aspect if wl.flags & 2 else 0.0
    for i in (0, 1249, 1250, 5000, 9999):           # first / last image of a shard, a shard boundary, the middle
        G = orc.image_gram(wl.obj, wl.img[i], r["rvecs"][i], r["tvecs"][i], r["intr"], asp)
        grad = G[:6, 15] / np.sqrt(np.diag(G)[:6] * G[15, 15])
        assert np.abs(grad).max() < 5e-3            # (criteria (30, 1e-3): stopped at the reference's loose rule; oracle: 7e-4)
    from cameracalibrations_amd import _ffi
    _ffi.lib().ccal_thread_cleanup()


def test_c3_single_gpu_matches_oracle_full_size():
    """BASELINE.json configs[2] at full size against the oracle (the sharded test's single-GPU twin)."""
    wl, ro = _c3_oracle()
    with _solver(wl, REF) as s:
        st, r = s.calibrate()
    assert r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < 1e-8
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6
    assert np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5 and np.abs(r["tvecs"] - ro["tvecs"]).max() < 1e-4


def test_uneven_shards_across_the_exchange_threshold():
    """2561 images on two ranks = 1281 + 1280: 321 and 320 workgroups, one on each side of the limit up to which the
    exchange is fused into the trial's prologue.  The mode is chosen from the GLOBAL size, so both ranks take the same
    path (a mixed pair would stop exchanging at different trials); the solve is run twice (graph replay, then the
    surplus launches of the first solve must not have left anything behind)."""
    import socket

    import torch.multiprocessing as mp
    n_total = 2561
    wl = syn.named_workload("C2", n_images=n_total)
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    mgr = mp.Manager()
    out = mgr.dict()
    spawn_ranks(_two_rank_worker, lambda port: (2, port, n_total, out, "p2p"), 2, out)
    a, b = out[0], out[1]
    assert (a["hi"] - a["lo"], b["hi"] - b["lo"]) == (1281, 1280)
    assert a["intr"] == b["intr"] and a["iters"] == base["iters"] and abs(a["rms"] - base["rms"]) < 1e-12


def test_many_image_maps_match_per_image_calls_and_oracle():
    """ccal_project_many / ccal_unproject_many (one call for all images, per-image poses) == the per-image entry points
    == the oracle's Julia-map restatement (src/meta.jl:27-76); small calls (pinned zero-copy path) and large ones
    (device scratch path)."""
    rng = np.random.default_rng(4)
    f, c, k = np.array([1600.0, 1610.0]), np.array([750.0, 1000.0]), -0.11
    for N, M in ((3, 7), (40, 54), (300, 150)):
        rv = rng.uniform(-0.4, 0.4, (N, 3))
        tv = np.c_[rng.uniform(-2, 2, (N, 2)), rng.uniform(9, 15, N)]
        xyz = np.c_[rng.uniform(0, 8, (M, 2)), np.zeros(M)]
        with Solver(0) as s:
            rc = s.project_many(xyz, f, c, k, rv, tv, 2.5)
            back = s.unproject_many(rc, f, c, k, rv, tv, 2.5)
            per_img = s.project_many(np.broadcast_to(xyz, (N, M, 3)).copy(), f, c, k, rv, tv, 2.5)
            one = np.stack([s.project(xyz, f, c, k, rv[i], tv[i], 2.5) for i in range(min(N, 5))])
        assert rc.shape == (N, M, 2) and back.shape == (N, M, 3)
        assert np.array_equal(per_img, rc) and np.array_equal(one, rc[: one.shape[0]])
        for i in range(min(N, 5)):
            assert np.abs(rc[i] - orc.map_forward(xyz, f, c, k, rv[i], tv[i], 2.5)).max() < 1e-9
            assert np.abs(back[i] - orc.map_inverse(rc[i], f, c, k, rv[i], tv[i], 2.5)).max() < 1e-8
        assert np.abs(back - xyz[None]).max() < 1e-8          # round trip on the board plane


def test_one_pipeline_fit_equals_the_stages():
    """ccal_fit (initialisation + LM enqueued back to back, one wait) == the three stage calls, bit for bit, also when
    repeated and on a device-resident corner array; a failed initialisation still surfaces as its own error."""
    import torch

    from cameracalibrations_amd import _ffi as F
    for name, n in (("C2", 37), ("REF", 9), ("C5", 7)):
        wl = syn.named_workload(name, n_images=n)
        with _solver(wl, REF) as s:
            st, a = s.calibrate()
            for _ in range(2):
                assert s.fit() == st
                b = s.result()
                for k in ("intr", "rvecs", "tvecs"):
                    assert np.array_equal(a[k], b[k]), (name, k)
                assert a["rms"] == b["rms"] and a["iters"] == b["iters"] and a["sweeps"] == b["sweeps"]
        with Solver(0) as s:
            s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
            s.set_image_points(torch.from_numpy(wl.img).cuda())
            assert s.fit() == st and np.array_equal(s.result()["intr"], a["intr"])
    # with an intrinsic guess (CALIB_USE_INTRINSIC_GUESS) the pipeline starts at the pose initialisation
    wl = syn.named_workload("C2", n_images=11)
    guess = np.r_[wl.intr_true[:4] * 1.01, 0.0, 0.0, 0.0, 0.0, 0.0]
    with Solver(0) as s:
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=wl.aspect, flags=wl.flags | syn.USE_INTRINSIC_GUESS)
        s.set_image_points(wl.img)
        with pytest.raises(F.CcalError):
            s.fit()                                            # no guess supplied yet
        st, a = s.calibrate(init_intr=guess)
        s.set_intrinsics(guess)
        assert s.fit() == st and np.array_equal(s.result()["intr"], a["intr"]) and np.array_equal(s.result()["tvecs"], a["tvecs"])
    wl = syn.named_workload("C2", n_images=5)
    bad = wl.img.copy()
    bad[2] = bad[2, 0]                                         # one image with all corners in one spot: no homography
    with Solver(0) as s:
        s.set_problem(wl.obj, 5, wl.image_size, aspect=wl.aspect, flags=wl.flags)
        s.set_image_points(bad)
        with pytest.raises(F.CcalError) as e:
            s.fit()
        assert e.value.code in (F.E_DEGENERATE, F.E_NONFINITE)
        s.set_image_points(wl.img)                             # the context is usable afterwards
        assert s.fit() in (0, 1) and 0.05 < s.result()["rms"] < 0.3


def test_point_wise_calls_through_a_batch():
    """Calibration.batch(): a loop of point-wise calls like the reference's round-trip metric (src/buildcalibrations.jl:53-58)
    is evaluated in two device calls; results equal the per-point calls bit for bit and the oracle's maps; sequences of
    points and Calibration.map_all (image n through pose n) likewise."""
    from cameracalibrations_amd.calibration import Calibration
    rng = np.random.default_rng(9)
    N = 6
    rv = rng.uniform(-0.4, 0.4, (N, 3))
    tv = np.c_[rng.uniform(-2, 2, (N, 2)), rng.uniform(9, 15, N)]
    c = Calibration(np.array([1600.0, 1610.0]), np.array([750.0, 1000.0]), rv, tv, 2.5, -0.11, [f"{i}.png" for i in range(N)])
    pts = [(np.r_[rng.uniform(0, 8, 2), 0.0], int(rng.integers(1, N + 1))) for _ in range(100)]
    with c.batch() as b:
        fwd = [b(x, i) for x, i in pts]
        assert np.isnan(fwd[0]).all()                          # nothing is evaluated before the block ends
    with c.batch() as b:
        back = [b(rc, i) for rc, (_, i) in zip(fwd, pts)]
    for (x, i), rc, xyz in zip(pts[:12], fwd, back):
        assert np.array_equal(rc, c(x, i)) and np.array_equal(xyz, c(rc, i))
        assert np.abs(rc - orc.map_forward(x[None], c.f, c.c, c.k, rv[i - 1], tv[i - 1], 2.5)[0]).max() < 1e-9
    assert max(np.abs(xyz - x).max() for (x, _), xyz in zip(pts, back)) < 1e-8      # round trip on the board plane
    seq = c([x for x, _ in pts[:7]], 2)                        # a sequence of points: one call
    assert seq.shape == (7, 2) and np.array_equal(seq[3], c(pts[3][0], 2))
    tmpl = np.c_[rng.uniform(0, 8, (9, 2)), np.zeros(9)]
    allp = c.map_all(tmpl)
    assert allp.shape == (N, 9, 2) and np.array_equal(allp[4], c(tmpl, 5))
    assert np.abs(c.map_all(allp) - tmpl[None]).max() < 1e-8
    with pytest.raises(TypeError):
        with c.batch() as b:
            b(np.zeros(4), 1)


def test_cached_contexts_are_released_at_thread_exit():
    """ccal_calibrate keeps one context per host thread (stream, ~25 device buffers, launch graph); a thread that ends
    must give them back (thread_local guard in csrc/ccal_api.hip), or a pool of short-lived worker threads leaks."""
    import threading

    import torch
    wl = syn.named_workload("REF", n_images=12)
    free = []

    def work():
        m = fit_model(wl.image_size, wl.obj, list(wl.img), wl.n_corners, True, 1.0)
        assert 0.05 < m.rms < 0.2
    for _ in range(6):
        t = threading.Thread(target=work)
        t.start()
        t.join()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info(0)[0])
    # the first thread pays for one-time runtime allocations; after that nothing may pile up
    assert free[-1] >= free[1] - (4 << 20), free


@pytest.mark.parametrize("world", [2, 8, 16])
def test_row_exchange_protocol_at_world_8_and_16_across_the_tag_wrap(world):
    """`p2p_exchange_rows` (what ranks with a GPU each run, fused into k_lm_trial) at world sizes no multi-process test on one
    GPU can reach: `world` workgroups of ONE launch act as the ranks of a node (ccal_debug_p2p_world), 24 consecutive exchanges
    numbered like eight solves of three trials, starting at sequence number 0xFFFC so that the 16-bit sequence field of the tag
    (ccal_p2p_dev.hpp:10) wraps in the middle.  Every rank must hold the RANK-ORDERED sum, bit for bit, after every exchange."""
    rng = np.random.default_rng(world)
    rows = rng.standard_normal((24, world, 68)) * 10.0 ** rng.integers(-8, 8, size=(24, world, 68))
    with Solver(0) as s:
        got = s.debug_p2p_world(rows, seq0=0xFFFC)
    want = np.zeros((24, 68))
    for r in range(world):
        want = want + rows[:, r, :]             # ((r0 + r1) + r2) + ...: the order every rank adds in
    for r in range(world):
        assert np.array_equal(got[:, r, :], want), r


def test_sharded_solves_across_the_exchange_epoch_wrap():
    """Four sharded fits (two and three shards of one host thread: k_lm_send / k_lm_recv + the split focal all-reduce) with the
    solve sequence number started at 0xFFFE: the tag's 16-bit sequence field wraps between them.  Every fit must return the
    bits of the fit before the wrap."""
    from cameracalibrations_amd import _ffi
    from cameracalibrations_amd.solver import calibrate_multi, debug_multi_set_epoch
    wl = syn.named_workload("C2", n_images=48)
    for devs in ([0, 0], [0, 0, 0]):
        st, base = calibrate_multi(wl.obj, wl.img, wl.image_size, devs, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
        assert st in (0, 1)
        debug_multi_set_epoch(devs, 0xFFFE)
        for rep in range(4):
            st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, devs, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
            assert st in (0, 1)
            assert r["iters"] == base["iters"] and r["rms"] == base["rms"]
            assert np.array_equal(r["intr"], base["intr"]) and np.array_equal(r["rvecs"], base["rvecs"]) and np.array_equal(r["tvecs"], base["tvecs"])
        _ffi.lib().ccal_thread_cleanup()


def test_pose_initialisation_of_a_2025_corner_board():
    """No limit on n_corners (the reference puts none): a 45 x 45 board is 2025 corners per view -- 32 passes of 64 lanes in every
    kernel, and 65 KB of normalised correspondences, more than a workgroup's LDS: k_init_extrinsics must take them from memory
    (its LDSP = false form), both when the pixel-space homographies of the intrinsics stage are reused and when the poses
    are asked for on their own (OpenCV's own sequence: DLT + polish on normalised points).  Against the oracle."""
    rng = np.random.default_rng(7)
    n1 = n2 = 45
    obj = syn.board_template((n1, n2))
    N = 3
    intr = np.array([1600.0, 1600.0, 756.8, 995.4, 0, 0, 0, 0, 0])
    rts, imgs = [], []
    for i in range(N):
        r = rng.uniform(-0.3, 0.3, 3)
        t = np.array([-22.0 + rng.uniform(-2, 2), -22.0 + rng.uniform(-2, 2), 80.0 + rng.uniform(0, 20)])
        proj = orc.project_points(obj, r, t, intr)
        imgs.append(np.asarray(proj, dtype=np.float64) + rng.normal(0, 0.1, (n1 * n2, 2)))
        rts.append(np.r_[r, t])
    img = np.stack(imgs).astype(np.float32)
    rt_or = np.array([np.r_[orc.find_extrinsics(obj, img[i], intr)] for i in range(N)])
    with Solver(0) as s:
        s.set_problem(obj, N, (1500, 2000), aspect=1.0, flags=0, max_iter=30, eps=1e-3)
        s.set_image_points(img)
        s.set_intrinsics(intr)                 # no intrinsics stage has run: OpenCV's own sequence on normalised points
        s.init_extrinsics()
        _, a = s.get_init()
        s.init_intrinsics()                    # ... and with its pixel-space homographies at hand
        s.set_intrinsics(intr)
        s.init_extrinsics()
        _, b = s.get_init()
    for got in (a, b):
        assert np.abs(got[:, :3] - rt_or[:, :3]).max() < 1e-7 and np.abs(got[:, 3:] - rt_or[:, 3:]).max() < 1e-5
        assert np.abs(got - np.array(rts)).max() < 1e-2
    # ... and the whole fit of such a board (32 passes per image, split over four waves by the pass-parallel sweep) against the oracle
    ro = orc.calibrate(obj, img, (1500, 2000), aspect=1.0, flags=syn.FIX_K3, max_iter=30, eps=1e-3, solver="schur")
    with Solver(0) as s:
        s.set_problem(obj, N, (1500, 2000), aspect=1.0, flags=syn.FIX_K3, max_iter=30, eps=1e-3)
        s.set_image_points(img)
        st, r = s.calibrate()
    assert st in (0, 1) and r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < 1e-8
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-6 and np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-5


def test_sharded_initialisation_takes_the_form_of_the_whole_fit():
    """The initialisation kernels have a latency form and a throughput form whose trial errors are summed differently; which
    one runs follows from the GLOBAL image count: the first 550 images of a 1100-image fit (more images than the chip has
    SIMDs), initialised as a shard of that fit, get the poses the whole fit computes for them BIT for bit -- on their own,
    550 images would take the other form -- and the fit in two shards ends where the whole fit ends."""
    from cameracalibrations_amd.solver import calibrate_multi
    wl = syn.named_workload("C1", n_images=1100)
    intr0 = np.array([1601.0, 1599.0, 749.5, 999.5, 0, 0, 0, 0, 0])
    with _solver(wl, REF) as s:
        s.set_intrinsics(intr0)
        s.init_extrinsics()
        _, whole = s.get_init()
        st, rw = s.calibrate()
    with Solver(0) as s:
        s.set_problem(wl.obj, 550, wl.image_size, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
        s.set_image_points(np.ascontiguousarray(wl.img[:550]))
        s.set_allreduce(lambda buf, n, stream: 0, 1, 0, 1100)      # a shard of a 1100-image fit (sums of one rank: identity)
        s.set_intrinsics(intr0)
        s.init_extrinsics()
        _, part = s.get_init()
    assert np.array_equal(part, whole[:550])
    st2, r2 = calibrate_multi(wl.obj, wl.img, wl.image_size, [0, 0], aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
    assert st2 == st and r2["iters"] == rw["iters"] and abs(r2["rms"] - rw["rms"]) < 1e-12
    # (the two ranks' rows are added as doubles after each was decoded from its exact accumulators: last-bit differences in the
    #  reduced system, 1e-11 in the poses of the ill-conditioned views)
    assert np.abs(r2["rvecs"] - rw["rvecs"]).max() < 1e-9 and np.abs(r2["tvecs"] - rw["tvecs"]).max() < 1e-8
    from cameracalibrations_amd import _ffi
    _ffi.lib().ccal_thread_cleanup()


def _mismatched_model_set():
    """257 views of a 14 x 7 board generated WITH k1, k2, p1, p2 and fitted with k1 = k2 = k3 = 0 (flag word 0xE0), noise-free:
    2.3 px of model error (draw 421 of seed 104 of tools/fuzz_parity.py).  One of its views exhausts the 20 iterations of
    OpenCV's per-view pose polish, so where that view ends depends on where the polish started."""
    return syn.make_workload("fuzz", (14, 7), 257, "k1k2p1p2", seed=262586006, sigma=0.0, depth=(14.0, 28.0), aspect=1.0), 0xE0


@pytest.mark.parametrize("mode", [0, 1])
def test_pose_stage_follows_opencvs_trajectory_on_a_mismatched_model(mode):
    """ccal_set_pose_init 0 (OpenCV's own sequence) and 1 (the default: the same with the DLT's result reused from the
    intrinsics stage) must both reproduce the oracle's initial poses -- 1e-10, INCLUDING the view that runs into CvLevMarq's
    20-iteration cap -- and with them the oracle's LM trajectory: first-trial error to 1e-9 relative, RMS to 1e-9 px."""
    wl, flags = _mismatched_model_set()
    K4, _ = orc.init_intrinsics(wl.obj, wl.img, *wl.image_size, aspect=0.0)
    intr0 = np.r_[K4, np.zeros(5)]
    rt0 = np.array([np.r_[orc.find_extrinsics(wl.obj, wl.img[i], intr0)] for i in range(wl.n_images)])
    ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=1.0, flags=flags, max_iter=30, eps=1e-3, solver="schur")
    with Solver(0) as s:
        s.set_pose_init(mode)
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=1.0, flags=flags, max_iter=30, eps=1e-3)
        s.set_image_points(wl.img)
        s.init_intrinsics()
        s.set_intrinsics(intr0)        # the oracle's start intrinsics for the pose stage (the device's differ by 2e-7 here)
        s.init_extrinsics()
        _, rt = s.get_init()
        st, r = s.calibrate()
        tr = [t for t in s.trace() if t["accepted"] != 2]
    assert np.abs(rt[:, :3] - rt0[:, :3]).max() < 1e-10 and np.abs(rt[:, 3:] - rt0[:, 3:]).max() < 1e-9
    assert r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < 1e-9
    assert abs(tr[0]["err"] - ro["trace"][0]["err"]) < 1e-9 * ro["trace"][0]["err"]
    assert np.abs(r["intr"][:4] - ro["intr"][:4]).max() / ro["intr"][1] < 1e-9 and np.abs(r["rvecs"] - ro["rvecs"]).max() < 1e-7


def test_fast_pose_init_is_the_same_optimum_for_views_that_converge():
    """ccal_set_pose_init 2 (opt-in: the polished pixel-space homography decomposed directly, no second homography): on the
    bench's C3 set every pose within 1e-7 / 1e-6 of the oracle's and the fit that follows equal to the oracle's (same iteration
    count, RMS 1e-8); on the mismatched-model set all views but the one that exhausts its 20 iterations agree to 1e-7, that one
    ends 2.4e-4 rad away, and the fit's RMS stays within 1e-6 px of the oracle's -- inside north_star's 1e-4 px, outside the 1e-8
    this suite holds the default to: why the mode is opt-in."""
    wl3 = syn.named_workload("C3", n_images=200)
    intr3, rt3 = _oracle_init(wl3)
    ro3 = _oracle(wl3, REF)
    with _solver(wl3, REF) as s:
        s.set_pose_init(2)
        s.init_intrinsics()
        s.set_intrinsics(intr3)
        s.init_extrinsics()
        _, g3 = s.get_init()
        st, r3 = s.calibrate()
    assert np.abs(g3[:, :3] - rt3[:, :3]).max() < 1e-7 and np.abs(g3[:, 3:] - rt3[:, 3:]).max() < 1e-6
    assert r3["iters"] == ro3["iters"] and abs(r3["rms"] - ro3["rms"]) < 1e-8
    wl, flags = _mismatched_model_set()
    K4, _ = orc.init_intrinsics(wl.obj, wl.img, *wl.image_size, aspect=0.0)
    intr0 = np.r_[K4, np.zeros(5)]
    rt0 = np.array([np.r_[orc.find_extrinsics(wl.obj, wl.img[i], intr0)] for i in range(wl.n_images)])
    ro = orc.calibrate(wl.obj, wl.img, wl.image_size, aspect=1.0, flags=flags, max_iter=30, eps=1e-3, solver="schur")
    with Solver(0) as s:
        s.set_pose_init(2)
        s.set_problem(wl.obj, wl.n_images, wl.image_size, aspect=1.0, flags=flags, max_iter=30, eps=1e-3)
        s.set_image_points(wl.img)
        s.init_intrinsics()
        s.set_intrinsics(intr0)
        s.init_extrinsics()
        _, rt = s.get_init()
        st, r = s.calibrate()
    d = np.abs(rt - rt0).max(axis=1)
    assert (d > 1e-7).sum() <= 1 and d.max() < 1e-3
    assert r["iters"] == ro["iters"] and abs(r["rms"] - ro["rms"]) < 1e-6


def test_split_images_in_a_sharded_fit():
    """The pass-parallel sweep (four waves per 150-corner image) inside a sharded fit: 40 views of the 15 x 10 board in two and
    in three shards of one host thread (the exchanging instantiations of k_lm_trial with WPI = 4) against the one-GPU fit and
    the oracle."""
    from cameracalibrations_amd import _ffi
    from cameracalibrations_amd.solver import calibrate_multi
    wl = syn.named_workload("C5", n_images=40)
    ro = _oracle(wl, REF)
    with _solver(wl, REF) as s:
        _, base = s.calibrate()
    _check_against_oracle(base, ro)
    for devs in ([0, 0], [0, 0, 0]):
        st, r = calibrate_multi(wl.obj, wl.img, wl.image_size, devs, aspect=wl.aspect, flags=wl.flags, max_iter=REF[0], eps=REF[1])
        assert st in (0, 1)
        _check_against_oracle(r, ro)
        assert r["iters"] == base["iters"] and abs(r["rms"] - base["rms"]) < 1e-12
        assert np.abs(r["rvecs"] - base["rvecs"]).max() < 1e-9 and np.abs(r["tvecs"] - base["tvecs"]).max() < 1e-8
        _ffi.lib().ccal_thread_cleanup()


@pytest.mark.parametrize("name,n", [("C3", 300), ("C5", 1100), ("REF", 9)])
def test_one_pipeline_fit_equals_the_stages_bit_for_bit(name, n):
    """`ccal_fit` (the stages enqueued as one pipeline: the focal 2 x 2 solve runs in the pose kernel's prologue, the init's flags
    travel with the first LM launch) against the same stages called one by one (their own focal kernel, the read-back kernel):
    same initial values and same fit, bit for bit -- in the latency form of the init kernels (300 / 9 images) and in the throughput
    form (1100 images, more than the chip has SIMDs), whose pose kernel once compiled to other bits with the intrinsics known to
    the compiler."""
    wl = syn.named_workload(name, n_images=n)
    with _solver(wl, REF) as s:
        s.init_intrinsics()
        s.init_extrinsics()
        i_st, rt_st = s.get_init()
        s.lm_run()
        r_st = s.result()
        s.fit()
        i_pl, rt_pl = s.get_init()
        r_pl = s.result()
    assert np.array_equal(i_st, i_pl) and np.array_equal(rt_st, rt_pl)
    assert r_st["rms"] == r_pl["rms"] and r_st["iters"] == r_pl["iters"]
    for k in ("intr", "rvecs", "tvecs"):
        assert np.array_equal(r_st[k], r_pl[k]), k


@pytest.mark.parametrize("name,n,crit", [("REF", 4, REF), ("C1", 3, TIGHT), ("C2", 4, TIGHT), ("C5", 3, REF), ("C2", 1, REF), ("REF", 6, REF)])
def test_one_launch_solve_equals_the_launch_per_trial_form_bit_for_bit(name, n, crit, monkeypatch):
    """Fits of at most four images on one GPU run every LM trial in ONE launch of one workgroup (k_lm_trial<.., SOLO>: the sums
    of a trial stay in LDS and are added up the way the exact accumulators add the workgroups' sums).  Same decisions, same bits
    as one launch per trial (CCAL_NO_SOLO=1; multi-pass boards with one wave per image there too, as the one-launch form sweeps
    them), and the oracle's trajectory.  Six images: not eligible, both arms run the same launches."""
    wl = syn.named_workload(name, n_images=n)
    monkeypatch.setenv("CCAL_WPI", "1")
    out = []
    for no_solo in ("0", "1"):
        monkeypatch.setenv("CCAL_NO_SOLO", no_solo)
        with _solver(wl, crit) as s:
            st, r = s.calibrate()
            assert st in (0, 1)
            out.append((r, s.trace()))
    (a, ta), (b, tb) = out
    assert a["iters"] == b["iters"] and a["sweeps"] == b["sweeps"] and a["rejected"] == b["rejected"]
    assert np.array_equal(a["intr"], b["intr"]) and np.array_equal(a["rvecs"], b["rvecs"]) and np.array_equal(a["tvecs"], b["tvecs"])
    assert a["rms"] == b["rms"] and a["final_err"] == b["final_err"]
    assert len(ta) == len(tb) and all(x == y for x, y in zip(ta, tb))
    ro = _oracle(wl, crit)
    assert a["iters"] == ro["iters"] and abs(a["rms"] - ro["rms"]) < 1e-8


def test_small_drop_in_calls_stage_through_the_pinned_block(golden_dir, monkeypatch):
    """A ccal_calibrate call whose inputs fit the context's pinned block stages them there (a kernel copies them, no copy commands from
    pageable memory) and receives its poses from the launch that ends the solve (no read-back launch): the same bits as the copy-command
    path (CCAL_NO_SMALL_CALL=1), on the reference's example (6 x 40, launch-per-trial LM) and on a 3-image fit (one-launch LM)."""
    with open(os.path.join(golden_dir, "example_corners.json")) as f:
        ex = json.load(f)
    from cameracalibrations_amd import board_template
    obj = board_template(ex["n_corners"])
    imgs = [np.array(m, dtype=np.float64) for m in ex["corners"]]
    for sub in (imgs, imgs[:3]):
        out = []
        for no_small in ("0", "1"):
            if no_small == "1":
                monkeypatch.setenv("CCAL_NO_SMALL_CALL", "1")
            else:
                monkeypatch.delenv("CCAL_NO_SMALL_CALL", raising=False)
            out.append(fit_model(ex["image_size_wh"], obj, sub, ex["n_corners"], True, 1))
        a, b = out
        assert (a.frow, a.fcol, a.crow, a.ccol, a.k, a.rms, a.iters) == (b.frow, b.fcol, b.crow, b.ccol, b.k, b.rms, b.iters)
        assert np.array_equal(np.array(a.Rs), np.array(b.Rs)) and np.array_equal(np.array(a.ts), np.array(b.ts))
        assert np.all(np.isfinite(np.array(a.Rs))) and np.abs(np.array(a.ts)).max() > 1.0      # (poses really came back)


def test_sweep_form_names_the_kernel_that_ran():
    """ccal_stats.sweep_form = FLOW | LOOP << 4 | WPI << 8 | SOLO << 12: what bench.py prints as the roofline's kernel name."""
    cases = [("C2", 3, (1 << 8) | (1 << 12)),      # <= 4 images: every trial in one launch
             ("C2", 37, 4 << 8),                    # a one-pass board, few images: four waves per image
             ("C2", 600, 1 << 8),                   # one wave per image, one group per workgroup
             ("C5", 7, 4 << 8)]                     # a multi-pass board, few images
    for name, n, want in cases:
        wl = syn.named_workload(name, n_images=n)
        with _solver(wl, REF) as s:
            _, r = s.calibrate()
        assert r["sweep_form"] == want, (name, n, hex(r["sweep_form"]))


def test_fit_many_equals_one_fit_at_a_time(golden_dir):
    """ccal_fit_many: the pipelines of several independent problems enqueued on their contexts' streams before any is waited for.
    Every context ends with the bits its own ccal_fit gives (exact accumulators: nothing depends on what else the GPU runs), a
    failing problem (a non-planar template) is reported in its slot and by the call without disturbing the others, a repeated
    call reuses the captured graphs, and fit_models returns what fit_model returns for each set of the reference's example."""
    from cameracalibrations_amd import _ffi as F, fit_many, fit_models
    specs = [("C1", 7, REF), ("C2", 40, REF), ("REF", 6, REF), ("C5", 5, REF), ("C2", 300, TIGHT), ("C1", 100, REF), ("C2", 3, REF),
             ("C5", 40, REF), ("REF", 129, REF), ("C2", 600, REF)]
    wls = [syn.named_workload(n, n_images=k, seed_offset=i) for i, (n, k, _) in enumerate(specs)]
    alone = []
    for wl, (_, _, crit) in zip(wls, specs):
        with _solver(wl, crit) as s:
            st = s.fit()
            alone.append((st, s.result()))
    solvers = [_solver(wl, crit) for wl, (_, _, crit) in zip(wls, specs)]
    try:
        for rep in range(3):
            sts = fit_many(solvers)
            for i, (s, st, (st0, r0)) in enumerate(zip(solvers, sts, alone)):
                r = s.result()
                assert st == st0 and r["iters"] == r0["iters"] and r["sweeps"] == r0["sweeps"]
                # the small fits share merged launches (also the first fits of new contexts): the workgroups of several contexts in
                # one grid, each looking its own fit up in a table (sweep_form bit 16); 600 images stay on their own
                small = specs[i][1] * (r["sweep_form"] >> 8 & 15) <= 512
                assert (r["sweep_form"] >> 16) == (1 if small else 0), (rep, i, hex(r["sweep_form"]))
                assert (r["sweep_form"] & 0xFFFF) == (r0["sweep_form"] & 0xFFFF)
                for k in ("intr", "rvecs", "tvecs"):
                    assert np.array_equal(r[k], r0[k]), k
                assert r["rms"] == r0["rms"]
        # one bad problem among them
        bent = wls[1].obj.copy()
        bent[3, 2] = 0.5
        solvers[1].set_problem(bent, wls[1].n_images, wls[1].image_size, flags=wls[1].flags)
        solvers[1].set_image_points(wls[1].img)
        with pytest.raises(F.CcalError) as e:
            fit_many(solvers)
        assert e.value.code == F.E_NONPLANAR and "context 1" in str(e.value)
        sts = fit_many(solvers, check=False)
        assert sts[1] == F.E_NONPLANAR and all(st == st0 for i, (st, (st0, _)) in enumerate(zip(sts, alone)) if i != 1)
        for i, (s, (_, r0)) in enumerate(zip(solvers, alone)):
            if i != 1:
                assert np.array_equal(s.result()["intr"], r0["intr"])
        with pytest.raises(F.CcalError) as e:
            fit_many([solvers[0], solvers[0]])
        assert e.value.code == F.E_ARG
        assert fit_many([]) == []
        # Contexts that live on between calls: new corners behind the same pointers, a new problem, calls of ccal_fit in between and
        # a changed pose-initialisation mode all have to end with the bits of a fresh context's own fit.
        solvers[1].set_problem(wls[1].obj, wls[1].n_images, wls[1].image_size, aspect=wls[1].aspect, flags=wls[1].flags,
                               max_iter=specs[1][2][0], eps=specs[1][2][1])
        solvers[1].set_image_points(wls[1].img)
        for rep in range(5):
            if rep == 2:                                  # other corners for two of them, one back on its own ccal_fit in between
                for i in (0, 5):
                    wls[i] = syn.named_workload(specs[i][0], n_images=specs[i][1], seed_offset=100 + i)
                    solvers[i].set_image_points(wls[i].img)
                    with _solver(wls[i], specs[i][2]) as s:
                        alone[i] = (s.fit(), s.result())
                solvers[3].fit()
            if rep == 3:                                  # a smaller problem in a context that has held a larger one
                wls[4] = syn.named_workload("C2", n_images=120, seed_offset=55)
                solvers[4].set_problem(wls[4].obj, 120, wls[4].image_size, aspect=wls[4].aspect, flags=wls[4].flags, max_iter=REF[0], eps=REF[1])
                solvers[4].set_image_points(wls[4].img)
                with _solver(wls[4], REF) as s:
                    alone[4] = (s.fit(), s.result())
            if rep == 4:
                solvers[2].set_pose_init(0)
                with _solver(wls[2], specs[2][2]) as s:
                    s.set_pose_init(0)
                    alone[2] = (s.fit(), s.result())
            sts = fit_many(solvers)
            for i, (s, st, (st0, r0)) in enumerate(zip(solvers, sts, alone)):
                r = s.result()
                assert st == st0 and r["iters"] == r0["iters"] and r["sweeps"] == r0["sweeps"], (rep, i)
                for k in ("intr", "rvecs", "tvecs"):
                    assert np.array_equal(r[k], r0[k]), (rep, i, k)
    finally:
        for s in solvers:
            s.close()
    with open(os.path.join(golden_dir, "example_corners.json")) as f:
        ex = json.load(f)
    from cameracalibrations_amd import board_template
    obj = board_template(ex["n_corners"])
    imgs = [np.array(m, dtype=np.float64) for m in ex["corners"]]
    sets = [imgs, imgs[:4], imgs[1:], imgs[::-1], imgs[:2], imgs[2:]]
    assert fit_models(ex["image_size_wh"], obj, [], ex["n_corners"], True, 1) == []
    assert len(fit_models(ex["image_size_wh"], obj, sets[:2], ex["n_corners"], True, 1)) == 2       # (two: streams of their own)
    many = fit_models(ex["image_size_wh"], obj, sets, ex["n_corners"], True, 1)                        # (six: merged launches)
    # one array [calibration, image, corner, 2] in: the same
    arr = np.stack([np.stack(imgs[:4]), np.stack(imgs[2:]), np.stack(imgs[1:5]), np.stack(imgs[:4][::-1]), np.stack(imgs[:4])])
    for a, sub in zip(fit_models(ex["image_size_wh"], obj, arr, ex["n_corners"], True, 1), arr):
        b = fit_model(ex["image_size_wh"], obj, list(sub), ex["n_corners"], True, 1)
        assert (a.frow, a.fcol, a.crow, a.ccol, a.k, a.rms, a.iters) == (b.frow, b.fcol, b.crow, b.ccol, b.k, b.rms, b.iters)
        assert np.array_equal(np.array(a.Rs), np.array(b.Rs)) and np.array_equal(np.array(a.ts), np.array(b.ts))
    for sub, a in zip(sets, many):
        b = fit_model(ex["image_size_wh"], obj, sub, ex["n_corners"], True, 1)
        assert (a.frow, a.fcol, a.crow, a.ccol, a.k, a.rms, a.iters) == (b.frow, b.fcol, b.crow, b.ccol, b.k, b.rms, b.iters)
        assert np.array_equal(np.array(a.Rs), np.array(b.Rs)) and np.array_equal(np.array(a.ts), np.array(b.ts))


def test_fit_models_orders_staged_inputs_of_mixed_sizes_before_the_merged_launches():
    """Two small calibrations (inputs staged in their pinned blocks and copied by a launch on their OWN streams, never waited for:
    fewer than four small ones share no copy launch) next to three of 80 images (too large for the pinned block, small enough to
    merge): five merge-able contexts, so all of them run in merged launches on the first context's stream -- which must wait for
    the other streams' pending copies.  The pool's contexts are re-used with other corners from call to call, so a launch that ran
    ahead of a copy would fit the previous call's corners.  Every calibration equals its own fit_model bit for bit, and the call's
    status is the union of the calibrations' warning bits."""
    from cameracalibrations_amd import _ffi as F, fit_models
    shapes = [6, 80, 5, 80, 80]
    for rep in range(6):
        wls = [syn.named_workload("REF", n_images=k, seed_offset=1000 + 17 * rep + i) for i, k in enumerate(shapes)]
        if rep % 2:
            wls = wls[::-1]
        wl0 = wls[0]
        many = fit_models(wl0.image_size, wl0.obj, [list(w.img) for w in wls], wl0.n_corners, True, wl0.aspect)
        for w, a in zip(wls, many):
            b = fit_model(w.image_size, w.obj, list(w.img), w.n_corners, True, w.aspect)
            assert (a.frow, a.fcol, a.crow, a.ccol, a.k, a.rms, a.iters) == (b.frow, b.fcol, b.crow, b.ccol, b.k, b.rms, b.iters), rep
            assert np.array_equal(np.array(a.Rs), np.array(b.Rs)) and np.array_equal(np.array(a.ts), np.array(b.ts)), rep
    # warning bits of several contexts: the call returns their union (one fit stopped by max_iter, the others clean)
    wls = [syn.named_workload("C2", n_images=12, seed_offset=300 + i) for i in range(5)]
    crits = [(1, 1e-12)] + [REF] * 4
    solvers = [_solver(w, c) for w, c in zip(wls, crits)]
    try:
        from cameracalibrations_amd import fit_many
        sts = fit_many(solvers, check=False)
        assert sts[0] == F.W_MAX_ITER and all(st == 0 for st in sts[1:])
        import ctypes as C
        ptrs = (C.c_void_p * len(solvers))(*[s._h for s in solvers])
        st_arr = (C.c_int32 * len(solvers))()
        rc = F.lib().ccal_fit_many(ptrs, len(solvers), st_arr)
        union = 0
        for v in st_arr:
            union |= int(v)
        assert rc == union == F.W_MAX_ITER
    finally:
        for s in solvers:
            s.close()


def test_fits_from_several_host_threads():
    """The reference runs its per-file work under Threads.@threads (test/runtests.jl:45-51); a caller may well fit from several
    host threads too.  Each thread has its own cached context (ccal_calibrate) and its own Solver; problems of changing sizes force
    graph re-builds while other threads allocate, copy and launch.  Every result must equal the single-threaded one."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = [("REF", 6), ("C1", 25), ("C2", 40), ("REF", 3), ("C5", 5), ("C2", 130), ("C1", 9), ("REF", 12)]
    wls = [syn.named_workload(n, n_images=k, seed_offset=7 * i) for i, (n, k) in enumerate(jobs)]

    def one(i):
        wl = wls[i % len(wls)]
        m = fit_model(wl.image_size, wl.obj, list(wl.img), wl.n_corners, bool(wl.flags & syn.FIX_K1) is False, wl.aspect)
        with _solver(wl, REF) as s:
            st = s.fit()
            r = s.result()
        return (m.frow, m.fcol, m.crow, m.ccol, m.k, m.rms, m.iters), (st, r["iters"], r["rms"], r["intr"].tobytes())
    serial = [one(i) for i in range(len(wls))]
    with ThreadPoolExecutor(max_workers=4) as pool:
        threaded = list(pool.map(one, range(4 * len(wls))))
    for i, res in enumerate(threaded):
        assert res == serial[i % len(wls)], i
