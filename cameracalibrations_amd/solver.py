"""Host-side handle on the device fit (thin wrapper over the C ABI, include/ccal.h).

`Solver` exposes the stages of OpenCV's calibrateCamera that the reference reaches through
its single FFI call (reference src/detect_fit.jl:47) so that tests can compare each stage
with the oracle; `fit_model` below mirrors the reference's own wrapper function
(src/detect_fit.jl:27-61): same arguments, same NamedTuple-like result.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _ffi as F


class Solver:
    """One device context (`ccal_ctx`).  Not thread-safe; use one per host thread."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        self.device = int(device)
        F.check(F.lib().ccal_create(int(device), C.byref(self._h)))
        self._keep = []          # arrays / callbacks that must outlive the device work
        self.N = self.P = 0
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if self._h:
            F.lib().ccal_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration ------------------------------------------------------------------
    def set_stream(self, stream):
        """stream: a hipStream_t handle as int (e.g. torch.cuda.current_stream().cuda_stream)."""
        F.check(F.lib().ccal_set_stream(self._h, C.c_void_p(int(stream) if stream else None)))

    def set_allreduce(self, fn, world_size, rank, n_images_global):
        """fn(device_ptr:int, n:int, stream:int) -> 0 sums n doubles in place across ranks."""
        def _tramp(_user, buf, n, stream):
            try:
                return int(fn(int(buf or 0), int(n), int(stream or 0)) or 0)
            except Exception:  # never unwind through the C frame
                import traceback
                traceback.print_exc()
                return -1
        cb = F.ALLREDUCE_FN(_tramp)
        self._keep.append(cb)
        F.check(F.lib().ccal_set_allreduce(self._h, cb, None, int(world_size), int(rank), int(n_images_global)))

    def init_rccl(self, world_size, rank, n_images_global, broadcast_bytes):
        """Native RCCL all-reduce inside the library.  `broadcast_bytes(b: bytes|None) -> bytes`
        ships rank 0's 128-byte unique id to every rank (e.g. through torch.distributed)."""
        buf = C.create_string_buffer(128)
        if rank == 0:
            F.check(F.lib().ccal_comm_unique_id(buf))
        ident = broadcast_bytes(buf.raw if rank == 0 else None)
        assert len(ident) == 128
        F.check(F.lib().ccal_comm_init_rank(self._h, int(world_size), int(rank), ident, int(n_images_global)))

    def init_p2p(self, world_size, rank, n_images_global, allgather_bytes, shared_device=None):
        """Direct xGMI row exchange between the GPUs of one node (no collective library).
        `allgather_bytes(b: bytes) -> list[bytes]` returns every rank's bytes in rank order (e.g. through
        torch.distributed): one 80-byte blob per rank (ccal_p2p_export2: the inbox handle and the id of the rank's device).
        The library itself notices two ranks with the same device id (a rehearsal on a one-GPU box) and then takes the
        one-workgroup exchange, so that a waiting rank holds one compute unit and not a grid; `shared_device` overrides the
        detection (False: the fused form whatever was detected).  Returns True when the one-workgroup exchange is in use."""
        blob = C.create_string_buffer(80)
        rc = F.lib().ccal_p2p_export2(self._h, int(world_size), int(rank), blob)
        got = allgather_bytes(blob.raw if rc == 0 else bytes(80))   # the gather is collective: always join it
        F.check(rc)
        assert len(got) == world_size and all(len(h) == 80 for h in got)
        if any(h[:64] == bytes(64) for h in got):
            raise RuntimeError("init_p2p: a peer rank could not export its inbox")
        F.check(F.lib().ccal_p2p_connect2(self._h, b"".join(got), int(n_images_global)))
        if shared_device is not None:
            F.check(F.lib().ccal_p2p_set_exchange(self._h, 1 if shared_device else 2))
            return bool(shared_device)
        return bool(F.lib().ccal_p2p_shared_device(self._h))

    def p2p_selftest(self, rounds=8):
        """Exchanges known rows `rounds` times and checks the sums (all ranks call it together)."""
        F.check(F.lib().ccal_p2p_selftest(self._h, int(rounds)))

    def set_problem(self, obj, n_images, image_size, aspect=1.0, flags=0, max_iter=30, eps=1e-3):
        """obj (P,3); image_size = (width, height) as OpenCV sees it (src/detect_fit.jl:32)."""
        obj, op = F.f32(obj)
        assert obj.ndim == 2 and obj.shape[1] == 3
        self.P, self.N = obj.shape[0], int(n_images)
        F.check(F.lib().ccal_set_problem(self._h, op, self.P, self.N, int(image_size[0]), int(image_size[1]),
                                         float(aspect), int(flags), int(max_iter), float(eps)))

    def set_image_points(self, img):
        """img (N,P,2) float32 host array (copied), or a torch CUDA tensor (used in place)."""
        if hasattr(img, "data_ptr") and getattr(img, "is_cuda", False):
            assert tuple(img.shape) == (self.N, self.P, 2) and img.is_contiguous() and str(img.dtype) == "torch.float32"
            self._keep.append(img)
            F.check(F.lib().ccal_set_image_points_device(self._h, C.c_void_p(img.data_ptr())))
            return
        img, ip = F.f32(img)
        assert img.shape == (self.N, self.P, 2), (img.shape, (self.N, self.P, 2))
        F.check(F.lib().ccal_set_image_points(self._h, ip))

    # -- stages -------------------------------------------------------------------------
    def init_intrinsics(self):
        F.check(F.lib().ccal_init_intrinsics(self._h))

    def set_intrinsics(self, intr):
        F.check(F.lib().ccal_set_intrinsics(self._h, F.f64(intr)[1]))

    def init_extrinsics(self):
        F.check(F.lib().ccal_init_extrinsics(self._h))

    def set_extrinsics(self, rt):
        rt, rp = F.f64(rt)
        assert rt.shape == (self.N, 6)
        F.check(F.lib().ccal_set_extrinsics(self._h, rp))

    def get_init(self, extrinsics=True):
        intr, ip = F.out64(F.NINTR)
        rt, rp = F.out64(self.N, 6)
        F.check(F.lib().ccal_get_init(self._h, ip, rp if extrinsics else None))
        return intr, (rt if extrinsics else None)

    def debug_homographies(self):
        """(N, 2, 3, 3): per image the polished board -> pixel homography and the DLT's result before its polish (intrinsics stage)"""
        H, hp = F.out64(self.N, 18)
        F.check(F.lib().ccal_debug_get_homographies(self._h, hp))
        return H.reshape(self.N, 2, 3, 3)

    def set_sweep_timing(self, enable=True):
        F.check(F.lib().ccal_set_sweep_timing(self._h, int(bool(enable))))

    def set_pose_init(self, mode):
        """0 = OpenCV's own sequence for the pose stage's homography, 1 (default) = the same with the DLT reused from the
        intrinsics stage, 2 = the polished pixel-space homography decomposed directly (faster; see include/ccal.h)"""
        F.check(F.lib().ccal_set_pose_init(self._h, int(mode)))

    def set_graph(self, enable=True):
        """replay a solve as one hipGraph (default) or launch it kernel by kernel"""
        F.check(F.lib().ccal_set_graph(self._h, int(bool(enable))))

    def lm_run(self):
        return F.check(F.lib().ccal_lm_run(self._h))

    def fit(self):
        """initialisation + LM as one pipeline with a single wait (ccal_fit); returns the LM's status"""
        return F.check(F.lib().ccal_fit(self._h))

    def result(self):
        intr, ip = F.out64(F.NINTR)
        rv, rvp = F.out64(self.N, 3)
        tv, tvp = F.out64(self.N, 3)
        st = F.Stats()
        F.check(F.lib().ccal_get_result(self._h, ip, rvp, tvp, C.byref(st)))
        return dict(intr=intr, K4=intr[:4].copy(), dist5=intr[4:].copy(), rvecs=rv, tvecs=tv, **st.asdict())

    def trace(self, max_entries=4096):
        buf = (F.Trace * max_entries)()
        n = C.c_int32(0)
        F.check(F.lib().ccal_get_trace(self._h, buf, max_entries, C.byref(n)))
        return [dict(err=t.err, rel_step=t.rel_step, accepted=t.accepted, lambda_lg10=t.lambda_lg10, iters=t.iters)
                for t in buf[: n.value]]

    def calibrate(self, init_intr=None, init_rt=None):
        """init (device) -> LM (device) -> result dict; returns (status, result)."""
        if init_intr is not None:
            self.set_intrinsics(init_intr)
        else:
            self.init_intrinsics()
        if init_rt is not None:
            self.set_extrinsics(init_rt)
        else:
            self.init_extrinsics()
        status = self.lm_run()
        return status, self.result()

    # -- maps / metrics -----------------------------------------------------------------
    def project(self, xyz, f, c, k, rvec, tvec, checker_size=1.0):
        xyz, xp = F.f64(np.atleast_2d(xyz))
        out, o = F.out64(xyz.shape[0], 2)
        F.check(F.lib().ccal_project(self._h, xp, xyz.shape[0], F.f64(f)[1], F.f64(c)[1], float(k), F.f64(rvec)[1],
                                     F.f64(tvec)[1], float(checker_size), o))
        return out

    def unproject(self, rc, f, c, k, rvec, tvec, checker_size=1.0):
        rc, rp = F.f64(np.atleast_2d(rc))
        out, o = F.out64(rc.shape[0], 3)
        F.check(F.lib().ccal_unproject(self._h, rp, rc.shape[0], F.f64(f)[1], F.f64(c)[1], float(k), F.f64(rvec)[1],
                                       F.f64(tvec)[1], float(checker_size), o))
        return out

    def project_many(self, xyz, f, c, k, rvecs, tvecs, checker_size=1.0):
        """xyz (M,3): one template through every pose, or (N,M,3): image n through pose n  ->  (N,M,2)."""
        rv, rvp = F.f64(np.atleast_2d(rvecs))
        tv, tvp = F.f64(np.atleast_2d(tvecs))
        xyz, xp = F.f64(xyz)
        N = rv.shape[0]
        per_image = xyz.ndim == 3
        M = xyz.shape[-2]
        assert xyz.shape[-1] == 3 and tv.shape == rv.shape == (N, 3) and (not per_image or xyz.shape[0] == N)
        out, o = F.out64(N, M, 2)
        F.check(F.lib().ccal_project_many(self._h, xp, M, N, int(per_image), F.f64(f)[1], F.f64(c)[1], float(k), rvp, tvp,
                                          float(checker_size), o))
        return out

    def unproject_many(self, rc, f, c, k, rvecs, tvecs, checker_size=1.0):
        """rc (N,M,2): image n through the inverse map of pose n  ->  (N,M,3) on the board plane."""
        rv, rvp = F.f64(np.atleast_2d(rvecs))
        tv, tvp = F.f64(np.atleast_2d(tvecs))
        rc, rp = F.f64(rc)
        N, M = rc.shape[0], rc.shape[1]
        assert rc.shape == (N, M, 2) and tv.shape == rv.shape == (N, 3)
        out, o = F.out64(N, M, 3)
        F.check(F.lib().ccal_unproject_many(self._h, rp, M, N, F.f64(f)[1], F.f64(c)[1], float(k), rvp, tvp,
                                            float(checker_size), o))
        return out

    def errors(self, n_corners, obj_scaled, img, f, c, k, rvecs, tvecs, checker_size, inv_pts=None):
        obj, op = F.f64(obj_scaled)
        img, ip = F.f64(img)
        N = img.shape[0]
        S, pp = 0, None
        if inv_pts is not None:
            pts, pp = F.f64(inv_pts)
            S = pts.shape[1]
        out, o = F.out64(4)
        F.check(F.lib().ccal_errors(self._h, N, int(n_corners[0]), int(n_corners[1]), op, ip, F.f64(f)[1], F.f64(c)[1],
                                    float(k), F.f64(rvecs)[1], F.f64(tvecs)[1], float(checker_size), S, pp, o))
        return dict(reprojection=out[0], projection=out[1], distance=out[2], inverse=out[3])

    # -- diagnostics --------------------------------------------------------------------
    def image_grams(self, intr, rt):
        rt, rp = F.f64(rt)
        G, gp = F.out64(self.N, 16, 16)
        F.check(F.lib().ccal_debug_image_grams(self._h, F.f64(intr)[1], rp, gp))
        return G

    def debug_acc_sum(self, rows, replicas=8):
        """Diagnostics: column sums of `rows` (n, 68) through the device's exact accumulators -> sums[68]."""
        rows, rp = F.f64(np.atleast_2d(rows))
        assert rows.shape[1] == 68
        out, o = F.out64(68)
        F.check(F.lib().ccal_debug_acc_sum(self._h, rp, rows.shape[0], int(replicas), o))
        return out

    def debug_p2p_world(self, rows, seq0=0):
        """Diagnostics: rows (rounds, world, 68) through `world` workgroups acting as the ranks of a node (ccal_debug_p2p_world)
        -> what every rank holds after every exchange, (rounds, world, 68)."""
        rows, rp = F.f64(rows)
        rounds, world, n = rows.shape
        assert n == 68
        out, o = F.out64(rounds, world, 68)
        F.check(F.lib().ccal_debug_p2p_world(self._h, int(world), int(rounds), int(seq0) & 0xFFFFFFFF, rp, o))
        return out

    def time_sweep(self, reps=50):
        v = C.c_double(0)
        F.check(F.lib().ccal_time_sweep(self._h, int(reps), C.byref(v)))
        return v.value


def device_count():
    return F.lib().ccal_device_count()


# reference flag word, src/detect_fit.jl:40
def reference_flags(with_distortion=True):
    return (F.ZERO_TANGENT_DIST | F.FIX_K3 | F.FIX_K2 | (0 if with_distortion else F.FIX_K1) | F.FIX_ASPECT_RATIO)


CRITERIA = (30, 0.001)   # TermCriteria(EPS + MAX_ITER, 30, 0.001), src/CameraCalibrations.jl:16


def calibrate_multi(obj, img, image_size, devices, aspect=1.0, flags=0, max_iter=CRITERIA[0], eps=CRITERIA[1]):
    """One fit over several GPUs from this one host thread (ccal_calibrate_multi): images are sharded
    contiguously over `devices`.  obj (P,3), img (N,P,2) host arrays -> (status, dict like Solver.result())."""
    obj, op = F.f32(obj)
    img, ip = F.f32(img)
    N, P = img.shape[0], obj.shape[0]
    assert img.shape == (N, P, 2)
    devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    K4, dist5 = np.zeros(4), np.zeros(5)
    rv, tv = np.zeros((N, 3)), np.zeros((N, 3))
    rms, iters = C.c_double(0), C.c_int32(0)
    st = F.Stats()
    dp = F._dp
    status = F.check(F.lib().ccal_calibrate_multi(op, ip, P, N, int(image_size[0]), int(image_size[1]), float(aspect),
                                                  int(flags), int(max_iter), float(eps), len(devices), devs,
                                                  K4.ctypes.data_as(dp), dist5.ctypes.data_as(dp), rv.ctypes.data_as(dp),
                                                  tv.ctypes.data_as(dp), C.byref(rms), C.byref(iters), C.byref(st)))
    d = st.asdict()
    d.update(intr=np.r_[K4, dist5], rvecs=rv, tvecs=tv, rms=rms.value, iters=iters.value)
    return status, d


def fit_many(solvers, check=True):
    """`Solver.fit()` of several INDEPENDENT problems, one Solver each (ccal_fit_many): every pipeline is enqueued on its context's
    stream before any is waited for, so the GPU overlaps them.  Returns the per-solver statuses; results through each
    `Solver.result()`.  check=False: failures are reported in the list (negative codes) instead of raised."""
    n = len(solvers)
    hs = (C.c_void_p * n)(*[s._h for s in solvers])
    st = (C.c_int32 * n)()
    rc = F.lib().ccal_fit_many(hs, n, st)
    if check:
        F.check(rc)
    return [int(v) for v in st]


def fit_models(sz, objpoints, imgpointsss, n_corners, with_distortion, aspect, device=0):
    """`fit_model` (src/detect_fit.jl:27-61) for several calibrations at once -- the loop a caller with many cameras runs around
    src/buildcalibrations.jl:12-26: imgpointsss = one list / array of per-image corners PER calibration, or one array
    [calibration, image, corner, 2] when all have the same number of images (the template, image size, flags and aspect are shared, as
    one caller's are).  One call of ccal_calibrate_many: host arrays in and out, the library's own
    pool of contexts, the fits through ccal_fit_many (small ones share merged launches).  Returns a list of fit_model's namespaces,
    the same values fit_model returns for each set on its own."""
    P = int(n_corners[0]) * int(n_corners[1])
    obj = np.ascontiguousarray(objpoints, dtype=np.float32).reshape(P, 3)
    stacked = None
    if isinstance(imgpointsss, np.ndarray) and imgpointsss.ndim >= 3:     # one array of all calibrations (the same number of images each)
        stacked = np.ascontiguousarray(imgpointsss, dtype=np.float32).reshape(imgpointsss.shape[0], -1, P, 2)
        imgs = list(stacked)
    else:
        imgs = []
        for pts in imgpointsss:
            try:
                imgs.append(np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, P, 2))
            except ValueError:
                imgs.append(np.stack([np.asarray(m, dtype=np.float32).reshape(P, 2) for m in pts]))
    n = len(imgs)
    if stacked is not None and n > 0:                                      # ... one result block, addresses by arithmetic
        N0 = stacked.shape[1]
        block = np.zeros((n, 11 + 6 * N0))
        outs = list(block)
        Ns = (C.c_int32 * n)(*([N0] * n))
        b_in, s_in, b_out, s_out = stacked.ctypes.data, stacked.strides[0], block.ctypes.data, block.strides[0]
        ip = (C.c_void_p * n)(*range(b_in, b_in + n * s_in, s_in))
        op = (C.c_void_p * n)(*range(b_out, b_out + n * s_out, s_out))
    else:
        outs = [np.zeros(11 + 6 * a.shape[0]) for a in imgs]
        Ns = (C.c_int32 * n)(*[a.shape[0] for a in imgs])
        ip = (C.c_void_p * n)(*[a.ctypes.data for a in imgs])
        op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    st = (C.c_int32 * n)()
    F.check(F.lib().ccal_calibrate_many(F.f32(obj)[1], P, n, ip, Ns, int(sz[0]), int(sz[1]), float(aspect), reference_flags(with_distortion),
                                        CRITERIA[0], CRITERIA[1], int(device), op, st))
    res = []
    for o, a in zip(outs, imgs):
        N = a.shape[0]
        res.append(SimpleNamespace(k=float(o[4]), Rs=list(o[11:11 + 3 * N].reshape(N, 3)), ts=list(o[11 + 3 * N:].reshape(N, 3)),
                                   frow=float(o[0]), fcol=float(o[1]), crow=float(o[2]), ccol=float(o[3]), rms=float(o[9]), iters=int(o[10]),
                                   stats={"rms": float(o[9]), "iters": int(o[10])}))
    return res


def debug_multi_set_epoch(devices, epoch):
    """Diagnostics: set the solve sequence number of every shard of this thread's cached `calibrate_multi` device set."""
    devs = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    F.check(F.lib().ccal_debug_multi_set_epoch(len(devices), devs, int(epoch) & 0xFFFFFFFF))


def fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect, device=0):
    """Drop-in for the reference's `fit_model` (src/detect_fit.jl:27-61).

    sz = size of the Julia image (rows, cols) == OpenCV (width, height); objpoints = the
    n1 x n2 x 1 template of XYZ (any array reshaping to (P,3), first dim fastest);
    imgpointss = list of per-image corner arrays (each reshaping to (P,2));
    returns a namespace (k, Rs, ts, frow, fcol, crow, ccol) like the NamedTuple at :60.
    """
    P = int(n_corners[0]) * int(n_corners[1])
    obj = np.ascontiguousarray(objpoints, dtype=np.float32).reshape(P, 3)
    try:                                             # one array of all images' corners, or same-shaped per-image arrays: one conversion
        img = np.ascontiguousarray(imgpointss, dtype=np.float32).reshape(-1, P, 2)
    except ValueError:
        img = np.stack([np.asarray(m, dtype=np.float32).reshape(P, 2) for m in imgpointss])
    N = img.shape[0]
    res = np.zeros(10 + 6 * N)                       # K4 | dist5 | rms | rvecs [N,3] | tvecs [N,3]: one allocation, one address
    K4, dist5, rv, tv = res[0:4], res[4:9], res[10:10 + 3 * N].reshape(N, 3), res[10 + 3 * N:].reshape(N, 3)
    iters = C.c_int32(0)
    st = F.Stats()
    base = res.ctypes.data
    F.check(F.raw("ccal_calibrate")(obj.ctypes.data, img.ctypes.data, P, N, int(sz[0]), int(sz[1]), float(aspect),
                                    reference_flags(with_distortion), CRITERIA[0], CRITERIA[1], int(device),
                                    base, base + 32, base + 80, base + 80 + 24 * N, base + 72, C.addressof(iters), C.addressof(st)))
    rms = SimpleNamespace(value=float(res[9]))
    return SimpleNamespace(k=float(dist5[0]), Rs=list(rv), ts=list(tv),      # per-image vectors (rows of the two result arrays)
                           frow=float(K4[0]), fcol=float(K4[1]), crow=float(K4[2]), ccol=float(K4[3]),
                           rms=rms.value, iters=iters.value, stats=st.asdict())
