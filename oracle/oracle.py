"""ctypes binding of oracle/libccal_oracle.so (plain-C restatement, see ccal_oracle.h).

numpy conventions (C-order) mirror the Julia column-major arrays of the reference:
  obj  (P, 3)      <-> [3,P]      board template        (reference src/detect_fit.jl:69)
  img  (N, P, 2)   <-> [2,P,N]    observed corners      (reference src/detect_fit.jl:31)
  rvecs/tvecs (N, 3)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NINTR = 9
FLAG = dict(
    USE_INTRINSIC_GUESS=0x1, FIX_ASPECT_RATIO=0x2, FIX_PRINCIPAL_POINT=0x4, ZERO_TANGENT_DIST=0x8,
    FIX_FOCAL_LENGTH=0x10, FIX_K1=0x20, FIX_K2=0x40, FIX_K3=0x80, FIX_TANGENT_DIST=0x200000,
)
# reference flag word, src/detect_fit.jl:40
REF_FLAGS_DISTORTION = FLAG["ZERO_TANGENT_DIST"] | FLAG["FIX_K3"] | FLAG["FIX_K2"] | FLAG["FIX_ASPECT_RATIO"]
REF_FLAGS_NO_DISTORTION = REF_FLAGS_DISTORTION | FLAG["FIX_K1"]


class Trace(C.Structure):
    _fields_ = [("trial", C.c_int), ("accepted", C.c_int), ("lambda_lg10", C.c_int), ("iters", C.c_int),
                ("err", C.c_double), ("rel_step", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("rms", C.c_double), ("iters", C.c_int), ("j_sweeps", C.c_int), ("e_sweeps", C.c_int),
                ("lambda_lg10", C.c_int), ("n_trace", C.c_int), ("init_intr", C.c_double * NINTR),
                ("t_init_s", C.c_double), ("t_lm_s", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile).  CCAL_ORACLE_ASAN=1 selects the
    AddressSanitizer + UBSan build (`make asan`; run the tests with libasan preloaded)."""
    if os.environ.get("CCAL_ORACLE_ASAN"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
        return os.path.join(_HERE, "libccal_oracle_asan.so")
    so = os.path.join(_HERE, "libccal_oracle.so")
    src = os.path.join(_HERE, "ccal_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libccal_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        _LIB.orc_calibrate.restype = C.c_int
        _LIB.orc_calibrate.argtypes = [C.c_int, C.c_int, dp, dp, C.c_int, C.c_int, C.c_double, C.c_uint, C.c_int,
                                       C.c_double, C.c_int, dp, dp, dp, dp, dp, C.POINTER(Stats),
                                       C.POINTER(Trace), C.c_int, C.c_int]
        _LIB.orc_find_homography.restype = C.c_int
        _LIB.orc_init_intrinsics.restype = C.c_int
        _LIB.orc_find_extrinsics.restype = C.c_int
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def rodrigues(r, jac=False):
    r, rp = _d(r)
    R = np.zeros(9)
    J = np.zeros(27) if jac else None
    lib().orc_rodrigues(rp, _d(R)[1] if False else R.ctypes.data_as(C.POINTER(C.c_double)),
                        J.ctypes.data_as(C.POINTER(C.c_double)) if jac else None)
    return (R.reshape(3, 3), J.reshape(3, 9)) if jac else R.reshape(3, 3)


def rodrigues_inv(R):
    R, Rp = _d(R)
    r = np.zeros(3)
    lib().orc_rodrigues_inv(Rp, r.ctypes.data_as(C.POINTER(C.c_double)))
    return r


def project_points(obj, r, t, intr, aspect=0.0, jac=False):
    """-> proj (P,2) [, J (2P, 15) with columns r(3) t(3) fx fy cx cy k1 k2 p1 p2 k3]."""
    obj, op = _d(obj)
    P = obj.shape[0]
    r, rp = _d(r)
    t, tp = _d(t)
    intr, ip = _d(intr)
    proj = np.zeros((P, 2))
    pp = proj.ctypes.data_as(C.POINTER(C.c_double))
    if not jac:
        lib().orc_project_points(C.c_int(P), op, rp, tp, ip, C.c_double(aspect), pp, None, None, None, None, None)
        return proj
    dr, dt, df, dc, dk = (np.zeros((2 * P, n)) for n in (3, 3, 2, 2, 5))
    lib().orc_project_points(C.c_int(P), op, rp, tp, ip, C.c_double(aspect), pp,
                             *(a.ctypes.data_as(C.POINTER(C.c_double)) for a in (dr, dt, df, dc, dk)))
    return proj, np.hstack([dr, dt, df, dc, dk])


def find_homography(src_xy, dst_xy, refine=True):
    src, sp = _d(src_xy)
    dst, dp_ = _d(dst_xy)
    H = np.zeros(9)
    rc = lib().orc_find_homography(C.c_int(src.shape[0]), sp, dp_, H.ctypes.data_as(C.POINTER(C.c_double)),
                                   C.c_int(int(refine)))
    if rc:
        raise RuntimeError(f"orc_find_homography failed: {rc}")
    return H.reshape(3, 3)


def init_intrinsics(obj, img, width, height, aspect=0.0):
    obj, op = _d(obj)
    img, ip = _d(img)
    N, P = img.shape[0], img.shape[1]
    K4 = np.zeros(4)
    allH = np.zeros((N, 9))
    rc = lib().orc_init_intrinsics(C.c_int(N), C.c_int(P), op, ip, C.c_int(width), C.c_int(height),
                                   C.c_double(aspect), K4.ctypes.data_as(C.POINTER(C.c_double)),
                                   allH.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(f"orc_init_intrinsics failed: {rc}")
    return K4, allH.reshape(N, 3, 3)


def find_extrinsics(obj, img_i, intr):
    obj, op = _d(obj)
    img_i, ip = _d(img_i)
    intr, np_ = _d(intr)
    r, t = np.zeros(3), np.zeros(3)
    rc = lib().orc_find_extrinsics(C.c_int(obj.shape[0]), op, ip, np_, r.ctypes.data_as(C.POINTER(C.c_double)),
                                   t.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(f"orc_find_extrinsics failed: {rc}")
    return r, t


def image_gram(obj, img_i, r, t, intr, aspect=0.0):
    obj, op = _d(obj)
    img_i, ip = _d(img_i)
    r, rp = _d(r)
    t, tp = _d(t)
    intr, np_ = _d(intr)
    G = np.zeros(256)
    lib().orc_image_gram(C.c_int(obj.shape[0]), op, ip, rp, tp, np_, C.c_double(aspect),
                         G.ctypes.data_as(C.POINTER(C.c_double)))
    return G.reshape(16, 16)


def calibrate(obj, img, image_size, aspect=1.0, flags=REF_FLAGS_DISTORTION, max_iter=30, eps=1e-3,
              solver="schur", init_intr=None, init_rt=None, nthreads=1, max_trace=4096):
    """calibrateCamera restatement.  image_size = (width, height) as OpenCV sees it, i.e.
    (rows, cols) of the Julia image (reference src/detect_fit.jl:32).  Returns a dict."""
    obj, op = _d(obj)
    img, ip = _d(img)
    N, P = img.shape[0], img.shape[1]
    assert obj.shape == (P, 3) and img.shape == (N, P, 2)
    intr = np.zeros(NINTR)
    rv, tv = np.zeros((N, 3)), np.zeros((N, 3))
    st = Stats()
    tr = (Trace * max_trace)()
    ii = _d(init_intr)[1] if init_intr is not None else None
    keep = _d(init_intr)[0] if init_intr is not None else None  # noqa: F841
    if init_intr is not None:
        keep, ii = _d(init_intr)
    if init_rt is not None:
        keep2, ir = _d(init_rt)
    else:
        ir = None
    dp = C.POINTER(C.c_double)
    rc = lib().orc_calibrate(N, P, op, ip, int(image_size[0]), int(image_size[1]), float(aspect), int(flags),
                             int(max_iter), float(eps), 1 if solver == "schur" else 0, ii, ir,
                             intr.ctypes.data_as(dp), rv.ctypes.data_as(dp), tv.ctypes.data_as(dp),
                             C.byref(st), tr, max_trace, int(nthreads))
    if rc:
        raise RuntimeError(f"orc_calibrate failed: {rc}")
    trace = [dict(trial=t.trial, accepted=t.accepted, lambda_lg10=t.lambda_lg10, iters=t.iters, err=t.err,
                  rel_step=t.rel_step) for t in tr[: st.n_trace]]
    return dict(rms=st.rms, intr=intr, K4=intr[:4].copy(), dist5=intr[4:].copy(), rvecs=rv, tvecs=tv,
                iters=st.iters, j_sweeps=st.j_sweeps, e_sweeps=st.e_sweeps, lambda_lg10=st.lambda_lg10,
                init_intr=np.array(st.init_intr[:]), t_init_s=st.t_init_s, t_lm_s=st.t_lm_s, trace=trace)


def map_forward(xyz, f, c, k, r, t, checker_size=1.0):
    xyz, xp = _d(np.atleast_2d(xyz))
    out = np.zeros((xyz.shape[0], 2))
    lib().orc_map_forward(C.c_int(xyz.shape[0]), xp, _d(f)[1], _d(c)[1], C.c_double(k), _d(r)[1], _d(t)[1],
                          C.c_double(checker_size), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def map_inverse(rc, f, c, k, r, t, checker_size=1.0):
    rc, rp = _d(np.atleast_2d(rc))
    out = np.zeros((rc.shape[0], 3))
    lib().orc_map_inverse(C.c_int(rc.shape[0]), rp, _d(f)[1], _d(c)[1], C.c_double(k), _d(r)[1], _d(t)[1],
                          C.c_double(checker_size), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def errors(n_corners, obj_scaled, img, f, c, k, rvecs, tvecs, checker_size, inv_pts=None):
    """calculate_errors restatement -> dict(reprojection, projection, distance, inverse)."""
    obj, op = _d(obj_scaled)
    img, ip = _d(img)
    rv, rvp = _d(rvecs)
    tv, tvp = _d(tvecs)
    N = img.shape[0]
    S = 0
    pp = None
    if inv_pts is not None:
        pts, pp = _d(inv_pts)
        S = pts.shape[1]
    out = np.zeros(4)
    lib().orc_errors(C.c_int(N), C.c_int(n_corners[0]), C.c_int(n_corners[1]), op, ip, _d(f)[1], _d(c)[1],
                     C.c_double(k), rvp, tvp, C.c_double(checker_size), C.c_int(S), pp,
                     out.ctypes.data_as(C.POINTER(C.c_double)))
    return dict(reprojection=out[0], projection=out[1], distance=out[2], inverse=out[3])
