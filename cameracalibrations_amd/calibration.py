"""Host-side mirror of the reference's public interface for the fit path.

Same names, argument meaning and error behaviour as the Julia package
(reference src/CameraCalibrations.jl:18 exports `fit, Calibration, RowCol, XYZ, rectification`):

    c, eps = fit_corners(imgpointss, n_corners, checker_size, sz)     # corner arrays in
    c(xyz, i) / c(rowcol, i) / c(x)                                   # src/meta.jl:82,88,90-93
    rectification(c, i)                                                # src/meta.jl:99-103

All arithmetic (fit, maps, error metrics) runs in the HIP library through the C ABI; this
module only packs arguments.  Corner detection (reference src/detect_fit.jl:5-21) uses the device
detector of csrc/ccal_detect.hip unless a `detector` callable is passed to `fit`.
"""
import re
import threading
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import solver as _solver


def RowCol(row, col):
    """2-vector (row, col) == (x_cv, y_cv) (src/meta.jl:9; layout: SURVEY.md row A2)."""
    return np.array([row, col], dtype=np.float64)


def XYZ(x, y, z):
    """3-vector real-world coordinate (src/meta.jl:15)."""
    return np.array([x, y, z], dtype=np.float64)


_tls = threading.local()


def _shared_solver(device=0):
    """One device context per host thread (a `ccal_ctx` serves one thread at a time; the reference's
    FFI wrapper is exercised under Threads.@threads, test/runtests.jl:45-51)."""
    cache = getattr(_tls, "solvers", None)
    if cache is None:
        cache = _tls.solvers = {}
    s = cache.get(device)
    if s is None:
        s = cache[device] = _solver.Solver(device)
    return s


@dataclass
class Calibration:
    """Mirror of the reference struct (src/meta.jl:17-25): intrinsic = (f, c) diagonal affine map,
    extrinsics = per-image (rvec, tvec), scale = 1/checker_size, single radial coefficient k."""
    f: np.ndarray                      # (frow, fcol)   intrinsic.linear diagonal
    c: np.ndarray                      # (crow, ccol)   intrinsic.translation
    rvecs: np.ndarray                  # (N,3) RotationVec per image
    tvecs: np.ndarray                  # (N,3)
    checker_size: float                # scale = LinearMap(I / checker_size), src/buildcalibrations.jl:4
    k: float
    files: List[str] = field(default_factory=list)
    device: int = 0

    def _index(self, extrinsic_index):
        if extrinsic_index is None:
            # src/meta.jl:90-93: first file whose name matches r"extrinsic"
            hits = [i for i, f in enumerate(self.files) if re.search(r"extrinsic", f)]
            if not hits:
                raise KeyError("no file name matches r\"extrinsic\" (reference: `nothing` index error)")
            return hits[0]
        i = int(extrinsic_index) - 1       # the reference API is 1-based
        if not 0 <= i < len(self.rvecs):
            raise IndexError(f"extrinsic index {extrinsic_index} out of range 1:{len(self.rvecs)}")
        return i

    def __call__(self, x, extrinsic_index=None):
        """c(xyz, i) -> RowCol (real2image[i]); c(rowcol, i) -> XYZ (image2real[i]).
        Batched: (M,3) -> (M,2), (M,2) -> (M,3)."""
        i = self._index(extrinsic_index)
        if isinstance(x, (list, tuple)) and len(x) and np.ndim(x[0]) == 1:
            x = np.array(x, dtype=np.float64)              # a sequence of RowCol / XYZ points: one call for all of them
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        pts = np.atleast_2d(x)
        s = _shared_solver(self.device)
        if pts.shape[1] == 3:
            out = s.project(pts, self.f, self.c, self.k, self.rvecs[i], self.tvecs[i], self.checker_size)
        elif pts.shape[1] == 2:
            out = s.unproject(pts, self.f, self.c, self.k, self.rvecs[i], self.tvecs[i], self.checker_size)
        else:
            raise TypeError("expected an XYZ (3-vector) or RowCol (2-vector)")
        return out[0] if single else out


class _MapBatch:
    """Buffer of point-wise map calls (Calibration.batch): `b(x, i)` queues one point and hands back the array its result
    will be written to (a row of a buffer that is filled when the queue is evaluated); leaving the `with` block (or
    `flush()`) evaluates everything queued in at most two device calls (ccal_project_many / ccal_unproject_many with one
    point per pose entry).  A loop over points like the reference's round-trip metric (src/buildcalibrations.jl:53-58)
    then costs two calls instead of two per point."""
    _CHUNK = 256

    def __init__(self, c):
        self._c = c
        self._n_poses = len(c.rvecs)
        self._first = c._index(None) if any("extrinsic" in f for f in c.files) else None
        # width of the input -> [points, pose indices, result chunks (CHUNK x output width), queued count]
        self._q = {3: [[], [], [], 0], 2: [[], [], [], 0]}

    def __call__(self, x, extrinsic_index=None):
        if extrinsic_index is None:
            i = self._c._index(None) if self._first is None else self._first
        else:
            i = int(extrinsic_index) - 1                   # the reference API is 1-based
            if not 0 <= i < self._n_poses:
                raise IndexError(f"extrinsic index {extrinsic_index} out of range 1:{self._n_poses}")
        try:
            q = self._q[len(x)]
        except (KeyError, TypeError):
            raise TypeError("expected an XYZ (3-vector) or RowCol (2-vector)") from None
        n = q[3]
        if n % self._CHUNK == 0 and n // self._CHUNK == len(q[2]):
            q[2].append(np.full((self._CHUNK, 5 - len(x)), np.nan))
        q[0].append(x)
        q[1].append(i)
        q[3] = n + 1
        return q[2][n // self._CHUNK][n % self._CHUNK]

    def flush(self):
        c = self._c
        s = _shared_solver(c.device)
        for w, fn in ((3, s.project_many), (2, s.unproject_many)):
            pts, idx, chunks, n = self._q[w]
            if not n:
                continue
            x = np.array(pts, dtype=np.float64).reshape(n, 1, w)
            res = fn(x, c.f, c.c, c.k, c.rvecs[idx], c.tvecs[idx], c.checker_size)[:, 0, :]
            for k, ch in enumerate(chunks):                  # rows already handed out are views of these chunks
                m = min(self._CHUNK, n - k * self._CHUNK)
                if m > 0:
                    ch[:m] = res[k * self._CHUNK:k * self._CHUNK + m]
            self._q[w] = [[], [], [], 0]

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *_):
        if exc_type is None:
            self.flush()


def _batch(self):
    """`with c.batch() as b: outs = [b(rc, i) for rc in pts]` -- point-wise calls, evaluated together (see _MapBatch)."""
    return _MapBatch(self)


def _map_all(self, x):
    """Image n through pose n for all images in one call: x (N,M,3) -> (N,M,2) (real2image), (N,M,2) -> (N,M,3)
    (image2real), or one (M,3) template through every pose (`c.(objpoints, i)` for all i, src/buildcalibrations.jl:29)."""
    x = np.asarray(x, dtype=np.float64)
    s = _shared_solver(self.device)
    if x.shape[-1] == 3:
        return s.project_many(x, self.f, self.c, self.k, self.rvecs, self.tvecs, self.checker_size)
    if x.ndim == 3 and x.shape[-1] == 2:
        return s.unproject_many(x, self.f, self.c, self.k, self.rvecs, self.tvecs, self.checker_size)
    raise TypeError("expected (N,M,3) / (M,3) XYZ or (N,M,2) RowCol points")


Calibration.batch = _batch
Calibration.map_all = _map_all


def rectification(c: Calibration, extrinsic_index=None) -> Callable:
    """pop o image2real[i] (src/meta.jl:99-103): RowCol -> xy on the board plane."""
    i1 = None if extrinsic_index is None else extrinsic_index

    def f(rc):
        out = c(rc, i1)
        return out[..., :2]
    return f


def obj2img(Rs, ts, frow, fcol, crow, ccol, checker_size):
    """src/buildcalibrations.jl:1-6 -- pack the solver output into the map parameters."""
    return (np.array([frow, fcol], dtype=np.float64), np.array([crow, ccol], dtype=np.float64),
            np.asarray(Rs, dtype=np.float64).reshape(-1, 3), np.asarray(ts, dtype=np.float64).reshape(-1, 3),
            float(checker_size))


def board_template(n_corners):
    """XYZ.(Tuple.(CartesianIndices((0:n1-1, 0:n2-1, 0:0)))) (src/detect_fit.jl:69), as (P,3)."""
    n1, n2 = int(n_corners[0]), int(n_corners[1])
    p = np.arange(n1 * n2)
    return np.stack([p % n1, p // n1, np.zeros_like(p)], axis=1).astype(np.float64)


def calculate_errors(c: Calibration, imgpointss, objpoints, checker_size, sz, files, n_corners, inverse_samples=100,
                     rng: Optional[np.random.Generator] = None, inverse_points=None):
    """src/buildcalibrations.jl:37-67.  The reference draws the round-trip points from the
    global RNG (:54); here they come from `rng` (or are passed in) so the metric is reproducible."""
    N = len(imgpointss)
    P = int(n_corners[0]) * int(n_corners[1])
    img = np.stack([np.asarray(m, dtype=np.float64).reshape(P, 2) for m in imgpointss])
    if inverse_points is None and inverse_samples > 0:
        rng = rng or np.random.default_rng()
        inverse_points = rng.random((N, inverse_samples, 2)) * (np.asarray(sz, dtype=np.float64) - 1) + 1
    e = _shared_solver(c.device).errors(n_corners, np.asarray(objpoints, dtype=np.float64).reshape(P, 3), img, c.f,
                                         c.c, c.k, c.rvecs, c.tvecs, checker_size, inv_pts=inverse_points)
    return dict(n=len(files) if files is not None else N, **e)


def fit_corners(imgpointss: Sequence, n_corners, checker_size, sz, files: Optional[Sequence[str]] = None, aspect=1,
                with_distortion=True, inverse_samples=100, rng=None, device=0):
    """`fit` from already-detected corners: everything after `_detect_corners` in
    src/detect_fit.jl:63-72 and src/buildcalibrations.jl:12-26.  Returns (c, eps)."""
    if len(imgpointss) == 0:
        # src/detect_fit.jl:66
        raise AssertionError("No checkers were detected in any of the images, perhaps try a different `n_corners`.")
    files = list(files) if files is not None else [f"{i + 1}.png" for i in range(len(imgpointss))]
    objpoints = board_template(n_corners)
    m = _solver.fit_model(sz, objpoints, imgpointss, n_corners, with_distortion, aspect, device=device)
    objpoints = objpoints * float(checker_size)        # src/buildcalibrations.jl:15 (Float64, not Int)
    f, cc, rv, tv, cs = obj2img(m.Rs, m.ts, m.frow, m.fcol, m.crow, m.ccol, checker_size)
    c = Calibration(f, cc, rv, tv, cs, m.k, files, device)
    eps = calculate_errors(c, imgpointss, objpoints, checker_size, sz, files, n_corners, inverse_samples, rng=rng)
    return c, eps


def fit(files, n_corners, checker_size, aspect=1, with_distortion=True, inverse_samples=100, with_plot=False,
        detector: Optional[Callable] = None, sz=None, rng=None, device=0):
    """fit(files, n_corners, checker_size; aspect, with_distortion, inverse_samples, with_plot)
    (src/buildcalibrations.jl:12-26).  `detector(file, n_corners) -> (P,2) corners or None`
    overrides the built-in device detector that stands in for `_detect_corners`
    (src/detect_fit.jl:5-21)."""
    builtin = detector is None
    if detector is None:
        from .detect import detect_corners, load_gray      # device detector (csrc/ccal_detect.hip)
        cache = {}

        def detector(f, nc):
            img = cache.setdefault(f, load_gray(f))
            c = detect_corners(img, nc)
            return None if c is None else c.transpose(1, 0, 2).reshape(-1, 2)   # first dim fastest
        if sz is None:
            sz = load_gray(next(iter(files))).shape       # size(FileIO.load(first(files))), src/detect_fit.jl:64
    seen, uniq = set(), []
    for f in files:                                      # unique(files), src/buildcalibrations.jl:14
        if f not in seen:
            seen.add(f)
            uniq.append(f)
    found = None
    if builtin:                                          # the calibration's images in one call: three in flight on the device
        from .detect import detect_corners_many, load_gray
        grays = [load_gray(f) for f in uniq]
        if len({g.shape for g in grays}) == 1:
            cs = detect_corners_many(grays, n_corners)
            found = [(f, None if c is None else c.transpose(1, 0, 2).reshape(-1, 2)) for f, c in zip(uniq, cs)]
    if found is None:
        found = [(f, detector(f, n_corners)) for f in uniq]
    found = [(f, m) for f, m in found if m is not None]  # skipmissing, src/detect_fit.jl:65
    if sz is None:
        raise ValueError("sz (size of the first image, (rows, cols)) is required")
    c, eps = fit_corners([m for _, m in found], n_corners, checker_size, sz, [f for f, _ in found], aspect,
                         with_distortion, inverse_samples, rng, device)
    if with_plot:                                        # src/buildcalibrations.jl:22-24
        from .plot import plot
        plot(c, [m for _, m in found], n_corners, checker_size, sz)
    return c, eps
