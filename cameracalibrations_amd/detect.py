"""Corner detection: mirror of the reference's `_detect_corners` (reference src/detect_fit.jl:5-21),
which calls OpenCV `findChessboardCorners` + `cornerSubPix`.  Here the image goes to the device
detector of csrc/ccal_detect.hip through `ccal_detect_corners`; loading the file stays on the host.
"""
import ctypes as C

import numpy as np

from . import _ffi as F


def detect_corners(img, n_corners, solver=None, stages=False):
    """img: 2-D uint8 array (rows, cols) -- any strides.  -> (n1, n2, 2) RowCol corners with the
    first n_corners dimension fastest (as `reshape(..., n_corners)` at src/detect_fit.jl:19 gives),
    or None when no board is found (the reference's `missing`)."""
    from .calibration import _shared_solver
    s = solver or _shared_solver(0)
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("expected a 2-D uint8 gray image")
    rows, cols = img.shape
    sr, sc = (st // img.itemsize for st in img.strides)
    if min(sr, sc) < 1 or (rows - 1) * sr + (cols - 1) * sc + 1 > rows * cols:
        img = np.ascontiguousarray(img)
        sr, sc = cols, 1
    P = int(n_corners[0]) * int(n_corners[1])
    out = np.zeros((P, 2))
    found = C.c_int32(0)
    stage = np.zeros((2, rows, cols)) if stages else None
    F.check(F.lib().ccal_detect_corners(s._h, C.c_void_p(img.ctypes.data), rows, cols, int(sr), int(sc), int(n_corners[0]),
                                        int(n_corners[1]), out.ctypes.data_as(F._dp), C.byref(found),
                                        stage.ctypes.data_as(F._dp) if stages else None))
    res = out.reshape(int(n_corners[1]), int(n_corners[0]), 2).transpose(1, 0, 2) if found.value else None
    return (res, stage) if stages else res


def detect_corners_many(imgs, n_corners, solver=None):
    """`detect_corners` for the images of one calibration (2-D uint8 arrays of one shape): one call of ccal_detect_corners_many, which
    keeps three images in flight (the host's share of an image overlaps the device's share of the next).  -> list of (n1, n2, 2)
    corner arrays / None, the same values `detect_corners` returns for each image."""
    from .calibration import _shared_solver
    s = solver or _shared_solver(0)
    imgs = [np.ascontiguousarray(im) for im in imgs]
    if not imgs:
        return []
    rows, cols = imgs[0].shape
    for im in imgs:
        if im.dtype != np.uint8 or im.ndim != 2 or im.shape != (rows, cols):
            raise TypeError("expected 2-D uint8 gray images of one shape")
    n, P = len(imgs), int(n_corners[0]) * int(n_corners[1])
    outs = [np.zeros((P, 2)) for _ in imgs]
    found = (C.c_int32 * n)()
    ip = (C.c_void_p * n)(*[im.ctypes.data for im in imgs])
    op = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    F.check(F.lib().ccal_detect_corners_many(s._h, ip, n, rows, cols, cols, 1, int(n_corners[0]), int(n_corners[1]), op, found))
    return [o.reshape(int(n_corners[1]), int(n_corners[0]), 2).transpose(1, 0, 2) if f else None for o, f in zip(outs, found)]


def load_gray(file):
    """FileIO.load + Gray + rawview (src/detect_fit.jl:6-7): 8-bit gray matrix (rows, cols)."""
    from PIL import Image
    with Image.open(file) as im:
        return np.asarray(im.convert("L"), dtype=np.uint8)


def _detect_corners(file, n_corners, sz=None):
    """Same contract as the reference function: (file, corners) or None (`missing`)."""
    img = load_gray(file)
    if sz is not None and tuple(sz) != img.shape:
        raise ValueError(f"{file}: image size {img.shape} differs from sz {tuple(sz)}")
    corners = detect_corners(img, n_corners)
    return None if corners is None else (file, corners)
