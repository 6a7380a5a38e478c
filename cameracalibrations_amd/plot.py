"""Rectified debug images: mirror of the reference's `plot` (reference src/plot_calibration.jl).

The warp runs on the device (`ccal_rectify_image`, csrc/ccal_maps.hip); drawing the crosses and
writing PNGs stays on the host.  Conventions follow the reference, including its use of the fitted
0-based RowCol as 1-based Julia indices when sampling (index_offset = 1).
"""
import ctypes as C
import os

import numpy as np

from . import _ffi as F
from .calibration import _shared_solver


def get_ratio(imgpoints, checker_size):
    """src/plot_calibration.jl:8-13: mean corner spacing in pixels per world unit."""
    p = np.asarray(imgpoints, dtype=np.float64)              # (n1, n2, 2)
    l1 = np.linalg.norm(np.diff(p, axis=0), axis=-1).mean()
    l2 = np.linalg.norm(np.diff(p, axis=1), axis=-1).mean()
    return 0.5 * (l1 + l2) / checker_size


def _round_half_even(x):
    return np.rint(np.asarray(x, dtype=np.float64)).astype(int)   # Julia round(Int, x): ties to even


def get_axes(ratio, checker_size, n_corners, sz):
    """src/plot_calibration.jl:1-6 -> ((a0, a1), (b0, b1)) inclusive index ranges."""
    w = _round_half_even(ratio * checker_size * (np.asarray(n_corners) - 1))
    mn = _round_half_even((w - np.asarray(sz)) / 2)
    mx = mn + np.asarray(sz) - 1
    return (int(mn[0]), int(mx[0])), (int(mn[1]), int(mx[1]))


def rectified_image(c, extrinsic_index, img, imgpoints, n_corners, checker_size, sz=None, index_offset=1.0):
    """image_transformations + warp (src/plot_calibration.jl:15-22,40).  img: (rows, cols) or
    (rows, cols, 3) uint8; imgpoints: (n1, n2, 2) corners of that image.  -> (warped uint8 array,
    axes, itform) where itform maps RowCol -> warped-image coordinates."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise TypeError("expected a uint8 gray or RGB image")
    sz = tuple(sz) if sz is not None else img.shape[:2]
    i = c._index(extrinsic_index)
    ratio = get_ratio(imgpoints, checker_size)
    (a0, a1), (b0, b1) = get_axes(ratio, checker_size, n_corners, sz)
    ch = 1 if img.ndim == 2 else img.shape[2]
    src = np.ascontiguousarray(img)
    out = np.zeros((a1 - a0 + 1, b1 - b0 + 1, ch), dtype=np.uint8)
    s = _shared_solver(c.device)
    F.check(F.lib().ccal_rectify_image(s._h, C.c_void_p(src.ctypes.data), src.shape[0], src.shape[1], ch,
                                       src.shape[1] * ch, ch, 1, F.f64(c.f)[1], F.f64(c.c)[1], float(c.k),
                                       F.f64(c.rvecs[i])[1], F.f64(c.tvecs[i])[1], float(c.checker_size), float(ratio),
                                       a0, b0, out.shape[0], out.shape[1], float(index_offset),
                                       C.c_void_p(out.ctypes.data)))

    def itform(rc):                                           # s o pop o image2real[i]
        return ratio * c(rc, extrinsic_index)[..., :2]
    return (out[..., 0] if img.ndim == 2 else out), ((a0, a1), (b0, b1)), itform


def draw_crosses(img, points, n1, color, origin=(0, 0)):
    """src/plot_calibration.jl:24-28: crosses of radius |p_1 - p_n1| / n1 / 5 at the rounded points.
    points: (n1, n2, 2), vec()-ed with the first index fastest; origin: index of img[0, 0]."""
    p = np.asarray(points, dtype=np.float64)
    pts = _round_half_even(p.transpose(1, 0, 2).reshape(-1, 2))
    radius = int(_round_half_even(np.linalg.norm(pts[0] - pts[n1 - 1]) / n1 / 5))
    h, w = img.shape[:2]
    for r, cc in pts - np.asarray(origin):
        for d in range(-radius, radius + 1):
            for rr, c2 in ((r + d, cc), (r, cc + d)):
                if 0 <= rr < h and 0 <= c2 < w:
                    img[rr, c2] = color
    return img


def plot(c, imgpointss, n_corners, checker_size, sz, directory="debug"):
    """plot(c, imgpointss, n_corners, checker_size, sz) (src/plot_calibration.jl:34-44): one rectified
    PNG per calibration image; red crosses = detected corners (drawn before the warp), blue crosses =
    their rectified locations."""
    from PIL import Image
    os.makedirs(directory, exist_ok=True)
    written = []
    for idx, (file, imgpoints) in enumerate(zip(c.files, imgpointss)):
        with Image.open(file) as im:
            img = np.array(im.convert("RGB"), dtype=np.uint8)
        p = np.asarray(imgpoints, dtype=np.float64).reshape(n_corners[1], n_corners[0], 2).transpose(1, 0, 2) \
            if np.asarray(imgpoints).ndim == 2 else np.asarray(imgpoints, dtype=np.float64)
        draw_crosses(img, p, n_corners[0], (255, 0, 0))
        imgw, ((a0, _), (b0, _)), itform = rectified_image(c, idx + 1, img, p, n_corners, checker_size, sz)
        draw_crosses(imgw, itform(p.reshape(-1, 2)).reshape(p.shape), n_corners[0], (0, 0, 255), origin=(a0, b0))
        out = os.path.join(directory, os.path.basename(file))
        Image.fromarray(imgw).save(out)
        written.append(out)
    return written
