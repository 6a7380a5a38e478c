"""Rectified images (reference src/plot_calibration.jl): device warp vs a
numpy restatement of `warp(img, tform, axs)` with linear interpolation, and the geometric property
the reference's docstring states (detected and rectified crosses coincide on a regular grid)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PNG = os.path.join(ROOT, "tests", "golden", "example_png")


def _warp_numpy(img, fwd, ratio, a0, b0, shape, off):
    """warp(img, tform, axs): out[a, b] = img(tform((a, b))) bilinear, 0 outside (host oracle)."""
    A, B = np.meshgrid(np.arange(shape[0]) + a0, np.arange(shape[1]) + b0, indexing="ij")
    xyz = np.stack([A / ratio, B / ratio, np.zeros_like(A, dtype=float)], -1).reshape(-1, 3)
    rc = fwd(xyz) - off
    y, x = rc[:, 0], rc[:, 1]
    h, w = img.shape[:2]
    inside = (y >= 0) & (y <= h - 1) & (x >= 0) & (x <= w - 1)
    y0 = np.where(inside, np.floor(y), 0).astype(int)
    x0 = np.where(inside, np.floor(x), 0).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (y - y0)[:, None], (x - x0)[:, None]
    im = img.reshape(h, w, -1).astype(np.float64)
    v = (1 - fy) * ((1 - fx) * im[y0, x0] + fx * im[y0, x1]) + fy * ((1 - fx) * im[y1, x0] + fx * im[y1, x1])
    v = np.where(inside[:, None], v, 0.0)
    return np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8).reshape(shape[0], shape[1], -1)


def test_axes_and_ratio_host():
    from cameracalibrations_amd.plot import get_axes, get_ratio
    p = np.stack(np.meshgrid(np.arange(5) * 40.0 + 100, np.arange(8) * 40.0 + 50, indexing="ij"), -1)
    assert get_ratio(p, 2.0) == 20.0
    assert get_axes(20.0, 2.0, (5, 8), (375, 500)) == ((-108, 266), (-110, 389))


@pytest.mark.gpu
def test_gpu_rectified_image_matches_numpy_warp(tmp_path):
    from PIL import Image

    import cameracalibrations_amd as cc
    from cameracalibrations_amd.plot import plot, rectified_image
    ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
    files = [os.path.join(PNG, f) for f in ex["files"]]
    nc = tuple(ex["n_corners"])
    corners = [cc.detect_corners(np.asarray(Image.open(f)), nc) for f in files]
    c, _ = cc.fit_corners([p.transpose(1, 0, 2).reshape(-1, 2) for p in corners], nc, 2.5, ex["image_size_wh"], files=files,
                          rng=np.random.default_rng(0))
    for gray in (True, False):
        img = np.asarray(Image.open(files[1]).convert("L" if gray else "RGB"))
        out, ((a0, a1), (b0, b1)), itform = rectified_image(c, 2, img, corners[1], nc, 2.5)
        assert out.shape[:2] == img.shape[:2] == (a1 - a0 + 1, b1 - b0 + 1)
        from cameracalibrations_amd.plot import get_ratio
        ratio = get_ratio(corners[1], 2.5)
        want = _warp_numpy(img, lambda xyz: c(xyz, 2), ratio, a0, b0, out.shape[:2], 1.0)
        diff = np.abs(out.reshape(want.shape).astype(int) - want.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3          # rounding ties only
        assert out.std() > 20                                          # a real picture, not a blank
    # rectified corners sit on a regular grid with spacing ratio * checker_size
    g = itform(corners[1].reshape(-1, 2)).reshape(nc[0], nc[1], 2)
    step = ratio * 2.5
    assert np.abs(np.linalg.norm(np.diff(g, axis=0), axis=-1) - step).max() < 0.05 * step
    assert np.abs(np.linalg.norm(np.diff(g, axis=1), axis=-1) - step).max() < 0.05 * step
    written = plot(c, corners, nc, 2.5, ex["image_size_wh"], directory=str(tmp_path / "debug"))
    assert len(written) == 6 and all(os.path.getsize(w) > 1000 for w in written)
