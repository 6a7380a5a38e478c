"""Corner detection (next row after the fit).

The reference's bar for this step is its own test suite: every detected corner within 1 px of the
analytic position on synthetic boards (reference test/runtests.jl:6-43) and within 1 px of
test/example/corners.json on the six example PNGs (test/runtests.jl:53-65), in OpenCV's corner
order.  CPU tests hold the oracle (oracle/detect_oracle.py) to that bar; GPU tests hold the HIP
detector to the same bar, to the oracle (1e-6 px), and run the reference's end-to-end
"Full calibration" test (test/runtests.jl:68-85) from the PNG files.
"""
import json
import os

import numpy as np
import pytest

from cameracalibrations_amd import synthetic as syn
from oracle import detect_oracle as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PNG = os.path.join(ROOT, "tests", "golden", "example_png")


def _example():
    from PIL import Image
    ex = json.load(open(os.path.join(ROOT, "tests", "golden", "example_corners.json")))
    imgs = [np.asarray(Image.open(os.path.join(PNG, f))) for f in ex["files"]]
    return ex, imgs


def test_oracle_real_images_within_one_pixel():
    ex, imgs = _example()
    for f, im, tgt in zip(ex["files"], imgs, ex["corners"]):
        res = D.detect_corners(im, tuple(ex["n_corners"]))
        assert res is not None, f
        assert np.linalg.norm(res - np.array(tgt, dtype=float), axis=1).max() < 1, f     # runtests.jl:63


@pytest.mark.parametrize("w,h,ratio", [(13, 14, 95), (14, 15, 100)])
def test_oracle_synthetic_boards(w, h, ratio):
    xys, img = syn.generate_checkerboard((w, h), ratio)
    res = D.detect_corners(img, (w, h))
    assert res is not None and np.linalg.norm(res - xys, axis=1).max() < 1                # runtests.jl:26,40


def test_oracle_rejects_wrong_board():
    ex, imgs = _example()
    assert D.detect_corners(imgs[0], (7, 9)) is None            # `missing` (src/detect_fit.jl:16)
    assert D.detect_corners(np.full((200, 200), 128, dtype=np.uint8), (5, 8)) is None


# a strongly sheared view (draw 40 of tools/fuzz_detect.py --seed 6 --max-tilt 0.9: profiles/r06_detector_probe.txt): with OpenCV's 23 x 23
# window five of its 28 corners are pulled more than 2 px off their response peak by a neighbouring corner's edges and are reverted, the worst
# kept corner ends 1.3 px off; with the window cut short of the nearest neighbour every corner lands within 0.1 px
_SHEARED = dict(n_corners=(7, 4), rvec=[-0.8590072086231154, -0.8978087384501281, -0.6208911503178958],
                tvec=[-3.0475641056510874, -0.012294546882449131, 10.423637389537959], blur_sigma=0.8)


def _board_error(res, truth, n_corners):
    n1, n2 = n_corners
    T = truth.reshape(n2, n1, 2)
    return min(np.linalg.norm(res - c.reshape(-1, 2), axis=1).max() for c in (T, T[::-1, ::-1], T[::-1], T[:, ::-1]))


def test_oracle_window_follows_the_nearest_neighbour():
    """cornerSubPix's window per response peak: OpenCV's (11, 11) unless another peak is closer than 27 px, then short of the half-way point
    to it (floor(0.45 d) - 1, at least 3).  The sheared view goes from 1.3 px worst corner to under 0.2 px."""
    assert D.subpix_windows(np.array([[10, 10], [10, 40], [40, 10]])).tolist() == [11, 11, 11]
    assert D.subpix_windows(np.array([[10, 10], [10, 26], [40, 10]])).tolist() == [6, 6, 11]
    assert D.subpix_windows(np.array([[10, 10], [10, 17]])).tolist() == [3, 3]
    assert D.subpix_windows(np.array([[10, 10]])).tolist() == [11]
    img, truth = syn.render_board_view(**_SHEARED)
    res = D.detect_corners(img, _SHEARED["n_corners"])
    assert res is not None and _board_error(res, truth, _SHEARED["n_corners"]) < 0.2


def test_oracle_reverted_refinements_take_the_response_peaks_parabola(monkeypatch):
    """A refinement that leaves its response peak by more than 2 px falls back to the peak's own sub-pixel position: per axis the vertex of
    the parabola through the peak's response and its two neighbours'.  Exact on a response that IS a parabola; within half a pixel of the
    integer peak always; and the sheared board is found with it under the fixed window too, every corner within 1.5 px of the truth."""
    # (a) an X corner 6.5 px beside the point, inside cornerSubPix's window: the iteration runs to it -> more than 2 px -> reverted
    rr, cc = np.meshgrid(np.arange(64.0), np.arange(64.0), indexing="ij")
    im = np.where((rr < 30.5) ^ (cc < 38.5), 200.0, 40.0)
    resp = -((rr - 30.3) ** 2 + 2.0 * (cc - 31.8) ** 2)
    out = D.corner_subpix(im, np.array([[30, 32]]), resp)
    assert np.allclose(out[0], (30.3, 31.8), atol=1e-12)
    assert np.array_equal(D.corner_subpix(im, np.array([[30, 32]]), None)[0], (30.0, 32.0))      # without a response: the integer peak
    # (b) the sheared view under OpenCV's window for every peak
    monkeypatch.setenv("CCAL_DET_FIXED_WIN", "1")
    img, truth = syn.render_board_view(**_SHEARED)
    imf = img.astype(np.float64)
    R = D.chess_response(D.blur(imf))
    pts, _ = D.peaks(R)
    a, b = D.corner_subpix(imf, pts, None), D.corner_subpix(imf, pts, R)
    moved = np.any(a != b, axis=1)
    assert moved.sum() >= 3 and np.abs(a - b).max() <= 0.5 and np.array_equal(a[moved], np.rint(a[moved]))
    res = D.detect_corners(img, _SHEARED["n_corners"])
    assert res is not None and 1.0 < _board_error(res, truth, _SHEARED["n_corners"]) < 1.5


@pytest.mark.gpu
def test_gpu_sheared_view_matches_the_oracle():
    """The kernel's per-peak windows (7 ... 11 on this view) against the oracle's, and -- in a process of its own with OpenCV's window forced
    for every peak -- the kernel's fallback branch (parabola through the response peak) against the oracle's."""
    import subprocess
    import sys
    from cameracalibrations_amd.detect import detect_corners
    img, truth = syn.render_board_view(**_SHEARED)
    n1, n2 = _SHEARED["n_corners"]
    want = D.detect_corners(img, (n1, n2))
    got = detect_corners(img, (n1, n2))
    assert want is not None and got is not None
    assert np.abs(got.transpose(1, 0, 2).reshape(-1, 2) - want).max() < 1e-6
    assert _board_error(want, truth, (n1, n2)) < 0.2
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from cameracalibrations_amd import synthetic as syn; "
            "from cameracalibrations_amd.detect import detect_corners; from oracle import detect_oracle as D; "
            "S = %r; img, _ = syn.render_board_view(**S); w = D.detect_corners(img, S['n_corners']); g = detect_corners(img, S['n_corners']); "
            "print('DIFF', float(np.abs(g.transpose(1, 0, 2).reshape(-1, 2) - w).max()))" % (ROOT, _SHEARED))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, CCAL_DET_FIXED_WIN="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    assert float(p.stdout.strip().splitlines()[-1].split()[1]) < 1e-6

# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_stages_match_oracle():
    from cameracalibrations_amd.detect import detect_corners
    ex, imgs = _example()
    im = imgs[2]
    res, stage = detect_corners(im, tuple(ex["n_corners"]), stages=True)
    g = D.blur(im.astype(np.float64))
    R = D.chess_response(g)
    assert np.abs(stage[0] - g).max() < 1e-10
    assert np.abs(stage[1] - R).max() < 1e-9
    want = D.detect_corners(im, tuple(ex["n_corners"]))
    assert np.abs(res.transpose(1, 0, 2).reshape(-1, 2) - want).max() < 1e-6


@pytest.mark.gpu
def test_gpu_real_images_reference_bar():
    from cameracalibrations_amd.detect import detect_corners
    ex, imgs = _example()
    nc = tuple(ex["n_corners"])
    for f, im, tgt in zip(ex["files"], imgs, ex["corners"]):
        res = detect_corners(im, nc)
        assert res is not None and res.shape == (nc[0], nc[1], 2), f
        flat = res.transpose(1, 0, 2).reshape(-1, 2)                                       # vec(corners)
        assert np.linalg.norm(flat - np.array(tgt, dtype=float), axis=1).max() < 1, f     # runtests.jl:63
        assert np.abs(flat - D.detect_corners(im, nc)).max() < 1e-6, f
        # Julia hands OpenCV the column-major matrix: same image, transposed strides
        jl = np.asfortranarray(im)
        assert np.array_equal(detect_corners(jl, nc), res)
    assert detect_corners(imgs[0], (7, 9)) is None


@pytest.mark.gpu
def test_gpu_synthetic_boards_reference_loop():
    """for w in 13:15, h in 13:15, ratio in 95:100 if isodd(w) != isodd(h)   (runtests.jl:36-43)"""
    from cameracalibrations_amd.detect import detect_corners
    n = 0
    for w in (13, 14, 15):
        for h in (13, 14, 15):
            if (w % 2) == (h % 2):
                continue
            for ratio in range(95, 101):
                xys, img = syn.generate_checkerboard((w, h), ratio)
                res = detect_corners(img.astype(np.uint8), (w, h))
                assert res is not None, (w, h, ratio)
                flat = res.transpose(1, 0, 2).reshape(-1, 2)
                assert np.linalg.norm(flat - xys, axis=1).max() < 1, (w, h, ratio)
                n += 1
    assert n == 24


@pytest.mark.gpu
def test_gpu_full_calibration_from_files():
    """The reference's "Full calibration" testset (test/runtests.jl:68-85), files in, everything on
    the device: detection, fit, maps, error metrics."""
    import cameracalibrations_amd as cc
    files = sorted(os.path.join(PNG, f) for f in os.listdir(PNG) if f.endswith(".png"))
    c, eps = cc.fit(files, (5, 8), 1, rng=np.random.default_rng(3))
    assert eps["n"] == 6
    assert all(eps[k] < 1 for k in ("reprojection", "projection", "distance", "inverse")), eps
    f = cc.rectification(c, 1)
    i = cc.RowCol(1, 2)
    assert np.array_equal(f(i), c(i, 1)[[0, 1]])
    # sub-pixel corners give a far better fit than the integer-rounded fixture (0.469 px)
    assert eps["reprojection"] < 0.35   # sub-pixel corners: 0.27 px, against 0.469 px from the integer-rounded fixture


@pytest.mark.gpu
def test_gpu_fit_drops_images_without_a_board_and_duplicates(tmp_path):
    """src/buildcalibrations.jl:14 (`unique(files)`) and src/detect_fit.jl:65 (`skipmissing`): a file listed twice counts
    once, an image in which no board is found is dropped -- the fit is the one of the remaining files."""
    from PIL import Image

    import cameracalibrations_amd as cc
    files = sorted(os.path.join(PNG, f) for f in os.listdir(PNG) if f.endswith(".png"))
    blank = str(tmp_path / "blank.png")
    Image.fromarray(np.full((375, 500), 128, dtype=np.uint8)).save(blank)
    noise = str(tmp_path / "noise.png")
    Image.fromarray(np.random.default_rng(1).integers(0, 256, (375, 500), dtype=np.uint8)).save(noise)
    c0, e0 = cc.fit(files, (5, 8), 1, rng=np.random.default_rng(3))
    c1, e1 = cc.fit([files[0], blank] + files + [noise, files[2]], (5, 8), 1, rng=np.random.default_rng(3))
    assert e1["n"] == 6 and list(c1.files) == list(c0.files)
    assert np.array_equal(c1.f, c0.f) and np.array_equal(c1.c, c0.c) and c1.k == c0.k
    assert np.array_equal(c1.rvecs, c0.rvecs) and np.array_equal(c1.tvecs, c0.tvecs)
    assert e1["reprojection"] == e0["reprojection"]


@pytest.mark.gpu
def test_gpu_detection_threaded():
    """runtests.jl:45-51: the detection wrapper is used under Threads.@threads -- one device context
    per host thread, identical results."""
    from concurrent.futures import ThreadPoolExecutor

    from cameracalibrations_amd.detect import detect_corners
    ex, imgs = _example()
    nc = tuple(ex["n_corners"])
    serial = [detect_corners(im, nc) for im in imgs]
    with ThreadPoolExecutor(max_workers=4) as pool:
        threaded = list(pool.map(lambda im: detect_corners(im, nc), imgs * 4))
    for k, res in enumerate(threaded):
        assert np.array_equal(res, serial[k % len(imgs)])


@pytest.mark.gpu
def test_gpu_detection_of_many_images_in_flight():
    """ccal_detect_corners_many keeps three images of a calibration in flight (detector, pinned block and stream per slot): the same
    corners as image-by-image calls, for the reference's six example images (several times over, a blank and a noise image among them:
    `missing` stays `missing`), for one image, for none; images of different shapes are refused."""
    from cameracalibrations_amd.detect import detect_corners, detect_corners_many
    ex, imgs = _example()
    nc = tuple(ex["n_corners"])
    blank = np.full_like(imgs[0], 128)
    noise = np.random.default_rng(1).integers(0, 256, imgs[0].shape, dtype=np.uint8)
    batch = imgs + [blank] + imgs[::-1] + [noise] + imgs[:1] + [blank, blank] + imgs[2:4]
    one = [detect_corners(im, nc) for im in batch]
    for rep in range(3):
        many = detect_corners_many(batch, nc)
        assert len(many) == len(batch)
        for a, b in zip(one, many):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a, b)
    assert sum(m is None for m in many) == 4
    assert detect_corners_many([], nc) == []
    assert np.array_equal(detect_corners_many(imgs[:1], nc)[0], one[0])
    assert np.array_equal(detect_corners_many(imgs[:2], nc)[1], one[1])
    with pytest.raises(TypeError):
        detect_corners_many([imgs[0], imgs[1][:-8]], nc)
    # other board sizes on the same context afterwards (buffers are per slot)
    xys, img = syn.generate_checkerboard((13, 14), 97)
    res = detect_corners_many([img.astype(np.uint8)] * 4, (13, 14))
    for r in res:
        assert r is not None and np.linalg.norm(r.transpose(1, 0, 2).reshape(-1, 2) - xys, axis=1).max() < 1
