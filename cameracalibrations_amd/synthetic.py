"""Seeded synthetic checkerboard-corner sets.

Pure numpy; used by bench.py and the tests to produce the named workloads.  The layout
follows the reference: board template with the first `n_corners` dimension fastest and
z = 0 (reference src/detect_fit.jl:69), observations cast to Float32
(src/detect_fit.jl:31), image size given as OpenCV sees it, (width, height) = (rows, cols)
of the Julia image (src/detect_fit.jl:32).
"""
from dataclasses import dataclass, field

import numpy as np

# OpenCV calib3d flag bits
USE_INTRINSIC_GUESS = 0x1
FIX_ASPECT_RATIO = 0x2
FIX_PRINCIPAL_POINT = 0x4
ZERO_TANGENT_DIST = 0x8
FIX_FOCAL_LENGTH = 0x10
FIX_K1 = 0x20
FIX_K2 = 0x40
FIX_K3 = 0x80
FIX_TANGENT_DIST = 0x200000

IMAGE_SIZE = (1500, 2000)  # (width, height)


def board_template(n_corners, dtype=np.float64):
    """(P, 3) template, first dim fastest, unit squares, z = 0 (src/detect_fit.jl:69)."""
    n1, n2 = n_corners
    p = np.arange(n1 * n2)
    return np.stack([p % n1, p // n1, np.zeros_like(p)], axis=1).astype(dtype)


def rodrigues(rvecs):
    """(N,3) rotation vectors -> (N,3,3) matrices."""
    rvecs = np.asarray(rvecs, dtype=np.float64).reshape(-1, 3)
    theta = np.linalg.norm(rvecs, axis=1)
    safe = np.where(theta > 0, theta, 1.0)
    n = rvecs / safe[:, None]
    K = np.zeros((len(rvecs), 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -n[:, 2], n[:, 1]
    K[:, 1, 0], K[:, 1, 2] = n[:, 2], -n[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -n[:, 1], n[:, 0]
    c, s = np.cos(theta)[:, None, None], np.sin(theta)[:, None, None]
    eye = np.eye(3)[None]
    return c * eye + (1 - c) * n[:, :, None] * n[:, None, :] + s * K


def project(obj, rvecs, tvecs, intr):
    """Pinhole + (k1,k2,p1,p2,k3) projection.  obj (P,3); rvecs,tvecs (N,3);
    intr = fx fy cx cy k1 k2 p1 p2 k3.  -> (N,P,2) float64."""
    fx, fy, cx, cy, k1, k2, p1, p2, k3 = [float(v) for v in intr]
    R = rodrigues(rvecs)
    X = np.einsum("nij,pj->npi", R, np.asarray(obj, dtype=np.float64)) + np.asarray(tvecs)[:, None, :]
    x, y = X[..., 0] / X[..., 2], X[..., 1] / X[..., 2]
    r2 = x * x + y * y
    cd = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
    xd = x * cd + p1 * a1 + p2 * a2
    yd = y * cd + p1 * a3 + p2 * a1
    return np.stack([fx * xd + cx, fy * yd + cy], axis=-1)


@dataclass
class Workload:
    name: str
    n_corners: tuple
    n_images: int
    flags: int
    aspect: float
    intr_true: np.ndarray
    image_size: tuple = IMAGE_SIZE
    obj: np.ndarray = field(default=None, repr=False)       # (P,3) float32
    img: np.ndarray = field(default=None, repr=False)       # (N,P,2) float32
    rvecs_true: np.ndarray = field(default=None, repr=False)
    tvecs_true: np.ndarray = field(default=None, repr=False)
    sigma: float = 0.1

    @property
    def n_points(self):
        return self.n_corners[0] * self.n_corners[1]


def _true_intrinsics(model, aspect=1.0):
    w, h = IMAGE_SIZE
    fy = 1600.0
    fx = aspect * fy if model == "ref" else 1610.0
    cx, cy = (w - 1) / 2 + 7.3, (h - 1) / 2 - 4.1
    dist = {
        "ref": (-0.12, 0, 0, 0, 0),
        "ref_nodist": (0, 0, 0, 0, 0),
        "pinhole": (0, 0, 0, 0, 0),
        "k1k2p1p2": (-0.12, 0.03, 1e-3, -5e-4, 0),
        "full": (-0.12, 0.03, 1e-3, -5e-4, -4e-3),
    }[model]
    if model == "ref_nodist":
        fx = aspect * fy
    return np.array([fx, fy, cx, cy, *dist], dtype=np.float64)


MODEL_FLAGS = {
    # reference flag words, src/detect_fit.jl:40
    "ref": ZERO_TANGENT_DIST | FIX_K3 | FIX_K2 | FIX_ASPECT_RATIO,
    "ref_nodist": ZERO_TANGENT_DIST | FIX_K3 | FIX_K2 | FIX_K1 | FIX_ASPECT_RATIO,
    # BASELINE.json configs 1..4 (SURVEY.md 8d "LM settings for timing")
    "pinhole": ZERO_TANGENT_DIST | FIX_K1 | FIX_K2 | FIX_K3,
    "k1k2p1p2": FIX_K3,
    "full": 0,
}


def make_workload(name, n_corners, n_images, model, seed, sigma=0.1, aspect=1.0, depth=(8.0, 16.0),
                  margin=2.0):
    """Rejection-sample `n_images` poses whose corners all land inside the image."""
    rng = np.random.default_rng(seed)
    intr = _true_intrinsics(model, aspect)
    obj = board_template(n_corners)
    centre = obj.mean(axis=0)
    w, h = IMAGE_SIZE
    rv_keep, tv_keep = [], []
    need = n_images
    while need > 0:
        batch = max(4 * need, 1024)
        rv = rng.uniform(-0.5, 0.5, size=(batch, 3))
        z = rng.uniform(depth[0], depth[1], size=batch)
        u0 = rng.uniform(0, w - 1, size=batch)
        v0 = rng.uniform(0, h - 1, size=batch)
        R = rodrigues(rv)
        ray = np.stack([(u0 - intr[2]) / intr[0], (v0 - intr[3]) / intr[1], np.ones(batch)], axis=1)
        tv = ray * z[:, None] - np.einsum("nij,j->ni", R, centre)
        uv = project(obj, rv, tv, intr)
        Zc = np.einsum("nij,pj->npi", R, obj)[..., 2] + tv[:, None, 2]
        ok = ((uv[..., 0] >= margin) & (uv[..., 0] <= w - 1 - margin) & (uv[..., 1] >= margin)
              & (uv[..., 1] <= h - 1 - margin) & (Zc > 1.0)).all(axis=1)
        idx = np.nonzero(ok)[0][:need]
        rv_keep.append(rv[idx])
        tv_keep.append(tv[idx])
        need -= len(idx)
    rvecs = np.concatenate(rv_keep)
    tvecs = np.concatenate(tv_keep)
    uv = project(obj, rvecs, tvecs, intr)
    if sigma > 0:
        uv = uv + rng.normal(0.0, sigma, size=uv.shape)
    return Workload(name=name, n_corners=tuple(n_corners), n_images=n_images, flags=MODEL_FLAGS[model],
                    aspect=aspect, intr_true=intr, obj=obj.astype(np.float32), img=uv.astype(np.float32),
                    rvecs_true=rvecs, tvecs_true=tvecs, sigma=sigma)


_BASE_SEED = 20241220

CONFIGS = {
    # name: (n_corners, N, model, config index)
    "C1": ((9, 6), 100, "pinhole", 1),
    "C2": ((9, 6), 1000, "k1k2p1p2", 2),
    "C3": ((9, 6), 1000, "k1k2p1p2", 2),   # same set as C2; C3/C4 shard it across GPUs
    "C5": ((15, 10), 10000, "full", 4),
    "REF": ((9, 6), 100, "ref", 5),        # reference flag word 0xCA on synthetic data
}


def named_workload(name, n_images=None, sigma=0.1, seed_offset=0):
    """BASELINE.json workloads by name; `n_images` overrides N (per-rank shards, small tests)."""
    n_corners, N, model, idx = CONFIGS[name]
    depth = (8.0, 16.0) if n_corners == (9, 6) else (14.0, 28.0)
    return make_workload(name, n_corners, n_images or N, model, _BASE_SEED + idx + seed_offset, sigma=sigma,
                         depth=depth)


def generate_checkerboard(n_corners, n):
    """An axis-aligned synthetic board of the shape the reference's detector tests render (test/runtests.jl:6-18: alternating
    squares of n x n pixels with a one-square white border): -> expected RowCol corners (P, 2), image (rows, cols) in 0..255.
    Input generator only (bench.py's `detect` block, tools/, tests); the detector itself never sees it."""
    n1, n2 = n_corners
    i = np.arange(1, n1 + 2)[:, None]
    j = np.arange(1, n2 + 2)[None, :]
    img = np.where(i % 2 == 1, j % 2 == 1, j % 2 == 0)
    pad = np.ones((n1 + 3, n2 + 3), dtype=bool)
    pad[1:-1, 1:-1] = img
    imgl = np.kron(pad.astype(np.float64), np.ones((n, n)))
    xys = np.array([(n * ii - 0.5, n * jj - 0.5) for jj in range(n2 + 1, 1, -1) for ii in range(n1 + 1, 1, -1)])
    return xys, imgl * 255.0


def render_board_view(n_corners, rvec, tvec, size=(480, 640), f=600.0, blur_sigma=0.0, noise=0.0, rng=None, supersample=3):
    """A perspective view of an n1 x n2-corner checkerboard (unit squares, one-square margin, dark squares 30 / bright 225) seen by a
    pinhole camera of focal length f (pixel (row, col) = (x_cv, y_cv) of the reference convention) under pose (rvec, tvec): -> uint8 image
    (rows, cols) and the projected true corners (P, 2), first corner dimension fastest.  Input generator only (tools/fuzz_detect.py, tests:
    perspective, blur and noise beyond the reference's axis-aligned synthetic boards); the detector never sees it."""
    n1, n2 = n_corners
    H, W = size
    R = rodrigues(np.asarray(rvec, dtype=np.float64)[None])[0]
    tv = np.asarray(tvec, dtype=np.float64)
    ss = int(supersample)
    rr, cc = np.meshgrid((np.arange(H * ss) + 0.5) / ss - 0.5, (np.arange(W * ss) + 0.5) / ss - 0.5, indexing="ij")
    d = np.stack([(rr - (H - 1) / 2) / f, (cc - (W - 1) / 2) / f, np.ones_like(rr)], axis=-1)
    n = R[:, 2]
    t = (tv @ n) / (d @ n)
    b = (d * t[..., None] - tv) @ R                # board coordinates (squares), z ~ 0
    u, v = b[..., 0], b[..., 1]
    inside = (u > -1) & (u < n1) & (v > -1) & (v < n2) & (t > 0)
    dark = ((np.floor(u).astype(int) + np.floor(v).astype(int)) % 2) == 0
    img = np.where(inside, np.where(dark, 30.0, 225.0), 225.0)
    img = img.reshape(H, ss, W, ss).mean(axis=(1, 3))
    if blur_sigma > 0:
        kr = max(4, int(np.ceil(3 * blur_sigma)))
        k = np.exp(-0.5 * (np.arange(-kr, kr + 1) / blur_sigma) ** 2)
        k /= k.sum()
        img = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 0, img)
        img = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), 1, img)
    if rng is not None or noise > 0:               # (a caller's generator is always advanced: the draws of tools/fuzz_detect.py stay what they were)
        img = img + (rng or np.random.default_rng(0)).normal(0, noise, img.shape)
    pc = board_template((n1, n2)) @ R.T + tv
    truth = np.stack([f * pc[:, 0] / pc[:, 2] + (H - 1) / 2, f * pc[:, 1] / pc[:, 2] + (W - 1) / 2], axis=1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), truth
