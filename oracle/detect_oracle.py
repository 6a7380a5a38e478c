"""CPU oracle of the checkerboard-corner detector (TEST INFRASTRUCTURE ONLY).

The reference detects corners with OpenCV `findChessboardCorners` + `cornerSubPix`
(reference src/detect_fit.jl:5-21).  `findChessboardCorners` is a large heuristic pipeline
that cannot be restated from the reference tree (OpenCV is an un-vendored dependency and is not
installed here), so the detector of this repository is its OWN algorithm; what is pinned is the
reference's own bar for this step: every corner within 1 px of the analytic position on the
synthetic boards of test/runtests.jl:6-43 and within 1 px of test/example/corners.json on the six
example PNGs (test/runtests.jl:53-65), in OpenCV's corner order.  Only the sub-pixel stage is a
restatement of a published algorithm ([OCV-K] `cornerSubPix`, window (11,11), criteria (30, 1e-3)).

Algorithm (the HIP kernels in cameracalibrations_amd/csrc/ccal_detect.hip follow it step by step):
  1. 9-tap Gaussian blur, sigma = 1, replicated borders.
  2. ChESS response (Bennett & Lasenby) on the radius-5 16-sample ring:
        R = sum_{n<4} |I_n + I_{n+8} - I_{n+4} - I_{n+12}|  -  sum_{n<8} |I_n - I_{n+8}|  -  16 |mean_ring - mean_5|
     zeroed within 7 px of the border.
  3. Peaks: maxima of R over a 13 x 13 window (ties broken by raster order), R > 0.15 max(R).
  4. cornerSubPix on the ORIGINAL image around every peak; peaks that converge within 2 px of a
     stronger one are merged.
  5. Lattice: breadth-first growth from a central seed along two local basis vectors, best fully
     occupied n1 x n2 window by summed response, right-handed, first index fastest, and of the two
     180-degree orientations the one whose OUTER square at the first corner is dark (this is how
     OpenCV orders both the reference's synthetic boards and its example photographs).

Coordinates: the image is the Julia matrix img[row, col]; a corner is RowCol = (row, col) in 0-based
pixel-centre coordinates, which is what OpenCV returns as (x, y) for the transposed view it is
handed (reference src/detect_fit.jl:7).
"""
import os

import numpy as np

SIGMA = 1.0
BLUR_R = 4
RING = [(0, 5), (2, 5), (3, 3), (5, 2), (5, 0), (5, -2), (3, -3), (2, -5), (0, -5), (-2, -5), (-3, -3), (-5, -2),
        (-5, 0), (-5, 2), (-3, 3), (-2, 5)]          # (d_row, d_col)
BORDER = 7
NMS_R = 6
REL_THRESHOLD = 0.15
SUBPIX_WIN = 11
SUBPIX_MIN_WIN = 3
SUBPIX_WIN_FRAC = float(os.environ.get("CCAL_DET_WIN_FRAC", "0.45"))     # (the environment variable: experiments only)
SUBPIX_ITERS = 30
SUBPIX_EPS = 1e-3
SUBPIX_MAX_SHIFT = 2.0    # a refinement that leaves its response peak by more than this is not believed: the peak stands (OpenCV: the half window, 11)
MERGE_DIST = 2.0


def _shift(G, pad, dr, dc, h, w):
    return G[pad + dr:pad + dr + h, pad + dc:pad + dc + w]


def blur(im):
    k = np.exp(-0.5 * (np.arange(-BLUR_R, BLUR_R + 1) / SIGMA) ** 2)
    k /= k.sum()
    h, w = im.shape
    P = np.pad(im, BLUR_R, mode="edge")
    tmp = sum(k[i] * P[BLUR_R:BLUR_R + h, i:i + w] for i in range(2 * BLUR_R + 1))          # along columns
    P = np.pad(tmp, ((BLUR_R, BLUR_R), (0, 0)), mode="edge")
    return sum(k[i] * P[i:i + h, :] for i in range(2 * BLUR_R + 1))                            # along rows


def chess_response(g):
    h, w = g.shape
    pad = 6
    G = np.pad(g, pad, mode="edge")
    I = [_shift(G, pad, dr, dc, h, w) for dr, dc in RING]
    SR = sum(np.abs((I[n] + I[n + 8]) - (I[n + 4] + I[n + 12])) for n in range(4))
    DR = sum(np.abs(I[n] - I[n + 8]) for n in range(8))
    ring_mean = sum(I) / 16.0
    loc = (_shift(G, pad, 0, 0, h, w) + _shift(G, pad, -1, 0, h, w) + _shift(G, pad, 1, 0, h, w)
           + _shift(G, pad, 0, -1, h, w) + _shift(G, pad, 0, 1, h, w)) / 5.0
    R = SR - DR - 16.0 * np.abs(ring_mean - loc)
    R[:BORDER, :] = 0
    R[-BORDER:, :] = 0
    R[:, :BORDER] = 0
    R[:, -BORDER:] = 0
    return R


def peaks(R):
    """-> (M,2) integer (row, col) and responses, raster order."""
    h, w = R.shape
    thr = REL_THRESHOLD * R.max()
    out = []
    cand = np.argwhere(R > max(thr, 0.0))
    for r, c in cand:
        r0, r1, c0, c1 = max(r - NMS_R, 0), min(r + NMS_R, h - 1), max(c - NMS_R, 0), min(c + NMS_R, w - 1)
        v = R[r, c]
        win = R[r0:r1 + 1, c0:c1 + 1]
        if (win > v).any():
            continue
        # ties: the first pixel in raster order wins
        eq = np.argwhere(win == v)
        first = eq[0]
        if (first[0] + r0, first[1] + c0) != (r, c):
            continue
        out.append((r, c, v))
    if not out:
        return np.zeros((0, 2)), np.zeros(0)
    a = np.array(out)
    return a[:, :2].astype(np.float64), a[:, 2]


def _bilinear(im, rr, cc):
    h, w = im.shape
    rr = np.clip(rr, 0, h - 1)
    cc = np.clip(cc, 0, w - 1)
    r0 = np.floor(rr).astype(int)
    c0 = np.floor(cc).astype(int)
    r1 = np.minimum(r0 + 1, h - 1)
    c1 = np.minimum(c0 + 1, w - 1)
    fr, fc = rr - r0, cc - c0
    return ((1 - fr) * (1 - fc) * im[r0, c0] + (1 - fr) * fc * im[r0, c1] + fr * (1 - fc) * im[r1, c0]
            + fr * fc * im[r1, c1])


def subpix_windows(pts):
    """Half-size of cornerSubPix's window per response peak: floor(0.45 d) - 1 for a peak whose nearest other peak is d pixels away, at
    least 3 and at most OpenCV's 11 (the reference's (11, 11), src/detect_fit.jl:19; reached from d = 27 px on).  A neighbouring corner's
    edges inside the window pull the estimate -- 1.3-2.2 px on strongly sheared lattices with the fixed window -- so the window ends short of
    the half-way point to the nearest neighbour (factors 0.4 ... 0.7 measured: profiles/r06_detector_probe.txt)."""
    p = np.asarray(pts, dtype=np.float64)
    if len(p) < 2:
        return np.full(len(p), SUBPIX_WIN, dtype=int)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(axis=2)
    np.fill_diagonal(d2, np.inf)
    return np.clip(np.floor(np.sqrt(d2.min(axis=1)) * SUBPIX_WIN_FRAC).astype(int) - 1, SUBPIX_MIN_WIN, SUBPIX_WIN)


def corner_subpix(im, pts, resp=None):
    """[OCV-K cornerSubPix]: iterate q <- q + G^-1 b over a Gaussian-weighted (2w+1)^2 window of
    image gradients sampled bilinearly around q."""
    h, w = im.shape
    out = np.array(pts, dtype=np.float64)
    wins = subpix_windows(pts) if not os.environ.get("CCAL_DET_FIXED_WIN") else np.full(len(out), SUBPIX_WIN, dtype=int)
    for i, (pr, pc) in enumerate(pts):
        win = int(wins[i])
        m1 = np.exp(-(np.arange(-win, win + 1) ** 2) / float(win * win))
        mask = np.outer(m1, m1)
        off = np.arange(-win, win + 1, dtype=np.float64)
        qr, qc = float(pr), float(pc)
        for _ in range(SUBPIX_ITERS):
            rr = qr + np.arange(-win - 1, win + 2)[:, None] + 0 * np.arange(-win - 1, win + 2)[None, :]
            cc = qc + 0 * np.arange(-win - 1, win + 2)[:, None] + np.arange(-win - 1, win + 2)[None, :]
            patch = _bilinear(im, rr, cc)
            gc = patch[1:-1, 2:] - patch[1:-1, :-2]          # d/dcol (x of the subpix formulas)
            gr = patch[2:, 1:-1] - patch[:-2, 1:-1]          # d/drow
            gcc, gcr, grr = gc * gc * mask, gc * gr * mask, gr * gr * mask
            a, b, c = gcc.sum(), gcr.sum(), grr.sum()
            bb1 = (gcc * off[None, :] + gcr * off[:, None]).sum()
            bb2 = (gcr * off[None, :] + grr * off[:, None]).sum()
            det = a * c - b * b
            if abs(det) <= 1e-300:
                break
            nc = qc + (c * bb1 - b * bb2) / det
            nr = qr + (a * bb2 - b * bb1) / det
            err = (nc - qc) ** 2 + (nr - qr) ** 2
            qr, qc = nr, nc
            if not (0 <= qr < h and 0 <= qc < w):
                break
            if err <= SUBPIX_EPS * SUBPIX_EPS:
                break
        if abs(qr - pr) > SUBPIX_MAX_SHIFT or abs(qc - pc) > SUBPIX_MAX_SHIFT:
            # not believed: the response peak stands -- at its own sub-pixel position (per axis the vertex of the parabola through the
            # peak's response and its two neighbours'), not at the integer pixel
            qr, qc = float(pr), float(pc)
            if resp is not None and not os.environ.get("CCAL_DET_NO_PARABOLA"):
                r, c = int(pr), int(pc)
                r0 = resp[r, c]
                dr = resp[r - 1, c] - 2.0 * r0 + resp[r + 1, c]
                dc = resp[r, c - 1] - 2.0 * r0 + resp[r, c + 1]
                if dr < 0.0:
                    qr += min(0.5, max(-0.5, 0.5 * (resp[r - 1, c] - resp[r + 1, c]) / dr))
                if dc < 0.0:
                    qc += min(0.5, max(-0.5, 0.5 * (resp[r, c - 1] - resp[r, c + 1]) / dc))
        out[i] = (qr, qc)
    return out


def merge_close(pts, resp):
    order = np.argsort(-resp, kind="stable")
    keep = []
    for i in order:
        if all(np.hypot(*(pts[i] - pts[j])) >= MERGE_DIST for j in keep):
            keep.append(i)
    keep = sorted(keep)
    return pts[keep], resp[keep]


def order_grid(pts, resp, n1, n2, im):
    """-> (n1*n2, 2) corners, first index fastest, or None."""
    M = len(pts)
    if M < n1 * n2:
        return None
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)

    def nearest(p):
        q = int(np.argmin(((pts - p) ** 2).sum(1)))
        return q, float(np.sqrt(((pts[q] - p) ** 2).sum()))

    def edge_contrast(p, v, w=None):
        """|gray difference| across the segment p -> p + v at its midpoint, a quarter of its length to either side: a lattice
        AXIS neighbour's segment runs along the edge between a dark and a bright square (large), a diagonal neighbour's through
        the middle of one square and a two-step neighbour's midpoint is itself a corner (both ~0).  w (a window's validation, where
        the lattice is known): the two samples sit a quarter of the OTHER axis' step w to either side instead -- under strong shear
        (axes 30 degrees apart) the squares beside an edge are slivers that the perpendicular leaves"""
        ln = float(np.hypot(v[0], v[1]))
        n = np.array([-v[1], v[0]]) / ln if w is None else np.asarray(w, dtype=np.float64) / ln
        mid = p + 0.5 * v

        def px(q):
            r = min(max(int(round(float(q[0]))), 0), im.shape[0] - 1)
            c = min(max(int(round(float(q[1]))), 0), im.shape[1] - 1)
            return float(im[r, c])
        return abs(px(mid + 0.25 * ln * n) - px(mid - 0.25 * ln * n))

    def x_corners(win):
        """every point of a candidate window is an X corner: an edge runs from it in all four lattice directions (towards the
        neighbour, or -- on the window's border -- the same step outwards: the board's outer ring of squares).  A window shifted by
        a row onto response peaks of the outer squares' L corners has no edge beyond them."""
        g = pts[win]
        m1_, m2_ = win.shape
        cs = []
        for i in range(m1_):
            for j in range(m2_):
                for di, dj in ((1, 0), (-1, 0), (0, 1), (0, -1)):
                    i2, j2 = i + di, j + dj
                    # the other axis' step at this point (towards the inside of the window where there is no neighbour outside)
                    io, jo = (i, j + 1 if j + 1 < m2_ else j - 1) if di != 0 else (i + 1 if i + 1 < m1_ else i - 1, j)
                    w = g[io, jo] - g[i, j]
                    if 0 <= i2 < m1_ and 0 <= j2 < m2_:
                        if di + dj < 0:
                            continue                      # (each inner segment once)
                        v = g[i2, j2] - g[i, j]
                    else:
                        # extrapolated outwards; under perspective consecutive steps along a lattice line form a geometric
                        # progression: the step beyond the border is the last one times the ratio of the last two
                        v = g[i, j] - g[i - di, j - dj]
                        i3, j3 = i - 2 * di, j - 2 * dj
                        if 0 <= i3 < m1_ and 0 <= j3 < m2_:
                            prev = g[i - di, j - dj] - g[i3, j3]
                            v = v * min(1.5, max(0.5, float(np.hypot(v[0], v[1]) / np.hypot(prev[0], prev[1]))))
                    cs.append(edge_contrast(g[i, j], v, w))
        cs = np.array(cs)
        return bool(cs.min() >= 0.25 * np.median(cs))

    seeds = np.argsort(((pts - pts.mean(0)) ** 2).sum(1), kind="stable")[:5]
    best = None
    for seed in seeds:
        # the lattice's two axis directions at the seed: of its 8 nearest neighbours the one whose segment has the strongest edge
        # contrast, and the strongest one not collinear with it.  (Nearest-neighbour geometry alone -- rounds 1-4: the nearest
        # point and the first of the next three within 60 degrees of perpendicular -- picks a diagonal under strong perspective,
        # where a foreshortened lattice's diagonal is shorter than its long axis: a sheared basis indexes the board as a
        # parallelogram and no full n1 x n2 window exists; 13 of 100 boards at tilts up to 0.9 rad, all lost here.)
        nb = np.argsort(d2[seed], kind="stable")[1:9]
        v = pts[nb] - pts[seed]
        con = np.array([edge_contrast(pts[seed], q) for q in v])
        ia = int(np.argmax(con))                          # (first maximum: the nearer neighbour on ties)
        a, b, cb = v[ia], None, -1.0
        for k, q in enumerate(v):
            if k != ia and abs(a @ q) < 0.85 * np.linalg.norm(a) * np.linalg.norm(q) and con[k] > cb:
                b, cb = q, con[k]
        if b is None or cb < 0.25 * con[ia]:
            continue
        coord, basis, queue = {int(seed): (0, 0)}, {int(seed): (a, b)}, [int(seed)]
        while queue:
            p = queue.pop(0)
            (i, j), (u, w_) = coord[p], basis[p]
            for di, dj, step in ((1, 0, u), (-1, 0, -u), (0, 1, w_), (0, -1, -w_)):
                q, dq = nearest(pts[p] + step)
                if dq < 0.35 * np.linalg.norm(step) and q not in coord:
                    coord[q] = (i + di, j + dj)
                    actual = pts[q] - pts[p]
                    basis[q] = (actual * di, w_) if di != 0 else (u, actual * dj)
                    queue.append(q)
        if len(coord) < n1 * n2:
            continue
        ks = np.array(list(coord.keys()))
        ij = np.array([coord[k] for k in ks])
        ij -= ij.min(0)
        ext = ij.max(0) + 1
        occ = -np.ones(tuple(ext), dtype=int)
        occ[ij[:, 0], ij[:, 1]] = ks
        found = None
        for m1, m2, swap in ((n1, n2, False), (n2, n1, True)):
            if ext[0] < m1 or ext[1] < m2:
                continue
            score = -1.0
            for o1 in range(ext[0] - m1 + 1):
                for o2 in range(ext[1] - m2 + 1):
                    win = occ[o1:o1 + m1, o2:o2 + m2]
                    if (win < 0).any() or not x_corners(win):
                        continue
                    sc = resp[win].sum()
                    if sc > score:
                        score, found = sc, (win.T if swap else win)
            if found is not None:
                break
        if found is not None:
            best = pts[found]
            break
    if best is None:
        return None
    grid = best
    du, dv = grid[1, 0] - grid[0, 0], grid[0, 1] - grid[0, 0]
    if du[0] * dv[1] - du[1] * dv[0] < 0:             # keep the ordering right-handed in (row, col)
        grid = grid[::-1, :]

    def outer_intensity(g):
        q = g[0, 0] - 0.5 * (g[1, 0] - g[0, 0]) - 0.5 * (g[0, 1] - g[0, 0])
        r = min(max(int(round(q[0])), 1), im.shape[0] - 2)
        c = min(max(int(round(q[1])), 1), im.shape[1] - 2)
        return im[r - 1:r + 2, c - 1:c + 2].mean()

    cands = [grid, grid[::-1, ::-1]]
    g = cands[int(np.argmin([outer_intensity(x) for x in cands]))]
    return g.transpose(1, 0, 2).reshape(-1, 2)


def detect_corners(im, n_corners):
    """im: 2-D array (rows, cols) of gray values.  -> (P, 2) RowCol corners or None."""
    im = np.asarray(im, dtype=np.float64)
    R = chess_response(blur(im))
    pts, resp = peaks(R)
    if len(pts) < n_corners[0] * n_corners[1]:
        return None
    pts = corner_subpix(im, pts, R)
    pts, resp = merge_close(pts, resp)
    return order_grid(pts, resp, n_corners[0], n_corners[1], im)
