"""Host logic of the exact accumulation (csrc/ccal_acc.hpp): the same functions the kernels use to add the
reduced intrinsic system across workgroups, built as a host shim and checked against exact rational
arithmetic.  CPU only."""
import ctypes as C
import os
import shutil
import subprocess
from fractions import Fraction

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    so = str(tmp_path_factory.mktemp("acc") / "libacc_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Werror", "-Wno-unknown-pragmas", "-ffp-contract=off", "-o", so,
                           os.path.join(HERE, "host_shim_acc.cpp")])
    L = C.CDLL(so)
    L.shim_sum.restype = C.c_double
    L.shim_sum.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.shim_gj_solve9.restype = C.c_int
    L.shim_gj_solve9.argtypes = [C.POINTER(C.c_double)] * 3
    L.shim_split.argtypes = [C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
    dp = C.POINTER(C.c_double)
    L.shim_point_rows.restype = None
    L.shim_point_rows.argtypes = [dp, C.c_double, C.c_double, dp, dp, dp, C.c_double, dp, dp]
    return L


def acc_sum(L, xs, R=8):
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    flags = (C.c_int * 2)()
    v = L.shim_sum(xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs), R, flags)
    return v, (flags[0], flags[1])


def exact(xs):
    lsb = Fraction(1, 2 ** 150)
    tot = Fraction(0)
    for x in xs:
        f = Fraction(float(x))
        q = abs(f) // lsb * lsb          # the kernels drop bits below 2^-150 towards zero
        tot += q if f >= 0 else -q
    return tot


def ulp(v):
    return float(np.spacing(abs(v))) if v != 0 else 0.0


def test_split_reconstructs_the_value(shim):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.normal(size=200) * 10.0 ** rng.integers(-30, 30, 200), [0.0, -0.0, 1.0, -1.0, 2.0 ** 108, -2.0 ** -149,
                         2.0 ** -150, 2.0 ** -151, 1e-320]])
    for x in xs:
        j, lo, hi, fl = C.c_int(), C.c_longlong(), C.c_longlong(), C.c_int()
        shim.shim_split(float(x), C.byref(j), C.byref(lo), C.byref(hi), C.byref(fl))
        assert fl.value == 0 and 0 <= j.value <= 3
        assert abs(lo.value) < 2 ** 52 and abs(hi.value) < 2 ** 52
        got = Fraction(lo.value) * Fraction(2) ** (52 * j.value - 150) + Fraction(hi.value) * Fraction(2) ** (52 * (j.value + 1) - 150)
        assert got == exact([x])
    for x, want in ((float("nan"), 1), (float("inf"), 2), (-float("inf"), 2), (2.0 ** 109, 2), (-1e300, 2)):
        j, lo, hi, fl = C.c_int(), C.c_longlong(), C.c_longlong(), C.c_int()
        shim.shim_split(x, C.byref(j), C.byref(lo), C.byref(hi), C.byref(fl))
        assert fl.value == want and lo.value == 0 and hi.value == 0


@pytest.mark.parametrize("R", [1, 8, 64])
def test_sum_is_exact_and_order_independent(shim, R):
    rng = np.random.default_rng(7 + R)
    for case in range(30):
        n = int(rng.integers(1, 1024))
        scale = 10.0 ** rng.integers(-25, 25)
        xs = rng.normal(size=n) * scale * 10.0 ** rng.integers(-8, 8, n)
        if case % 3 == 0:
            xs = np.concatenate([xs, -xs[: n // 2] * (1 + 1e-13)])     # heavy cancellation
        v, flags = acc_sum(shim, xs, R)
        assert flags == (0, 0)
        ex = exact(xs)
        assert abs(Fraction(v) - ex) <= Fraction(ulp(v)) * 2 + Fraction(1, 2 ** 150)
        for _ in range(3):
            assert acc_sum(shim, rng.permutation(xs), R)[0] == v            # bit-identical in any order
        assert acc_sum(shim, xs, 1)[0] == v                                 # ... and for any replica count: the decoded
                                                                            # value is a function of the exact total


def test_many_adds_do_not_overflow(shim):
    # the sweep grid is capped at 1024 workgroups, i.e. at most 1024 adds per column over all replicas: the largest
    # admissible magnitude, all of one sign, 1024 times (the top limb then exceeds 53 bits: rounded, not wrapped)
    xs = np.full(1024, 2.0 ** 108 * (2 - 2.0 ** -52))
    for R in (1, 8, 64):
        v, flags = acc_sum(shim, xs, R)
        assert flags == (0, 0) and abs(Fraction(v) - exact(xs)) <= 2 * Fraction(ulp(v))
        assert acc_sum(shim, -xs, R)[0] == -v
    rng = np.random.default_rng(3)
    xs = rng.uniform(0.5, 1.0, 1024) * 2.0 ** 40
    v, _ = acc_sum(shim, xs, 8)
    assert abs(Fraction(v) - exact(xs)) <= Fraction(ulp(v))


@pytest.mark.parametrize("n", [4096, 6000, 12000])
def test_replica_totals_beyond_an_int64(shim, n):
    # The focal sums of the initialisation take one add per image (up to LM_FOC_MAX_IMAGES = 12000 on one GPU) and a sweep grid of
    # more than 2^11 workgroups one per workgroup: all of one sign, low limbs near 2^52 each, the total over the 8 replicas is
    # past 2^63 from ~4096 adds on.  The readers split every replica on its own (round 4 added them up as one int64 first and
    # came out wrong by 2^64 limb units at 6000 and 12000 adds -- advisor's finding).
    rng = np.random.default_rng(n)
    xs = rng.uniform(0.75, 1.0, n) * 2.0 ** -99          # mantissas fill limb 0's 52 payload bits
    v, flags = acc_sum(shim, xs, 8)
    assert flags == (0, 0)
    assert abs(Fraction(v) - exact(xs)) <= 2 * Fraction(ulp(v))
    assert acc_sum(shim, xs, 64)[0] == v and acc_sum(shim, rng.permutation(xs), 8)[0] == v
    xs = rng.uniform(0.5, 1.0, n) * 2.3e5                # magnitudes of the AtA diagonal of the focal least squares
    v, _ = acc_sum(shim, xs, 8)
    assert abs(Fraction(v) - exact(xs)) <= Fraction(ulp(v))


def test_flags(shim):
    v, flags = acc_sum(shim, [1.0, float("nan"), 2.0, float("inf"), 1e40, 3.0])
    assert flags == (1, 2) and v == 6.0


def _exact_solve(S, g):
    n = len(g)
    A = [[Fraction(float(S[i, j])) for j in range(n)] + [Fraction(float(g[i]))] for i in range(n)]
    for j in range(n):
        p = A[j][j]
        A[j] = [v / p for v in A[j]]
        for i in range(n):
            if i != j:
                f = A[i][j]
                A[i] = [a - f * b for a, b in zip(A[i], A[j])]
    return np.array([float(A[i][n]) for i in range(n)])


def _gj(L, S, g):
    S = np.ascontiguousarray(S, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    x = np.zeros(9)
    dp = C.POINTER(C.c_double)
    ok = L.shim_gj_solve9(S.ctypes.data_as(dp), g.ctypes.data_as(dp), x.ctypes.data_as(dp))
    return ok, x


def test_scaled_gauss_jordan_matches_exact_solution(shim):
    """csrc/ccal_math.hpp gj_row_step: the division-free elimination the LM advance uses for the damped 9x9 Schur
    system, on SPD matrices with the scale spread of real reduced systems (f^2-sized entries next to distortion ones)."""
    rng = np.random.default_rng(11)
    worst = 0.0
    for case in range(200):
        J = rng.normal(size=(40, 9))
        scales = 10.0 ** rng.uniform(-6, 7, 9)            # columns of very different magnitude
        J = J * scales
        if case % 4 == 0:
            J[:, 5] = J[:, 4] * 1.001 + 1e-4 * J[:, 5]    # strongly correlated pair (k1, k2)
        S = J.T @ J
        S[np.diag_indices(9)] *= 1 + 10.0 ** rng.integers(-16, 3)
        g = J.T @ rng.normal(size=40)
        if case % 5 == 0:                                 # fixed parameters: identity rows, zero right-hand side
            for a_ in rng.choice(9, 3, replace=False):
                S[a_, :] = 0; S[:, a_] = 0; S[a_, a_] = 1; g[a_] = 0
        ok, x = _gj(shim, S, g)
        assert ok == 1
        xe = _exact_solve(S, g)
        D = np.sqrt(np.diag(S))
        cond = np.linalg.cond(S / np.outer(D, D))
        err = np.abs(x - xe).max() / np.abs(xe).max() if np.abs(xe).max() > 0 else 0.0
        assert err < 1e-14 * max(cond, 10), (case, err, cond)
        worst = max(worst, err / cond)
        assert np.all(np.isfinite(x))
    # tiny and huge overall scales: the per-step rescaling keeps everything in range
    for scale in (1e-140, 1e140):
        J = rng.normal(size=(30, 9))
        S, g = (J.T @ J) * scale, (J.T @ rng.normal(size=30)) * scale
        ok, x = _gj(shim, S, g)
        assert ok == 1 and np.abs(x - _exact_solve(S, g)).max() < 1e-12 * np.abs(x).max()


def test_scaled_gauss_jordan_flags_indefinite_systems(shim):
    rng = np.random.default_rng(12)
    J = rng.normal(size=(30, 9))
    S = J.T @ J
    S[4, 4] = -S[4, 4]
    assert _gj(shim, S, np.ones(9))[0] == 0
    S = J.T @ J
    S[:, 8] = S[:, 7]; S[8, :] = S[7, :]                  # singular: a zero (or rounding-sized negative) pivot may slip ...
    S[8, 8] = S[7, 7] * (1 - 1e-6)                        # ... so make it clearly indefinite
    assert _gj(shim, S, np.ones(9))[0] == 0
    S = J.T @ J
    S[2, 2] = np.nan
    assert _gj(shim, S, np.ones(9))[0] == 0


def test_factored_jacobian_rows_match_the_plain_form_and_the_oracle(shim):
    """csrc/ccal_math.hpp:point_rows_K -- the kernels' rows with the twelve pose entries factored through one 2x2 distortion
    Jacobian per corner (W = diag(fx, fy) Jd z [1 0 -x; 0 1 -y]; translation columns = W's columns; rotation columns
    K[:,j] . ((R M) x W_r)) -- against (i) the plain form with the 3x9 Rodrigues Jacobian of SURVEY.md A.2 / A.3 built from
    the same header, entry by entry, and (ii) the ORACLE's per-image Gram matrix assembled from these rows."""
    from cameracalibrations_amd import synthetic as syn
    from oracle import oracle as orc
    dp = C.POINTER(C.c_double)

    def rows(M, uv, r, t, intr, aspect):
        a, b = np.zeros((2, 16)), np.zeros((2, 16))
        arr = [np.ascontiguousarray(x, dtype=np.float64) for x in (M, r, t, intr)]
        shim.shim_point_rows(arr[0].ctypes.data_as(dp), float(uv[0]), float(uv[1]), arr[1].ctypes.data_as(dp), arr[2].ctypes.data_as(dp),
                             arr[3].ctypes.data_as(dp), float(aspect), a.ctypes.data_as(dp), b.ctypes.data_as(dp))
        return a, b
    rng = np.random.default_rng(12)
    worst = 0.0
    for trial in range(3000):
        r = rng.uniform(-0.7, 0.7, 3) * (4.0 if trial % 50 == 0 else 1.0)          # now and then beyond one radian
        t = np.r_[rng.uniform(-3, 3, 2), rng.uniform(8, 16)]
        intr = np.r_[1610.0, 1600.0, 750.0, 1000.0, rng.uniform(-0.2, 0.2), rng.uniform(-0.05, 0.05), rng.uniform(-2e-3, 2e-3, 2),
                     rng.uniform(-0.01, 0.01)]
        M = np.r_[rng.uniform(0, 14), rng.uniform(0, 9), 0.0 if trial % 2 else rng.uniform(-1, 1)]
        aspect = 1.2 if trial % 3 == 0 else 0.0
        f, p = rows(M, (700.0, 900.0), r, t, intr, aspect)
        scale = np.maximum(np.abs(p), 1e-3)
        worst = max(worst, float((np.abs(f - p) / scale).max()))
    assert worst < 1e-10, worst
    # (ii) a whole image's Gram from the factored rows against the oracle's
    wl = syn.named_workload("C2", n_images=3)
    intr = wl.intr_true.copy()
    for i in range(3):
        G = np.zeros((16, 16))
        for pnt in range(wl.n_points):
            f, _ = rows(wl.obj[pnt].astype(np.float64), wl.img[i, pnt].astype(np.float64), wl.rvecs_true[i], wl.tvecs_true[i], intr, 0.0)
            G += np.outer(f[0], f[0]) + np.outer(f[1], f[1])
        Go = orc.image_gram(wl.obj, wl.img[i], wl.rvecs_true[i], wl.tvecs_true[i], intr, 0.0)
        d = np.sqrt(np.abs(np.diag(Go))) + 1e-300
        assert (np.abs(G - Go) / np.outer(d, d)).max() < 1e-11
