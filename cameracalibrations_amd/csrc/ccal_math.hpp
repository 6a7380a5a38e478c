// ccal_math.hpp -- per-lane / per-wave-uniform FP64 math of the camera-model fit.
//
// Everything here is plain inlined C++ over scalars and small fixed-size arrays with
// compile-time indexing, so that on gfx950 the arrays live in VGPRs (never scratch).  The
// functions are __host__ __device__ only so that tests/ can build a host shim of the SAME
// math and compare it with the oracle on CPU; the product never runs them on the host.
//
// Model and Jacobians: OpenCV 5-coefficient projection as summarised in SURVEY.md A.2/A.3
// (the arithmetic the reference reaches through src/detect_fit.jl:47); the Julia forward
// model of src/meta.jl:39-44 is the k2 = k3 = p1 = p2 = 0 special case.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CCAL_HD __host__ __device__ __forceinline__
#else
#define CCAL_HD inline
#endif

namespace ccal {

constexpr int NINTR = 9;   // fx fy cx cy k1 k2 p1 p2 k3
constexpr int NCOL = 16;   // Gram columns: 0-5 rvec,tvec | 6-14 intrinsic slots | 15 residual
constexpr int COL_E = 15;

// ---------------------------------------------------------------------------------------
// Rodrigues vector -> rotation matrix (row-major) and d vec(R) / d r_j
// ---------------------------------------------------------------------------------------
CCAL_HD void rodrigues(const double r[3], double R[9], double dR[27]) {
    const double th2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    const double th = sqrt(th2);
    if (th < 2.220446049250313e-16) {
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 27; k++) dR[k] = 0.0;
        dR[5] = dR[15] = dR[19] = -1.0;
        dR[7] = dR[11] = dR[21] = 1.0;
        return;
    }
    const double c = cos(th), s = sin(th), c1 = 1.0 - c, it = 1.0 / th;
    const double n[3] = {r[0] * it, r[1] * it, r[2] * it};
    double nn[9], nx[9] = {0.0, -n[2], n[1], n[2], 0.0, -n[0], -n[1], n[0], 0.0};
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) nn[a * 3 + b] = n[a] * n[b];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * nn[k] + s * nx[k];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        double dn[3];
#pragma unroll
        for (int a = 0; a < 3; a++) dn[a] = (((a == j) ? 1.0 : 0.0) - n[j] * n[a]) * it;
        const double dnx[9] = {0.0, -dn[2], dn[1], dn[2], 0.0, -dn[0], -dn[1], dn[0], 0.0};
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                const int k = a * 3 + b;
                dR[j * 9 + k] = n[j] * (-s * ((a == b) ? 1.0 : 0.0) + s * nn[k] + c * nx[k]) +
                                c1 * (dn[a] * n[b] + n[a] * dn[b]) + s * dnx[k];
            }
    }
}

// Rotation only (error-only evaluations, maps).
CCAL_HD void rodrigues_R(const double r[3], double R[9]) {
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 2.220446049250313e-16) {
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double c = cos(th), s = sin(th), c1 = 1.0 - c, it = 1.0 / th;
    const double n[3] = {r[0] * it, r[1] * it, r[2] * it};
    R[0] = c + c1 * n[0] * n[0];        R[1] = c1 * n[0] * n[1] - s * n[2]; R[2] = c1 * n[0] * n[2] + s * n[1];
    R[3] = c1 * n[1] * n[0] + s * n[2]; R[4] = c + c1 * n[1] * n[1];        R[5] = c1 * n[1] * n[2] - s * n[0];
    R[6] = c1 * n[2] * n[0] - s * n[1]; R[7] = c1 * n[2] * n[1] + s * n[0]; R[8] = c + c1 * n[2] * n[2];
}

// ---------------------------------------------------------------------------------------
// One corner: the two rows (u, v) of [Je | Ji | e]  
//   cols 0-2 d/drvec, 3-5 d/dtvec, 6 fx, 7 fy, 8 cx, 9 cy, 10 k1, 11 k2, 12 p1, 13 p2, 14 k3, 15 e
//   aspect > 0  <=>  CALIB_FIX_ASPECT_RATIO: fx column zero, fy column = (aspect*xd, yd)
// ---------------------------------------------------------------------------------------
CCAL_HD void point_rows(const double M[3], double u_obs, double v_obs, const double R[9],
                        const double dR[27], const double t[3], const double in[NINTR],
                        double aspect, double mu[NCOL], double mv[NCOL]) {
    const double fx = in[0], fy = in[1], cx = in[2], cy = in[3];
    const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    const double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    const double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    const double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    const double z = (Z != 0.0) ? 1.0 / Z : 1.0;
    const double x = X * z, y = Y * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
    const double cd = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = x * cd + p1 * a1 + p2 * a2;
    const double yd = y * cd + p1 * a3 + p2 * a1;
    const double dcd_dr2 = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;

    mu[COL_E] = fx * xd + cx - u_obs;
    mv[COL_E] = fy * yd + cy - v_obs;
    if (aspect > 0.0) { mu[6] = 0.0; mu[7] = xd * aspect; }
    else              { mu[6] = xd;  mu[7] = 0.0; }
    mv[6] = 0.0; mv[7] = yd;
    mu[8] = 1.0; mu[9] = 0.0; mv[8] = 0.0; mv[9] = 1.0;
    mu[10] = fx * x * r2; mu[11] = fx * x * r4; mu[12] = fx * a1; mu[13] = fx * a2; mu[14] = fx * x * r6;
    mv[10] = fy * y * r2; mv[11] = fy * y * r4; mv[12] = fy * a3; mv[13] = fy * a1; mv[14] = fy * y * r6;

#pragma unroll
    for (int j = 0; j < 6; j++) {
        double dX, dY, dZ;
        if (j < 3) {
            const double *D = dR + 9 * j;
            dX = D[0] * M[0] + D[1] * M[1] + D[2] * M[2];
            dY = D[3] * M[0] + D[4] * M[1] + D[5] * M[2];
            dZ = D[6] * M[0] + D[7] * M[1] + D[8] * M[2];
        } else {
            dX = (j == 3) ? 1.0 : 0.0; dY = (j == 4) ? 1.0 : 0.0; dZ = (j == 5) ? 1.0 : 0.0;
        }
        const double dx = z * (dX - x * dZ), dy = z * (dY - y * dZ);
        const double dr2 = 2.0 * x * dx + 2.0 * y * dy;
        const double dcd = dcd_dr2 * dr2;
        const double da1 = 2.0 * (x * dy + y * dx);
        mu[j] = fx * (cd * dx + x * dcd + p1 * da1 + p2 * (dr2 + 4.0 * x * dx));
        mv[j] = fy * (cd * dy + y * dcd + p1 * (dr2 + 4.0 * y * dy) + p2 * da1);
    }
}

// ---------------------------------------------------------------------------------------
// Rotation and its derivative in "left Jacobian" form.  With r = theta n:
//     R = cos(th) I + (sin(th)/th) [r]x + a r r^T           a = (1 - cos th) / th^2
//     K = (sin(th)/th) I + a [r]x + b r r^T  (= J_l(r))      b = (th - sin th) / th^3
// and  d(R M)/d r_j = K[:,j] x (R M)  -- the same derivative as the 3x9 Rodrigues Jacobian of
// SURVEY.md A.2 contracted with M (checked against it in tests), at a fraction of the
// wave-uniform arithmetic.  Series below th^2 = 1e-2 avoid cancellation in a and b.
// ---------------------------------------------------------------------------------------
CCAL_HD void rodrigues_RK(const double r[3], double R[9], double K[9]) {
    const double t2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    double sc, c, a, b;   // sc = sin(th)/th
    if (t2 <= 10.0) {
        // th <= pi (every rotation): alternating Taylor series in th^2, three independent Horner chains, no
        // sqrt / sincos / divide.  18 terms reach an absolute error < 1e-16 up to th^2 = 10; rotations of at
        // most one radian (th^2 <= 1: every pose of an ordinary calibration set) need only 11
        // (first neglected term 1 / 23! = 4e-23): the branch is wave-uniform.
        //   sc = sum (-t2)^k / (2k+1)!   a = sum (-t2)^k / (2k+2)!   b = sum (-t2)^k / (2k+3)!
        sc = 0.0; a = 0.0; b = 0.0;
        // 1/(2k+1)!, highest order first; the other two tables follow by division (folded at compile time)
        const double CS[18] = {1.0 / 1.0333147966386145e40, 1.0 / 8.683317618811886e36, 1.0 / 8.222838654177922e33,
                               1.0 / 8.841761993739702e30, 1.0 / 1.0888869450418352e28, 1.0 / 1.5511210043330986e25,
                               1.0 / 2.585201673888498e22, 1.0 / 5.109094217170944e19, 1.0 / 1.21645100408832e17,
                               1.0 / 3.55687428096e14, 1.0 / 1.307674368e12, 1.0 / 6227020800.0, 1.0 / 39916800.0,
                               1.0 / 362880.0, 1.0 / 5040.0, 1.0 / 120.0, 1.0 / 6.0, 1.0};
        if (t2 > 1.0) {
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int ord = 17 - k;                       // power of (-t2)
                const double cs = CS[k];                      // 1/(2 ord + 1)!
                const double ca = cs / (double)(2 * ord + 2); // 1/(2 ord + 2)!
                const double cb = ca / (double)(2 * ord + 3); // 1/(2 ord + 3)!
                sc = sc * (-t2) + cs;
                a = a * (-t2) + ca;
                b = b * (-t2) + cb;
            }
        }
#pragma unroll
        for (int k = 7; k < 18; k++) {
            const int ord = 17 - k;
            const double cs = CS[k];
            const double ca = cs / (double)(2 * ord + 2);
            const double cb = ca / (double)(2 * ord + 3);
            sc = sc * (-t2) + cs;
            a = a * (-t2) + ca;
            b = b * (-t2) + cb;
        }
        c = 1.0 - t2 * a;
    } else {
        const double th = sqrt(t2);
        double sn;
#if defined(__HIP_DEVICE_COMPILE__)
        sincos(th, &sn, &c);
#else
        sn = sin(th); c = cos(th);
#endif
        const double it2 = 1.0 / t2;
        sc = sn / th;
        a = (1.0 - c) * it2;
        b = (1.0 - sc) * it2;
    }
    const double xx = r[0] * r[0], yy = r[1] * r[1], zz = r[2] * r[2];
    const double xy = r[0] * r[1], xz = r[0] * r[2], yz = r[1] * r[2];
    R[0] = c + a * xx;          R[1] = a * xy - sc * r[2];  R[2] = a * xz + sc * r[1];
    R[3] = a * xy + sc * r[2];  R[4] = c + a * yy;          R[5] = a * yz - sc * r[0];
    R[6] = a * xz - sc * r[1];  R[7] = a * yz + sc * r[0];  R[8] = c + a * zz;
    K[0] = sc + b * xx;         K[1] = b * xy - a * r[2];   K[2] = b * xz + a * r[1];
    K[3] = b * xy + a * r[2];   K[4] = sc + b * yy;         K[5] = b * yz - a * r[0];
    K[6] = b * xz - a * r[1];   K[7] = b * yz + a * r[0];   K[8] = sc + b * zz;
}

// point_rows with the rotation derivative in K form (same columns as point_rows), the six pose columns factored:
// with d(x, y) the derivative of the normalised point, every pose column is
//     (mu_j, mv_j) = diag(fx, fy) Jd (dx_j, dy_j),     (dx_j, dy_j) = z [1 0 -x; 0 1 -y] dX_j
// where Jd = d(xd, yd)/d(x, y) = [Auu Auv; Auv Avv] is the SAME 2x2 distortion Jacobian for all six columns
//     Auu = cd + 2 x^2 D + 2 p1 y + 6 p2 x     Auv = 2 x y D + 2 p1 x + 2 p2 y     Avv = cd + 2 y^2 D + 6 p1 y + 2 p2 x
// (D = d cd / d r^2; collecting the terms of point_rows' per-column expressions).  So W = diag(fx, fy) Jd z [1 0 -x; 0 1 -y]
// (2 x 3) is formed once per corner; the translation columns ARE its columns (dX_j = e_j), and a rotation column is
//     W_r . (K[:,j] x (R M)) = K[:,j] . ((R M) x W_r)          (scalar triple product)
// i.e. two cross products per corner and one 3-term dot per entry: 52 operations for the 12 pose entries instead of
// 144 -- the rows are a third of a sweep's vector arithmetic where the chip is FP64-bound (10000 x 150).  Same
// derivative, different rounding (a few ulp per entry); the residual column is evaluated exactly as before.
// `vm` = 1 for a real corner, 0 for a padding lane: it rides on multiplications the rows make anyway (every Jacobian
// column carries a factor fx or fy), so that a padding lane's two rows come out as exact zeros without 32 value selects.
CCAL_HD void point_rows_K(const double M[3], double u_obs, double v_obs, const double R[9],
                          const double K[9], const double t[3], const double in[NINTR],
                          double aspect, double mu[NCOL], double mv[NCOL], double vm = 1.0) {
    const double fx = in[0], fy = in[1], cx = in[2], cy = in[3];
    const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    const double fxv = fx * vm, fyv = fy * vm;
    const double Xr = R[0] * M[0] + R[1] * M[1] + R[2] * M[2];
    const double Yr = R[3] * M[0] + R[4] * M[1] + R[5] * M[2];
    const double Zr = R[6] * M[0] + R[7] * M[1] + R[8] * M[2];
    const double X = Xr + t[0], Y = Yr + t[1], Z = Zr + t[2];
    const double z = (Z != 0.0) ? 1.0 / Z : 1.0;
    const double x = X * z, y = Y * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
    const double cd = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = x * cd + p1 * a1 + p2 * a2;
    const double yd = y * cd + p1 * a3 + p2 * a1;
    const double dcd_dr2 = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;

    mu[COL_E] = (fx * xd + cx - u_obs) * vm;
    mv[COL_E] = (fy * yd + cy - v_obs) * vm;
    const double xdv = xd * vm;
    if (aspect > 0.0) { mu[6] = 0.0; mu[7] = xdv * aspect; }
    else              { mu[6] = xdv; mu[7] = 0.0; }
    mv[6] = 0.0; mv[7] = yd * vm;
    mu[8] = vm; mu[9] = 0.0; mv[8] = 0.0; mv[9] = vm;
    const double fxx = fxv * x, fyy = fyv * y;
    mu[10] = fxx * r2; mu[11] = fxx * r4; mu[12] = fxv * a1; mu[13] = fxv * a2; mu[14] = fxx * r6;
    mv[10] = fyy * r2; mv[11] = fyy * r4; mv[12] = fyv * a3; mv[13] = fyv * a1; mv[14] = fyy * r6;

    // 2x2 distortion Jacobian (p-multiples are wave-uniform: hoisted out of the corner loop by the compiler)
    const double D2 = dcd_dr2 + dcd_dr2, xD2 = x * D2, yD2 = y * D2;
    const double p1_2 = 2.0 * p1, p2_2 = 2.0 * p2, p1_6 = 6.0 * p1, p2_6 = 6.0 * p2;
    const double Auu = fma(p2_6, x, fma(p1_2, y, fma(x, xD2, cd)));
    const double Auv = fma(p2_2, y, fma(p1_2, x, x * yD2));
    const double Avv = fma(p2_2, x, fma(p1_6, y, fma(y, yD2, cd)));
    // W = diag(fx, fy) Jd z [1 0 -x; 0 1 -y]: its columns are the translation columns
    const double zx = z * fxv, zy = z * fyv;
    const double w00 = zx * Auu, w01 = zx * Auv, w02 = -fma(w01, y, w00 * x);
    const double w10 = zy * Auv, w11 = zy * Avv, w12 = -fma(w11, y, w10 * x);
    mu[3] = w00; mu[4] = w01; mu[5] = w02;
    mv[3] = w10; mv[4] = w11; mv[5] = w12;
    // rotation columns: K[:,j] . ((R M) x W_r)
    const double q0x = Yr * w02 - Zr * w01, q0y = Zr * w00 - Xr * w02, q0z = Xr * w01 - Yr * w00;
    const double q1x = Yr * w12 - Zr * w11, q1y = Zr * w10 - Xr * w12, q1z = Xr * w11 - Yr * w10;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        mu[j] = K[j] * q0x + K[3 + j] * q0y + K[6 + j] * q0z;
        mv[j] = K[j] * q1x + K[3 + j] * q1y + K[6 + j] * q1z;
    }
}

// Residual only (the error-only evaluation of a trial point in the pose polish's lean form).
CCAL_HD void point_residual(const double M[3], double u_obs, double v_obs, const double R[9],
                            const double t[3], const double in[NINTR], double &eu, double &ev) {
    const double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
    const double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
    const double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
    const double z = (Z != 0.0) ? 1.0 / Z : 1.0;
    const double x = X * z, y = Y * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
    const double cd = 1.0 + in[4] * r2 + in[5] * r4 + in[8] * r6;
    eu = in[0] * (x * cd + in[6] * a1 + in[7] * a2) + in[2] - u_obs;
    ev = in[1] * (y * cd + in[6] * a3 + in[7] * a1) + in[3] - v_obs;
}

// ---------------------------------------------------------------------------------------
// Small dense LA on fixed-size, fully unrolled arrays (wave-uniform values in VGPRs)
// ---------------------------------------------------------------------------------------
// packed lower-triangular index, a >= b
CCAL_HD constexpr int tri(int a, int b) { return a * (a + 1) / 2 + b; }

// 1/sqrt(d): on the device one v_rsq_f64 + refinement (ocml rsqrt), no divide
CCAL_HD double rsqrt_pos(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return rsqrt(d);
#else
    return 1.0 / sqrt(d);
#endif
}

// In-place Cholesky of a packed lower SPD matrix: A = L L^T, inv[j] = 1 / L[j][j].
// Division-free (reciprocal square roots); the substitutions below multiply by inv.
// Returns false when a pivot is not strictly positive (or NaN); the factor is then unusable.
template <int N>
CCAL_HD bool chol_packed(double A[N * (N + 1) / 2], double inv[N]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[tri(j, k)] * A[tri(j, k)];
        ok = ok && (d > 0.0);
        const double r = rsqrt_pos(d);
        inv[j] = r;
        A[tri(j, j)] = d * r;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double s = A[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[tri(i, k)] * A[tri(j, k)];
            A[tri(i, j)] = s * r;
        }
    }
    return ok;
}

// y = L^-1 b  (forward substitution), in place
template <int N>
CCAL_HD void fwd_subst(const double L[N * (N + 1) / 2], const double inv[N], double b[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[tri(i, k)] * b[k];
        b[i] = s * inv[i];
    }
}
// x = L^-T y  (back substitution), in place
template <int N>
CCAL_HD void bwd_subst(const double L[N * (N + 1) / 2], const double inv[N], double b[N]) {
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= L[tri(k, i)] * b[k];
        b[i] = s * inv[i];
    }
}

// Marquardt damping of a packed lower matrix: diag *= 1 + lambda (CvLevMarq::step, A.6)
template <int N>
CCAL_HD void damp_diag(double A[N * (N + 1) / 2], double lambda) {
#pragma unroll
    for (int j = 0; j < N; j++) A[tri(j, j)] *= 1.0 + lambda;
}

// ---------------------------------------------------------------------------------------
// Scaled Gauss-Jordan elimination of an SPD system [A | b] without an exact reciprocal, one matrix
// row per caller (the device keeps row i in lane i and broadcasts the pivot row; tests/host_shim_acc.cpp
// loops over the rows).  Step j, for a row that is NOT the pivot row:
//     row <- row * (d s) - (row[j] s) * pivot_row          d = pivot, s ~ 1 / d
// Column j is eliminated because (row[j] s) / (d s) = row[j] / d whatever s is, so s only has to keep
// the magnitudes put: the raw v_rcp_f64 estimate does (no Newton refinement on the dependent chain);
// the two extra roundings are of the size every elimination step makes anyway.  The pivot row itself is
// left alone (`mine`).  `diag` follows this row's own pivot through the later rescalings, so that at the
// end x = b / diag.
// ---------------------------------------------------------------------------------------
template <int N>
CCAL_HD void gj_row_step(int j, bool mine, double d, double s, const double prow[N], double pb, double a[N], double &b,
                         double &diag) {
    const double dn = mine ? 1.0 : d * s;
    const double f = mine ? 0.0 : -(a[j] * s);
    diag = mine ? d : diag * dn;
#pragma unroll
    for (int k = 0; k < N; k++)
        if (k > j) a[k] = fma(prow[k], f, a[k] * dn);
    b = fma(pb, f, b * dn);
}

// ---------------------------------------------------------------------------------------
// Julia-side maps (reference src/meta.jl), single radial coefficient k
// ---------------------------------------------------------------------------------------
// largest real root of c0 + x^2 - x^3 (src/meta.jl:50-57), closed form + Newton polish
CCAL_HD double max_real_root_cubic(double c0) {
    const double p = -1.0 / 3.0, q = -2.0 / 27.0 - c0;
    const double disc = q * q * 0.25 + p * p * p / 27.0;
    double x;
    if (disc > 0.0) {
        const double sq = sqrt(disc);
        x = cbrt(-0.5 * q + sq) + cbrt(-0.5 * q - sq) + 1.0 / 3.0;
    } else {
        const double rho = 2.0 * sqrt(-p / 3.0);
        double arg = 3.0 * q / (p * rho);
        arg = arg > 1.0 ? 1.0 : (arg < -1.0 ? -1.0 : arg);
        x = rho * cos(acos(arg) / 3.0) + 1.0 / 3.0;
    }
#pragma unroll
    for (int it = 0; it < 3; it++) {
        const double fx = ((x - 1.0) * x) * x - c0, dfx = (3.0 * x - 2.0) * x;
        if (dfx != 0.0) x -= fx / dfx;
    }
    return x;
}

// real2image: intrinsic o distort o perspective o extrinsic o scale (src/meta.jl:27-33,39-44)
CCAL_HD void map_forward(const double xyz[3], const double f[2], const double c[2], double k,
                         const double R[9], const double t[3], double inv_cs, double rc[2]) {
    const double s0 = xyz[0] * inv_cs, s1 = xyz[1] * inv_cs, s2 = xyz[2] * inv_cs;
    const double X = R[0] * s0 + R[1] * s1 + R[2] * s2 + t[0];
    const double Y = R[3] * s0 + R[4] * s1 + R[5] * s2 + t[1];
    const double Z = R[6] * s0 + R[7] * s1 + R[8] * s2 + t[2];
    double v0 = X / Z, v1 = Y / Z;
    if (k != 0.0) {
        const double radial = 1.0 + k * (v0 * v0 + v1 * v1);
        v0 *= radial; v1 *= radial;
    }
    rc[0] = f[0] * v0 + c[0];
    rc[1] = f[1] * v1 + c[1];
}

// image2real: inv_scale o inv_extrinsic o inv_perspective o inv_distort o inv_intrinsic
// (src/meta.jl:50-76,82).  ti = -R^T t.
CCAL_HD void map_inverse(const double rc[2], const double f[2], const double c[2], double k,
                         const double R[9], const double ti[3], double cs, double xyz[3]) {
    double v0 = (rc[0] - c[0]) / f[0], v1 = (rc[1] - c[1]) / f[1];
    if (k != 0.0) {
        const double rho = max_real_root_cubic(k * (v0 * v0 + v1 * v1));
        v0 /= rho; v1 /= rho;
    }
    // depth(rc1, t, l) = -t / (l . rc1), l = last row of R^T (src/meta.jl:60-69)
    const double d = -ti[2] / (R[2] * v0 + R[5] * v1 + R[8]);
    const double p0 = d * v0, p1 = d * v1, p2 = d;
    xyz[0] = cs * (R[0] * p0 + R[3] * p1 + R[6] * p2 + ti[0]);
    xyz[1] = cs * (R[1] * p0 + R[4] * p1 + R[7] * p2 + ti[1]);
    xyz[2] = cs * (R[2] * p0 + R[5] * p1 + R[8] * p2 + ti[2]);
}

}  // namespace ccal
