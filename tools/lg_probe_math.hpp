// lg_probe_math.hpp -- the sweep's per-corner arithmetic in "lane-group" form (probe only: tools/lg_probe.hip, profiles/r06_lane_group_probe.txt).
//
// k_lm_trial's default form gives an image a whole wavefront: one corner per lane, the two 16-wide rows of [Je | Ji | e] go through
// an LDS tile into v_mfma_f64_16x16x4_f64, and everything that is the same for all corners of the image (pose update, rotation,
// 6 x 6 factor, Schur outputs) is computed by all 64 lanes alike.  That is the right shape while there are about as many images as
// SIMDs -- the matrix instruction sums over the lanes for free and the image's chain is short.  With ten images per SIMD (10000 x
// 150, BASELINE.json configs[4] on one GPU) the launch is bound by the FP64 instructions it issues, and two
// thirds of them are avoidable: the matrix instruction computes all 256 entries of a Gram matrix that has 132 distinct non-zero ones
// at the same 32 flop per cycle the vector unit has, and the image-uniform work is done 64 times.
//
// Lane-group form: a wave holds 64 / L images, L lanes each; a lane walks over every L-th corner of its image and keeps the Gram
// sums of its corners in registers -- no LDS tile, no matrix instruction, no padding lanes --, the L partial sums meet through
// log2(L) DPP steps, and the image-uniform work is shared by 64 / L images per instruction.  132 sums are 264 vector registers, so
// the corners are walked twice: pass A accumulates the blocks that involve the pose columns (81 sums), pass B the intrinsic block
// (51 sums, which are only ever needed summed over ALL images and therefore stay in the lane across images).
//
// Columns as in ccal_math.hpp: 0-2 rvec, 3-5 tvec, 6 fx, 7 fy, 8 cx, 9 cy, 10 k1, 11 k2, 12 p1, 13 p2, 14 k3, 15 residual.
// Two things are taken out of the per-corner arithmetic because they are linear maps that commute with the sum over corners:
//   * the rotation columns.  point_rows_K forms them as K^T (RM x W_r); here the sums run over the "pre-rotation" columns
//     c~ = [RM x W_r | W_r] and T = blockdiag(K, I3) is applied once per image to the summed blocks (lg_apply_T);
//   * CALIB_FIX_ASPECT_RATIO.  The sums are formed as if fx were free (column 6 = xd in u rows, column 7 = yd in v rows: then the
//     u rows have zeros in columns 7, 9 and the v rows in 6, 8 -- 4 structural zeros of the 9 x 9 block, and 2 x 7 instead of 2 x 9
//     products per pose column); the fixed-aspect fy column  aspect * (fx column) + (fy column)  is folded in afterwards
//     (lg_fold_aspect).
// Plain inlined C++ over fixed-size arrays with compile-time indexing (registers on the device); __host__ __device__ so that
// tests/ can run the same arithmetic on the CPU against point_rows_K and the oracle.
#pragma once

#include "ccal_math.hpp"

namespace ccal {

constexpr int LG_NA = 21, LG_NB = 54, LG_NH = 6;             // pass A: A~ (6 x 6 packed lower), B~ (6 x 9), h~ (6)
constexpr int LG_NC = 45, LG_NG = 9;                         // pass B: C (9 x 9 packed lower, 4 entries structurally zero), g (9), e.e

// which intrinsic slots (0..8 = columns 6..14) a u row / a v row touches, free-aspect form
CCAL_HD constexpr bool lg_in_u(int a) { return a != 1 && a != 3; }
CCAL_HD constexpr bool lg_in_v(int a) { return a != 0 && a != 2; }

struct LgPoint {
    double ju[9], jv[9];      // intrinsic columns of the u / v row (entries a row does not touch are not set)
    double eu, ev;
    double x, y, z, cd, D;    // normalised point, 1/Z, radial factor and its derivative w.r.t. r^2
    double Xr, Yr, Zr;        // R M
};

// projection, residual and the intrinsic columns, free-aspect form
CCAL_HD void lg_project(const double M[3], double u_obs, double v_obs, const double R[9], const double t[3], const double in[NINTR],
                        LgPoint &q) {
    const double fx = in[0], fy = in[1], cx = in[2], cy = in[3];
    const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    q.Xr = R[0] * M[0] + R[1] * M[1] + R[2] * M[2];
    q.Yr = R[3] * M[0] + R[4] * M[1] + R[5] * M[2];
    q.Zr = R[6] * M[0] + R[7] * M[1] + R[8] * M[2];
    const double X = q.Xr + t[0], Y = q.Yr + t[1], Z = q.Zr + t[2];
    const double z = (Z != 0.0) ? 1.0 / Z : 1.0;
    const double x = X * z, y = Y * z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2.0 * x * y, a2 = r2 + 2.0 * x * x, a3 = r2 + 2.0 * y * y;
    const double cd = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const double xd = x * cd + p1 * a1 + p2 * a2;
    const double yd = y * cd + p1 * a3 + p2 * a1;
    q.x = x; q.y = y; q.z = z; q.cd = cd;
    q.D = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4;
    q.eu = fx * xd + cx - u_obs;
    q.ev = fy * yd + cy - v_obs;
    const double fxx = fx * x, fyy = fy * y;
    q.ju[0] = xd; q.ju[2] = 1.0; q.ju[4] = fxx * r2; q.ju[5] = fxx * r4; q.ju[6] = fx * a1; q.ju[7] = fx * a2; q.ju[8] = fxx * r6;
    q.jv[1] = yd; q.jv[3] = 1.0; q.jv[4] = fyy * r2; q.jv[5] = fyy * r4; q.jv[6] = fy * a3; q.jv[7] = fy * a1; q.jv[8] = fyy * r6;
    q.ju[1] = q.ju[3] = 0.0; q.jv[0] = q.jv[2] = 0.0;        // (never read: lg_in_u / lg_in_v)
}

// the pre-rotation pose columns c~ = [RM x W_r | W_r] of the u and v row (point_rows_K's W and q, without K)
CCAL_HD void lg_pose_columns(const LgPoint &q, const double in[NINTR], double cu[6], double cv[6]) {
    const double fx = in[0], fy = in[1], p1 = in[6], p2 = in[7];
    const double x = q.x, y = q.y;
    const double D2 = q.D + q.D, xD2 = x * D2, yD2 = y * D2;
    const double p1_2 = 2.0 * p1, p2_2 = 2.0 * p2, p1_6 = 6.0 * p1, p2_6 = 6.0 * p2;
    const double Auu = fma(p2_6, x, fma(p1_2, y, fma(x, xD2, q.cd)));
    const double Auv = fma(p2_2, y, fma(p1_2, x, x * yD2));
    const double Avv = fma(p2_2, x, fma(p1_6, y, fma(y, yD2, q.cd)));
    const double zx = q.z * fx, zy = q.z * fy;
    const double w00 = zx * Auu, w01 = zx * Auv, w02 = -fma(w01, y, w00 * x);
    const double w10 = zy * Auv, w11 = zy * Avv, w12 = -fma(w11, y, w10 * x);
    cu[3] = w00; cu[4] = w01; cu[5] = w02;
    cv[3] = w10; cv[4] = w11; cv[5] = w12;
    cu[0] = q.Yr * w02 - q.Zr * w01; cu[1] = q.Zr * w00 - q.Xr * w02; cu[2] = q.Xr * w01 - q.Yr * w00;
    cv[0] = q.Yr * w12 - q.Zr * w11; cv[1] = q.Zr * w10 - q.Xr * w12; cv[2] = q.Xr * w11 - q.Yr * w10;
}

// pass A, one corner: A~ += c~ c~^T (packed lower), B~ += c~ ji^T, h~ += c~ e   (138 multiply-adds; 81 sums)
CCAL_HD void lg_accum_A(const LgPoint &q, const double cu[6], const double cv[6], double At[LG_NA], double Bt[LG_NB], double ht[LG_NH]) {
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) At[tri(a, b)] = fma(cv[a], cv[b], fma(cu[a], cu[b], At[tri(a, b)]));
#pragma unroll
        for (int s = 0; s < 9; s++) {
            if (lg_in_u(s)) Bt[a * 9 + s] = fma(cu[a], q.ju[s], Bt[a * 9 + s]);
            if (lg_in_v(s)) Bt[a * 9 + s] = fma(cv[a], q.jv[s], Bt[a * 9 + s]);
        }
        ht[a] = fma(cv[a], q.ev, fma(cu[a], q.eu, ht[a]));
    }
}
// ... and its two halves, for register budgets that do not hold 81 sums beside the rows: pass A1 = B~ alone (84 multiply-adds, 54
// sums), pass A2 = A~ and h~ (54 multiply-adds, 27 sums)
CCAL_HD void lg_accum_A1(const LgPoint &q, const double cu[6], const double cv[6], double Bt[LG_NB]) {
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int s = 0; s < 9; s++) {
            if (lg_in_u(s)) Bt[a * 9 + s] = fma(cu[a], q.ju[s], Bt[a * 9 + s]);
            if (lg_in_v(s)) Bt[a * 9 + s] = fma(cv[a], q.jv[s], Bt[a * 9 + s]);
        }
}
CCAL_HD void lg_accum_A2(const LgPoint &q, const double cu[6], const double cv[6], double At[LG_NA], double ht[LG_NH]) {
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) At[tri(a, b)] = fma(cv[a], cv[b], fma(cu[a], cu[b], At[tri(a, b)]));
        ht[a] = fma(cv[a], q.ev, fma(cu[a], q.eu, ht[a]));
    }
}

// pass B, one corner: C += ji ji^T (packed lower over the slots 0..8), g += ji e, ee += e.e   (72 multiply-adds)
CCAL_HD void lg_accum_B(const LgPoint &q, double Ct[LG_NC], double gt[LG_NG], double &ee) {
#pragma unroll
    for (int a = 0; a < 9; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) {
            if (lg_in_u(a) && lg_in_u(b)) Ct[tri(a, b)] = fma(q.ju[a], q.ju[b], Ct[tri(a, b)]);
            if (lg_in_v(a) && lg_in_v(b)) Ct[tri(a, b)] = fma(q.jv[a], q.jv[b], Ct[tri(a, b)]);
        }
        if (lg_in_u(a)) gt[a] = fma(q.ju[a], q.eu, gt[a]);
        if (lg_in_v(a)) gt[a] = fma(q.jv[a], q.ev, gt[a]);
    }
    ee = fma(q.ev, q.ev, fma(q.eu, q.eu, ee));
}

// T = blockdiag(K, I3) applied to the summed pose blocks of one image: A = T^T A~ T, B = T^T B~, h = T^T h~
// (rotation columns: Je_r = c~_r K, i.e. entry j of a row is sum_i c~_i K[i][j] -- K row-major as rodrigues_RK returns it)
CCAL_HD void lg_apply_T(const double K[9], double At[LG_NA], double Bt[LG_NB], double ht[LG_NH]) {
    // rows 0..2 of [A~ | B~ | h~] (and, by symmetry, columns 0..2 of A~) are mixed by K^T
    double Arr[9], Art[9];                                   // full 3 x 3 copies: rr block (symmetric), tr block A~[3 + i][j] (i: t, j: r)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Arr[i * 3 + j] = At[i >= j ? tri(i, j) : tri(j, i)];
            Art[i * 3 + j] = At[tri(3 + i, j)];
        }
    double tmp[9];
    // rr: K^T Arr K
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) tmp[i * 3 + j] = Arr[i * 3 + 0] * K[0 * 3 + j] + Arr[i * 3 + 1] * K[1 * 3 + j] + Arr[i * 3 + 2] * K[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) At[tri(i, j)] = K[0 * 3 + i] * tmp[0 * 3 + j] + K[1 * 3 + i] * tmp[1 * 3 + j] + K[2 * 3 + i] * tmp[2 * 3 + j];
    // tr: Art K
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) At[tri(3 + i, j)] = Art[i * 3 + 0] * K[0 * 3 + j] + Art[i * 3 + 1] * K[1 * 3 + j] + Art[i * 3 + 2] * K[2 * 3 + j];
    // B rows 0..2 and h[0..2]: K^T
#pragma unroll
    for (int s = 0; s < 9; s++) {
        const double b0 = Bt[0 * 9 + s], b1 = Bt[1 * 9 + s], b2 = Bt[2 * 9 + s];
#pragma unroll
        for (int j = 0; j < 3; j++) Bt[j * 9 + s] = K[0 * 3 + j] * b0 + K[1 * 3 + j] * b1 + K[2 * 3 + j] * b2;
    }
    {
        const double h0 = ht[0], h1 = ht[1], h2 = ht[2];
#pragma unroll
        for (int j = 0; j < 3; j++) ht[j] = K[0 * 3 + j] * h0 + K[1 * 3 + j] * h1 + K[2 * 3 + j] * h2;
    }
}

// CALIB_FIX_ASPECT_RATIO on the pose-intrinsic block of one image: fy column <- aspect * (fx column) + (fy column), fx column <- 0
CCAL_HD void lg_fold_aspect_B(double aspect, double Bt[LG_NB]) {
#pragma unroll
    for (int a = 0; a < 6; a++) {
        Bt[a * 9 + 1] = fma(aspect, Bt[a * 9 + 0], Bt[a * 9 + 1]);
        Bt[a * 9 + 0] = 0.0;
    }
}

// ... and on the intrinsic block (sums over any set of images): with f' = aspect * fx-column + fy-column,
//   C[1][1] <- aspect^2 C[0][0] + C[1][1]   (C[1][0] is structurally zero),  C[s][1] <- aspect * C[s][0] + C[s][1]  (s >= 2),
//   g[1] <- aspect * g[0] + g[1];  row / column 0 <- 0
CCAL_HD void lg_fold_aspect_C(double aspect, double Ct[LG_NC], double gt[LG_NG]) {
    Ct[tri(1, 1)] = fma(aspect * aspect, Ct[tri(0, 0)], Ct[tri(1, 1)]);
#pragma unroll
    for (int s = 2; s < 9; s++) {
        Ct[tri(s, 1)] = fma(aspect, Ct[tri(s, 0)], Ct[tri(s, 1)]);
        Ct[tri(s, 0)] = 0.0;
    }
    Ct[tri(0, 0)] = 0.0;
    Ct[tri(1, 0)] = 0.0;
    gt[1] = fma(aspect, gt[0], gt[1]);
    gt[0] = 0.0;
}

}  // namespace ccal
