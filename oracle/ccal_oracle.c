/*
 * ccal_oracle.c -- CPU restatement (plain C, FP64) of the calibrateCamera hot path.
 * TEST INFRASTRUCTURE ONLY; see ccal_oracle.h for scope, pinning status and conventions.
 *
 * Citations: `src/...` are files of the reference under /root/reference; [OCV-K] marks
 * the published OpenCV 4.10 calib3d algorithm as summarised in SURVEY.md Appendix A
 * (that source is a third-party dependency, not present in the reference tree).
 */
#include "ccal_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NSLOT 18 /* OpenCV keeps 18 intrinsic slots; slots 9..17 (k4..tauy) stay 0 here */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ small dense LA */

/* Cyclic Jacobi for a symmetric n x n matrix (row-major, destroyed).  Eigenvalues W
 * sorted descending, eigenvectors as ROWS of V (so V is "Vt" of the SVD of a PSD matrix). */
static void jacobi_eig(int n, double *A, double *W, double *V) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += A[i * n + i] * A[i * n + i];
            for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
        }
        if (off <= 1e-32 * diag || off == 0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                double apq = A[p * n + q];
                if (apq == 0) continue;
                double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
                double c = 1 / sqrt(tt * tt + 1), s = tt * c;
                for (int k = 0; k < n; k++) { /* A <- A * G */
                    double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) { /* A <- G^T * A */
                    double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) { /* rows of V are eigenvectors */
                    double vpk = V[p * n + k], vqk = V[q * n + k];
                    V[p * n + k] = c * vpk - s * vqk;
                    V[q * n + k] = s * vpk + c * vqk;
                }
            }
    }
    for (int i = 0; i < n; i++) W[i] = A[i * n + i];
    for (int i = 0; i < n - 1; i++) { /* selection sort, descending */
        int m = i;
        for (int j = i + 1; j < n; j++)
            if (W[j] > W[m]) m = j;
        if (m != i) {
            double tw = W[i]; W[i] = W[m]; W[m] = tw;
            for (int k = 0; k < n; k++) {
                double tv = V[i * n + k]; V[i * n + k] = V[m * n + k]; V[m * n + k] = tv;
            }
        }
    }
}

/* In-place Cholesky A = L L^T (lower, row-major).  Returns 0 ok, -1 not positive definite. */
static int chol_factor(int n, double *A) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(int n, const double *L, double *b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

/* Solve the symmetric system A x = b (A full, row-major, preserved).  Cholesky first;
 * if A is not positive definite fall back to the eigen pseudo-inverse with OpenCV's
 * SVD back-substitution threshold (sum(w) * 2*DBL_EPSILON) -- the stand-in for
 * solve(..., DECOMP_SVD) of CvLevMarq::step [OCV-K]. */
static void sym_solve(int n, const double *A, const double *b, double *x) {
    double *L = (double *)malloc(sizeof(double) * n * n);
    memcpy(L, A, sizeof(double) * n * n);
    memcpy(x, b, sizeof(double) * n);
    if (chol_factor(n, L) == 0) {
        chol_solve(n, L, x);
        free(L);
        return;
    }
    memcpy(L, A, sizeof(double) * n * n);
    double *W = (double *)malloc(sizeof(double) * n), *V = (double *)malloc(sizeof(double) * n * n);
    jacobi_eig(n, L, W, V);
    double thr = 0;
    for (int i = 0; i < n; i++) thr += fabs(W[i]);
    thr *= 2 * DBL_EPSILON;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int k = 0; k < n; k++) {
        if (fabs(W[k]) <= thr) continue;
        double s = 0;
        for (int i = 0; i < n; i++) s += V[k * n + i] * b[i];
        s /= W[k];
        for (int i = 0; i < n; i++) x[i] += s * V[k * n + i];
    }
    free(L); free(W); free(V);
}

static void mat3_mul(const double *A, const double *B, double *C) {
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
    memcpy(C, T, sizeof T);
}

/* ------------------------------------------------------------------ Rodrigues */

/* SURVEY.md A.2 [OCV-K cvRodrigues2]; Julia side uses Rotations.RotationVec, the same
 * exponential map (src/buildcalibrations.jl:3). */
void orc_rodrigues(const double r[3], double R[9], double *J) {
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0);
        if (J) {
            memset(J, 0, 27 * sizeof(double));
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1 - c, it = 1 / theta;
    double n[3] = {r[0] * it, r[1] * it, r[2] * it};
    double nn[9], nx[9] = {0, -n[2], n[1], n[2], 0, -n[0], -n[1], n[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) nn[i * 3 + j] = n[i] * n[j];
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0) + c1 * nn[k] + s * nx[k];
    if (!J) return;
    /* d/dr_j of  c I + c1 n n^T + s [n]x  with dtheta/dr_j = n_j, dn/dr_j = (e_j - n_j n)/theta */
    for (int j = 0; j < 3; j++) {
        double dn[3];
        for (int a = 0; a < 3; a++) dn[a] = ((a == j) - n[j] * n[a]) * it;
        double dnx[9] = {0, -dn[2], dn[1], dn[2], 0, -dn[0], -dn[1], dn[0], 0};
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                int k = a * 3 + b;
                J[j * 9 + k] = n[j] * (-s * (a == b) + s * nn[k] + c * nx[k]) +
                               c1 * (dn[a] * n[b] + n[a] * dn[b]) + s * dnx[k];
            }
    }
}

/* polar factor R = A (A^T A)^(-1/2) == U V^T of the SVD of A */
static void orthonormalise3(const double A[9], double R[9]) {
    double AtA[9], W[3], V[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            AtA[i * 3 + j] = A[i] * A[j] + A[3 + i] * A[3 + j] + A[6 + i] * A[6 + j];
    jacobi_eig(3, AtA, W, V);
    double Sinv[9] = {0};
    for (int k = 0; k < 3; k++) {
        double w = W[k] > 0 ? 1 / sqrt(W[k]) : 0;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Sinv[i * 3 + j] += w * V[k * 3 + i] * V[k * 3 + j];
    }
    mat3_mul(A, Sinv, R);
}

void orc_rodrigues_inv(const double Rin[9], double r[3]) {
    double R[9];
    orthonormalise3(Rin, R);
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1 ? 1 : (c < -1 ? -1 : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (R[0] + 1) * 0.5; r[0] = sqrt(t > 0 ? t : 0);
        t = (R[4] + 1) * 0.5; r[1] = sqrt(t > 0 ? t : 0) * (R[1] < 0 ? -1.0 : 1.0);
        t = (R[8] + 1) * 0.5; r[2] = sqrt(t > 0 ? t : 0) * (R[2] < 0 ? -1.0 : 1.0);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && ((R[5] > 0) != (r[1] * r[2] > 0)))
            r[2] = -r[2];
        double nrm = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        theta /= nrm;
        r[0] *= theta; r[1] *= theta; r[2] *= theta;
        return;
    }
    double vth = theta / (2 * s);
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

/* ------------------------------------------------------------------ projection */

/* SURVEY.md A.3 [OCV-K cvProjectPoints2Internal], 5-coefficient model.  The Julia
 * forward model (src/meta.jl:39-44, src/buildcalibrations.jl:1-6) is this with
 * k2 = k3 = p1 = p2 = 0. */
void orc_project_points(int P, const double *obj, const double rv[3], const double t[3],
                        const double in[ORC_NINTR], double aspect, double *proj, double *dpdr,
                        double *dpdt, double *dpdf, double *dpdc, double *dpdk) {
    double R[9], dRdr[27];
    orc_rodrigues(rv, R, dpdr ? dRdr : NULL);
    const double fx = in[0], fy = in[1], cx = in[2], cy = in[3];
    const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    for (int p = 0; p < P; p++) {
        const double M[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
        double X = R[0] * M[0] + R[1] * M[1] + R[2] * M[2] + t[0];
        double Y = R[3] * M[0] + R[4] * M[1] + R[5] * M[2] + t[1];
        double Z = R[6] * M[0] + R[7] * M[1] + R[8] * M[2] + t[2];
        double z = Z != 0 ? 1 / Z : 1;
        double x = X * z, y = Y * z;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cd = 1 + k1 * r2 + k2 * r4 + k3 * r6;
        double xd = x * cd + p1 * a1 + p2 * a2;
        double yd = y * cd + p1 * a3 + p2 * a1;
        if (proj) { proj[2 * p] = fx * xd + cx; proj[2 * p + 1] = fy * yd + cy; }
        if (dpdc) {
            double *d = dpdc + 4 * p;
            d[0] = 1; d[1] = 0; d[2] = 0; d[3] = 1;
        }
        if (dpdf) {
            double *d = dpdf + 4 * p;
            if (aspect > 0) { d[0] = 0; d[1] = xd * aspect; d[2] = 0; d[3] = yd; }
            else { d[0] = xd; d[1] = 0; d[2] = 0; d[3] = yd; }
        }
        if (dpdk) {
            double *d = dpdk + 10 * p;
            d[0] = fx * x * r2; d[1] = fx * x * r4; d[2] = fx * a1; d[3] = fx * a2; d[4] = fx * x * r6;
            d[5] = fy * y * r2; d[6] = fy * y * r4; d[7] = fy * a3; d[8] = fy * a1; d[9] = fy * y * r6;
        }
        if (dpdt || dpdr) {
            for (int part = 0; part < 2; part++) {
                double *out = part == 0 ? dpdt : dpdr;
                if (!out) continue;
                for (int j = 0; j < 3; j++) {
                    double dX, dY, dZ;
                    if (part == 0) { dX = (j == 0); dY = (j == 1); dZ = (j == 2); }
                    else {
                        const double *D = dRdr + 9 * j;
                        dX = D[0] * M[0] + D[1] * M[1] + D[2] * M[2];
                        dY = D[3] * M[0] + D[4] * M[1] + D[5] * M[2];
                        dZ = D[6] * M[0] + D[7] * M[1] + D[8] * M[2];
                    }
                    double dx = z * (dX - x * dZ), dy = z * (dY - y * dZ);
                    double dr2 = 2 * x * dx + 2 * y * dy;
                    double dcd = (k1 + 2 * k2 * r2 + 3 * k3 * r4) * dr2;
                    double da1 = 2 * (x * dy + y * dx);
                    double dxd = cd * dx + x * dcd + p1 * da1 + p2 * (dr2 + 4 * x * dx);
                    double dyd = cd * dy + y * dcd + p1 * (dr2 + 4 * y * dy) + p2 * da1;
                    out[6 * p + j] = fx * dxd;
                    out[6 * p + 3 + j] = fy * dyd;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ homography */

static void homog_residual(int P, const float *M, const float *m, const double h[8], double *err,
                           double *J) {
    /* [OCV-K HomographyRefineCallback::compute]; h[8] == 1 */
    for (int i = 0; i < P; i++) {
        double Mx = M[2 * i], My = M[2 * i + 1];
        double ww = h[6] * Mx + h[7] * My + 1.0;
        ww = fabs(ww) > DBL_EPSILON ? 1.0 / ww : 0;
        double xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
        double yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
        err[2 * i] = xi - m[2 * i];
        err[2 * i + 1] = yi - m[2 * i + 1];
        if (J) {
            double *a = J + 16 * i;
            a[0] = Mx * ww; a[1] = My * ww; a[2] = ww; a[3] = a[4] = a[5] = 0;
            a[6] = -Mx * ww * xi; a[7] = -My * ww * xi;
            a[8] = a[9] = a[10] = 0; a[11] = Mx * ww; a[12] = My * ww; a[13] = ww;
            a[14] = -Mx * ww * yi; a[15] = -My * ww * yi;
        }
    }
}

/* [OCV-K LMSolverImpl::run], eps = FLT_EPSILON, used with maxIters = 10 on 8 parameters */
static void lmsolver_homography(int P, const float *M, const float *m, double x[8], int max_iters) {
    enum { L = 8 };
    const double eps = FLT_EPSILON;
    double *r = (double *)malloc(sizeof(double) * 2 * P), *rd = (double *)malloc(sizeof(double) * 2 * P);
    double *J = (double *)malloc(sizeof(double) * 16 * P);
    double A[L * L], Ap[L * L], v[L], d[L], xd[L], D[L];
    homog_residual(P, M, m, x, r, J);
    double S = 0;
    for (int i = 0; i < 2 * P; i++) S += r[i] * r[i];
#define ACCUM_AV()                                                               \
    do {                                                                         \
        memset(A, 0, sizeof A); memset(v, 0, sizeof v);                          \
        for (int i = 0; i < 2 * P; i++)                                          \
            for (int a = 0; a < L; a++) {                                        \
                v[a] += J[i * L + a] * r[i];                                     \
                for (int b = 0; b < L; b++) A[a * L + b] += J[i * L + a] * J[i * L + b]; \
            }                                                                    \
    } while (0)
    ACCUM_AV();
    for (int a = 0; a < L; a++) D[a] = A[a * L + a];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1, lc = 0.75;
    for (int iter = 0;;) {
        memcpy(Ap, A, sizeof A);
        for (int a = 0; a < L; a++) Ap[a * L + a] += lambda * D[a];
        sym_solve(L, Ap, v, d);
        for (int a = 0; a < L; a++) xd[a] = x[a] - d[a];
        homog_residual(P, M, m, xd, rd, NULL);
        double Sd = 0;
        for (int i = 0; i < 2 * P; i++) Sd += rd[i] * rd[i];
        double dS = 0, dv = 0; /* dS = d . (2 v - A d) */
        for (int a = 0; a < L; a++) {
            double Ad = 0;
            for (int b = 0; b < L; b++) Ad += A[a * L + b] * d[b];
            dS += d[a] * (2 * v[a] - Ad);
            dv += d[a] * v[a];
        }
        double Rr = (S - Sd) / (fabs(dS) > DBL_EPSILON ? dS : 1);
        if (Rr > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0;
        } else if (Rr < Rlo) {
            double nu = (Sd - S) / (fabs(dv) > DBL_EPSILON ? dv : 1) + 2;
            nu = nu < 2 ? 2 : (nu > 10 ? 10 : nu);
            if (lambda == 0) {
                /* lambda = lc = 1 / max |diag(A^-1)| */
                double maxval = DBL_EPSILON, col[L], e[L];
                for (int a = 0; a < L; a++) {
                    for (int b = 0; b < L; b++) e[b] = (a == b);
                    sym_solve(L, A, e, col);
                    if (fabs(col[a]) > maxval) maxval = fabs(col[a]);
                }
                lambda = lc = 1.0 / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        if (Sd < S) {
            S = Sd;
            memcpy(x, xd, sizeof xd);
            homog_residual(P, M, m, x, r, J);
            ACCUM_AV();
        }
        iter++;
        double dmax = 0, rmax = 0;
        for (int a = 0; a < L; a++) dmax = fmax(dmax, fabs(d[a]));
        for (int i = 0; i < 2 * P; i++) rmax = fmax(rmax, fabs(r[i]));
        if (!(iter < max_iters && dmax >= eps && rmax >= eps)) break;
    }
#undef ACCUM_AV
    free(r); free(rd); free(J);
}

/* [OCV-K findHomography(method 0) -> HomographyEstimatorCallback::runKernel + LM polish] */
int orc_find_homography(int P, const double *src_xy, const double *dst_xy, double H[9], int refine) {
    float *M = (float *)malloc(sizeof(float) * 2 * P), *m = (float *)malloc(sizeof(float) * 2 * P);
    for (int i = 0; i < 2 * P; i++) { M[i] = (float)src_xy[i]; m[i] = (float)dst_xy[i]; }
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < P; i++) {
        cm[0] += m[2 * i]; cm[1] += m[2 * i + 1];
        cM[0] += M[2 * i]; cM[1] += M[2 * i + 1];
    }
    cm[0] /= P; cm[1] /= P; cM[0] /= P; cM[1] /= P;
    for (int i = 0; i < P; i++) {
        sm[0] += fabs(m[2 * i] - cm[0]); sm[1] += fabs(m[2 * i + 1] - cm[1]);
        sM[0] += fabs(M[2 * i] - cM[0]); sM[1] += fabs(M[2 * i + 1] - cM[1]);
    }
    if (fabs(sm[0]) < DBL_EPSILON || fabs(sm[1]) < DBL_EPSILON || fabs(sM[0]) < DBL_EPSILON ||
        fabs(sM[1]) < DBL_EPSILON) {
        free(M); free(m);
        return -1;
    }
    sm[0] = P / sm[0]; sm[1] = P / sm[1]; sM[0] = P / sM[0]; sM[1] = P / sM[1];
    double invHnorm[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
    double Hnorm2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double LtL[81] = {0};
    for (int i = 0; i < P; i++) {
        double x = (m[2 * i] - cm[0]) * sm[0], y = (m[2 * i + 1] - cm[1]) * sm[1];
        double X = (M[2 * i] - cM[0]) * sM[0], Y = (M[2 * i + 1] - cM[1]) * sM[1];
        double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
        double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int j = 0; j < 9; j++)
            for (int k = 0; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
    double W[9], V[81], T[9];
    jacobi_eig(9, LtL, W, V);
    mat3_mul(invHnorm, V + 72, T); /* smallest-eigenvalue eigenvector = last row */
    mat3_mul(T, Hnorm2, H);
    double s = 1.0 / H[8];
    for (int k = 0; k < 9; k++) H[k] *= s;
    if (refine && P > 4) {
        lmsolver_homography(P, M, m, H, 10);
        H[8] = 1;
    }
    free(M); free(m);
    return 0;
}

/* [OCV-K cvInitIntrinsicParams2D]; SURVEY.md row A4 */
int orc_init_intrinsics(int N, int P, const double *obj, const double *img, int width, int height,
                        double aspect, double K4[4], double *allH) {
    double a2 = !width ? 0.5 : (width - 1) * 0.5, a5 = !height ? 0.5 : (height - 1) * 0.5;
    double *oxy = (double *)malloc(sizeof(double) * 2 * P);
    for (int p = 0; p < P; p++) { oxy[2 * p] = obj[3 * p]; oxy[2 * p + 1] = obj[3 * p + 1]; }
    double AtA[4] = {0, 0, 0, 0}, Atb[2] = {0, 0};
    int rc = 0;
    for (int i = 0; i < N && rc == 0; i++) {
        double H[9];
        if (orc_find_homography(P, oxy, img + (size_t)2 * P * i, H, 1) != 0) { rc = -3; break; }
        if (allH) memcpy(allH + 9 * i, H, sizeof H);
        H[0] -= H[6] * a2; H[1] -= H[7] * a2; H[2] -= H[8] * a2;
        H[3] -= H[6] * a5; H[4] -= H[7] * a5; H[5] -= H[8] * a5;
        double h[3], v[3], d1[3], d2[3], n[4] = {0, 0, 0, 0};
        for (int j = 0; j < 3; j++) {
            double t0 = H[j * 3], t1 = H[j * 3 + 1];
            h[j] = t0; v[j] = t1; d1[j] = (t0 + t1) * 0.5; d2[j] = (t0 - t1) * 0.5;
            n[0] += t0 * t0; n[1] += t1 * t1; n[2] += d1[j] * d1[j]; n[3] += d2[j] * d2[j];
        }
        for (int j = 0; j < 4; j++) n[j] = 1. / sqrt(n[j]);
        for (int j = 0; j < 3; j++) { h[j] *= n[0]; v[j] *= n[1]; d1[j] *= n[2]; d2[j] *= n[3]; }
        double Ap[4] = {h[0] * v[0], h[1] * v[1], d1[0] * d2[0], d1[1] * d2[1]};
        double bp[2] = {-h[2] * v[2], -d1[2] * d2[2]};
        for (int row = 0; row < 2; row++) {
            AtA[0] += Ap[2 * row] * Ap[2 * row]; AtA[1] += Ap[2 * row] * Ap[2 * row + 1];
            AtA[3] += Ap[2 * row + 1] * Ap[2 * row + 1];
            Atb[0] += Ap[2 * row] * bp[row]; Atb[1] += Ap[2 * row + 1] * bp[row];
        }
    }
    free(oxy);
    if (rc) return rc;
    AtA[2] = AtA[1];
    double f[2];
    sym_solve(2, AtA, Atb, f); /* cvSolve(A, b, f, CV_NORMAL + CV_SVD) */
    double fx = sqrt(fabs(1. / f[0])), fy = sqrt(fabs(1. / f[1]));
    if (aspect != 0) {
        double tf = (fx + fy) / (aspect + 1.);
        fx = aspect * tf; fy = tf;
    }
    K4[0] = fx; K4[1] = fy; K4[2] = a2; K4[3] = a5;
    return 0;
}

/* ------------------------------------------------------------------ CvLevMarq state machine */

enum { ST_DONE = 0, ST_STARTED = 1, ST_CALC_J = 2, ST_CHECK_ERR = 3 };

typedef struct {
    int n, max_iter, lambdaLg10, iters, state, solver;
    double eps, errNorm, prevErrNorm;
    double *param, *prev, *JtJ, *JtErr; /* dense: n x n */
    unsigned char *mask;
    /* Schur storage (solver == 1): A 18x18 in JtJ[0..324), blocks per image */
    int N;
    double *Bi, *Ci; /* [N][NSLOT*6], [N][36] */
} lm_t;

static void lm_step_dense(lm_t *s) {
    /* [OCV-K CvLevMarq::step]: masked sub-system, upper -> lower symmetrise,
     * diag *= 1 + lambda, solve, param = prev - delta */
    double lambda = exp(s->lambdaLg10 * log(10.));
    int n = s->n, nz = 0;
    int *idx = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++)
        if (s->mask[i]) idx[nz++] = i;
    double *A = (double *)malloc(sizeof(double) * nz * nz), *b = (double *)malloc(sizeof(double) * nz);
    double *x = (double *)malloc(sizeof(double) * nz);
    for (int i = 0; i < nz; i++) {
        b[i] = s->JtErr[idx[i]];
        for (int j = i; j < nz; j++) A[i * nz + j] = A[j * nz + i] = s->JtJ[(size_t)idx[i] * n + idx[j]];
        A[i * nz + i] *= 1. + lambda;
    }
    sym_solve(nz, A, b, x);
    for (int i = 0; i < n; i++) s->param[i] = s->prev[i];
    for (int i = 0; i < nz; i++) s->param[idx[i]] -= x[i];
    free(idx); free(A); free(b); free(x);
}

/* Same damped system solved through the Schur complement over the 6x6 extrinsic
 * blocks: S = A_l - sum B C_l^-1 B^T, etc. */
static void lm_step_schur(lm_t *s) {
    double lambda = exp(s->lambdaLg10 * log(10.));
    int nz = 0, idx[NSLOT];
    for (int i = 0; i < NSLOT; i++)
        if (s->mask[i]) idx[nz++] = i;
    int N = s->N;
    double S[NSLOT * NSLOT], g[NSLOT], dint[NSLOT];
    for (int a = 0; a < nz; a++) {
        g[a] = s->JtErr[idx[a]];
        for (int b = 0; b < nz; b++) S[a * nz + b] = s->JtJ[idx[a] * NSLOT + idx[b]];
        S[a * nz + a] *= 1. + lambda;
    }
    double *Y = (double *)malloc(sizeof(double) * (size_t)N * (NSLOT * 6 + 6 + 36));
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        double *Li = Y + (size_t)i * (NSLOT * 6 + 6 + 36), *Yi = Li + 36, *yh = Yi + NSLOT * 6;
        memcpy(Li, s->Ci + 36 * i, 36 * sizeof(double));
        for (int a = 0; a < 6; a++) Li[a * 6 + a] *= 1. + lambda;
        if (chol_factor(6, Li) != 0) { /* degenerate view: drop its coupling */
            memset(Yi, 0, sizeof(double) * (NSLOT * 6 + 6));
            Li[0] = -1;
            continue;
        }
        for (int a = 0; a < nz; a++) { /* Yi[a] = C^-1 B_a^T  (6-vector) */
            double *y = Yi + 6 * a;
            memcpy(y, s->Bi + (size_t)i * NSLOT * 6 + idx[a] * 6, 6 * sizeof(double));
            chol_solve(6, Li, y);
        }
        memcpy(yh, s->JtErr + NSLOT + 6 * i, 6 * sizeof(double));
        chol_solve(6, Li, yh);
    }
    for (int i = 0; i < N; i++) { /* fixed-order accumulation */
        const double *Yi = Y + (size_t)i * (NSLOT * 6 + 6 + 36) + 36, *yh = Yi + NSLOT * 6;
        const double *B = s->Bi + (size_t)i * NSLOT * 6, *h = s->JtErr + NSLOT + 6 * i;
        (void)h;
        for (int a = 0; a < nz; a++) {
            const double *Ba = B + idx[a] * 6;
            double acc = 0;
            for (int k = 0; k < 6; k++) acc += Ba[k] * yh[k];
            g[a] -= acc;
            for (int b = 0; b < nz; b++) {
                double sb = 0;
                for (int k = 0; k < 6; k++) sb += Ba[k] * Yi[6 * b + k];
                S[a * nz + b] -= sb;
            }
        }
    }
    for (int a = 0; a < nz; a++) /* symmetrise against rounding */
        for (int b = a + 1; b < nz; b++) S[a * nz + b] = S[b * nz + a] = 0.5 * (S[a * nz + b] + S[b * nz + a]);
    sym_solve(nz, S, g, dint);
    for (int i = 0; i < s->n; i++) s->param[i] = s->prev[i];
    for (int a = 0; a < nz; a++) s->param[idx[a]] -= dint[a];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        const double *Li = Y + (size_t)i * (NSLOT * 6 + 6 + 36);
        if (Li[0] < 0) continue;
        const double *B = s->Bi + (size_t)i * NSLOT * 6;
        double rhs[6];
        for (int k = 0; k < 6; k++) {
            double acc = s->JtErr[NSLOT + 6 * i + k];
            for (int a = 0; a < nz; a++) acc -= B[idx[a] * 6 + k] * dint[a];
            rhs[k] = acc;
        }
        chol_solve(6, Li, rhs);
        for (int k = 0; k < 6; k++)
            if (s->mask[NSLOT + 6 * i + k]) s->param[NSLOT + 6 * i + k] -= rhs[k];
    }
    free(Y);
}

static void lm_step(lm_t *s) {
    if (s->solver == 1) lm_step_schur(s);
    else lm_step_dense(s);
}

static void lm_zero(lm_t *s) {
    if (s->solver == 1) {
        memset(s->JtJ, 0, sizeof(double) * NSLOT * NSLOT);
        memset(s->Bi, 0, sizeof(double) * (size_t)s->N * NSLOT * 6);
        memset(s->Ci, 0, sizeof(double) * (size_t)s->N * 36);
    } else
        memset(s->JtJ, 0, sizeof(double) * (size_t)s->n * s->n);
    memset(s->JtErr, 0, sizeof(double) * s->n);
}

static double rel_l2(int n, const double *a, const double *b) {
    double num = 0, den = 0;
    for (int i = 0; i < n; i++) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
    return sqrt(num) / sqrt(den);
}

/* [OCV-K CvLevMarq::updateAlt]; SURVEY.md A.6.  decision: out-param for tracing
 * (0 none, 1 accepted, 2 rejected). */
static int lm_update_alt(lm_t *s, int *decision, double *rel) {
    *decision = 0;
    if (s->state == ST_DONE) return 0;
    if (s->state == ST_STARTED) {
        lm_zero(s);
        s->errNorm = 0;
        s->state = ST_CALC_J;
        return 1;
    }
    if (s->state == ST_CALC_J) {
        memcpy(s->prev, s->param, sizeof(double) * s->n);
        lm_step(s);
        s->prevErrNorm = s->errNorm;
        s->errNorm = 0;
        s->state = ST_CHECK_ERR;
        return 1;
    }
    if (s->errNorm > s->prevErrNorm) {
        if (++s->lambdaLg10 <= 16) {
            lm_step(s);
            s->errNorm = 0;
            s->state = ST_CHECK_ERR;
            *decision = 2;
            return 1;
        }
    }
    *decision = 1;
    s->lambdaLg10 = s->lambdaLg10 - 1 > -16 ? s->lambdaLg10 - 1 : -16;
    *rel = rel_l2(s->n, s->param, s->prev);
    if (++s->iters >= s->max_iter || *rel < s->eps) {
        s->state = ST_DONE;
        return 0;
    }
    s->prevErrNorm = s->errNorm;
    lm_zero(s);
    s->state = ST_CALC_J;
    return 1;
}

/* ------------------------------------------------------------------ extrinsic initialisation */

static void undistort_point(double u, double v, const double in[ORC_NINTR], double *xo, double *yo) {
    /* [OCV-K cvUndistortPoints, 5 fixed-point iterations]; identity map when dist == 0 */
    double x = (u - in[2]) / in[0], y = (v - in[3]) / in[1];
    const double k1 = in[4], k2 = in[5], p1 = in[6], p2 = in[7], k3 = in[8];
    if (k1 != 0 || k2 != 0 || p1 != 0 || p2 != 0 || k3 != 0) {
        double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            double r2 = x * x + y * y;
            double icd = 1. / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            if (icd < 0) { x = x0; y = y0; break; }
            double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            double dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - dx) * icd; y = (y0 - dy) * icd;
        }
    }
    *xo = x; *yo = y;
}

/* [OCV-K cvFindExtrinsicCameraParams2, planar branch]; SURVEY.md row A5 */
int orc_find_extrinsics(int P, const double *obj, const double *img_i, const double in[ORC_NINTR],
                        double r[3], double t[3]) {
    double *mn = (double *)malloc(sizeof(double) * 2 * P), *Mxy = (double *)malloc(sizeof(double) * 2 * P);
    for (int p = 0; p < P; p++) undistort_point(img_i[2 * p], img_i[2 * p + 1], in, &mn[2 * p], &mn[2 * p + 1]);
    double Mc[3] = {0, 0, 0}, MM[9] = {0}, W[3], V[9];
    for (int p = 0; p < P; p++)
        for (int c = 0; c < 3; c++) Mc[c] += obj[3 * p + c];
    for (int c = 0; c < 3; c++) Mc[c] /= P;
    for (int p = 0; p < P; p++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) MM[a * 3 + b] += (obj[3 * p + a] - Mc[a]) * (obj[3 * p + b] - Mc[b]);
    jacobi_eig(3, MM, W, V);
    if (!(W[2] / W[1] < 1e-3)) { free(mn); free(Mxy); return -4; } /* non-planar rig: out of scope */
    double Rt[9], Tt[3];
    memcpy(Rt, V, sizeof Rt);
    if (V[2] * V[2] + V[5] * V[5] < 1e-10) {
        for (int k = 0; k < 9; k++) Rt[k] = (k % 4 == 0);
    }
    double det = Rt[0] * (Rt[4] * Rt[8] - Rt[5] * Rt[7]) - Rt[1] * (Rt[3] * Rt[8] - Rt[5] * Rt[6]) +
                 Rt[2] * (Rt[3] * Rt[7] - Rt[4] * Rt[6]);
    if (det < 0)
        for (int k = 0; k < 9; k++) Rt[k] = -Rt[k];
    for (int a = 0; a < 3; a++) Tt[a] = -(Rt[a * 3] * Mc[0] + Rt[a * 3 + 1] * Mc[1] + Rt[a * 3 + 2] * Mc[2]);
    for (int p = 0; p < P; p++) {
        const double *s = obj + 3 * p;
        Mxy[2 * p] = Rt[0] * s[0] + Rt[1] * s[1] + Rt[2] * s[2] + Tt[0];
        Mxy[2 * p + 1] = Rt[3] * s[0] + Rt[4] * s[1] + Rt[5] * s[2] + Tt[1];
    }
    double H[9], R[9], tt[3];
    int ok = orc_find_homography(P, Mxy, mn, H, 1) == 0;
    for (int k = 0; k < 9 && ok; k++)
        if (!isfinite(H[k])) ok = 0;
    if (ok) {
        double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]}, h3[3] = {H[2], H[5], H[8]};
        double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]);
        double n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
        for (int a = 0; a < 3; a++) {
            h1[a] /= fmax(n1, DBL_EPSILON);
            h2[a] /= fmax(n2, DBL_EPSILON);
            tt[a] = h3[a] * 2. / fmax(n1 + n2, DBL_EPSILON);
        }
        h3[0] = h1[1] * h2[2] - h1[2] * h2[1];
        h3[1] = h1[2] * h2[0] - h1[0] * h2[2];
        h3[2] = h1[0] * h2[1] - h1[1] * h2[0];
        double Hm[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
        double rr[3];
        orc_rodrigues_inv(Hm, rr);
        orc_rodrigues(rr, R, NULL);
        for (int a = 0; a < 3; a++) tt[a] += R[a * 3] * Tt[0] + R[a * 3 + 1] * Tt[1] + R[a * 3 + 2] * Tt[2];
        mat3_mul(R, Rt, R);
    } else {
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0);
        tt[0] = tt[1] = tt[2] = 0;
    }
    orc_rodrigues_inv(R, r);
    memcpy(t, tt, sizeof tt);
    free(mn); free(Mxy);

    /* refine: CvLevMarq(6, 2P, (COUNT+EPS, 20, FLT_EPSILON)), full-J `update` flavour:
     * same state machine as updateAlt with errNorm = ||err||_2 (not squared) */
    lm_t s;
    memset(&s, 0, sizeof s);
    double param[6], prev[6], JtJ[36], JtErr[6];
    unsigned char mask[6] = {1, 1, 1, 1, 1, 1};
    s.n = 6; s.max_iter = 20; s.eps = FLT_EPSILON; s.lambdaLg10 = -3; s.state = ST_STARTED;
    s.param = param; s.prev = prev; s.JtJ = JtJ; s.JtErr = JtErr; s.mask = mask; s.solver = 0;
    memcpy(param, r, 3 * sizeof(double));
    memcpy(param + 3, t, 3 * sizeof(double));
    double *proj = (double *)malloc(sizeof(double) * 2 * P), *dr = (double *)malloc(sizeof(double) * 6 * P);
    double *dt = (double *)malloc(sizeof(double) * 6 * P);
    for (;;) {
        int dec; double rel;
        int proceed = lm_update_alt(&s, &dec, &rel);
        if (!proceed) break;
        int calcJ = s.state == ST_CALC_J;
        orc_project_points(P, obj, param, param + 3, in, 0, proj, calcJ ? dr : NULL, calcJ ? dt : NULL, NULL,
                           NULL, NULL);
        double e2 = 0;
        for (int p = 0; p < P; p++) {
            double e[2] = {proj[2 * p] - img_i[2 * p], proj[2 * p + 1] - img_i[2 * p + 1]};
            e2 += e[0] * e[0] + e[1] * e[1];
            if (calcJ)
                for (int row = 0; row < 2; row++) {
                    double Jrow[6];
                    for (int j = 0; j < 3; j++) { Jrow[j] = dr[6 * p + 3 * row + j]; Jrow[3 + j] = dt[6 * p + 3 * row + j]; }
                    for (int a = 0; a < 6; a++) {
                        JtErr[a] += Jrow[a] * e[row];
                        for (int b = 0; b < 6; b++) JtJ[a * 6 + b] += Jrow[a] * Jrow[b];
                    }
                }
        }
        s.errNorm = sqrt(e2);
    }
    memcpy(r, param, 3 * sizeof(double));
    memcpy(t, param + 3, 3 * sizeof(double));
    free(proj); free(dr); free(dt);
    return 0;
}

/* ------------------------------------------------------------------ per-image Gram (diagnostic) */

void orc_image_gram(int P, const double *obj, const double *img_i, const double r[3], const double t[3],
                    const double in[ORC_NINTR], double aspect, double G[256]) {
    double *proj = (double *)malloc(sizeof(double) * 2 * P), *dr = (double *)malloc(sizeof(double) * 6 * P);
    double *dt = (double *)malloc(sizeof(double) * 6 * P), *df = (double *)malloc(sizeof(double) * 4 * P);
    double *dc = (double *)malloc(sizeof(double) * 4 * P), *dk = (double *)malloc(sizeof(double) * 10 * P);
    orc_project_points(P, obj, r, t, in, aspect, proj, dr, dt, df, dc, dk);
    memset(G, 0, 256 * sizeof(double));
    for (int p = 0; p < P; p++)
        for (int row = 0; row < 2; row++) {
            double m[16];
            for (int j = 0; j < 3; j++) { m[j] = dr[6 * p + 3 * row + j]; m[3 + j] = dt[6 * p + 3 * row + j]; }
            m[6] = df[4 * p + 2 * row]; m[7] = df[4 * p + 2 * row + 1];
            m[8] = dc[4 * p + 2 * row]; m[9] = dc[4 * p + 2 * row + 1];
            for (int j = 0; j < 5; j++) m[10 + j] = dk[10 * p + 5 * row + j];
            m[15] = proj[2 * p + row] - img_i[2 * p + row];
            for (int a = 0; a < 16; a++)
                for (int b = 0; b < 16; b++) G[a * 16 + b] += m[a] * m[b];
        }
    free(proj); free(dr); free(dt); free(df); free(dc); free(dk);
}

/* ------------------------------------------------------------------ calibrateCamera */

int orc_calibrate(int N, int P, const double *obj, const double *img, int width, int height,
                  double aspect_in, unsigned flags, int max_iter, double eps, int solver,
                  const double *init_intr, const double *init_rt, double intr_out[ORC_NINTR],
                  double *rvecs, double *tvecs, orc_stats_t *stats, orc_trace_t *trace, int max_trace,
                  int nthreads) {
    if (N < 1 || P < 4 || !obj || !img) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
    double t0 = now_s();
    const int fix_aspect = (flags & ORC_FIX_ASPECT_RATIO) != 0;
    double aspect = 0;
    double intr[ORC_NINTR] = {0};
    /* 1. initial intrinsics  [OCV-K cvCalibrateCamera2Internal step 1] */
    if (flags & ORC_USE_INTRINSIC_GUESS) {
        if (!init_intr) return -1;
        memcpy(intr, init_intr, sizeof intr);
        if (fix_aspect) aspect = intr[0] / intr[1];
    } else {
        for (int p = 0; p < P; p++)
            if (fabs(obj[3 * p + 2] - obj[2]) > 1e-5 || fabs(obj[2]) > 1e-5) return -2; /* non-planar / z != 0 */
        if (fix_aspect) aspect = aspect_in;
        if (init_intr) memcpy(intr, init_intr, sizeof intr);
        else {
            double K4[4];
            int rc = orc_init_intrinsics(N, P, obj, img, width, height, aspect, K4, NULL);
            if (rc) return rc;
            memcpy(intr, K4, sizeof K4);
        }
    }
    if (fix_aspect && (aspect < 0.01 || aspect > 100)) return -1;

    const int n = NSLOT + 6 * N;
    lm_t s;
    memset(&s, 0, sizeof s);
    s.n = n; s.N = N; s.max_iter = max_iter; s.eps = eps; s.lambdaLg10 = -3; s.state = ST_STARTED;
    s.solver = solver;
    s.param = (double *)calloc(n, sizeof(double));
    s.prev = (double *)calloc(n, sizeof(double));
    s.JtErr = (double *)calloc(n, sizeof(double));
    s.mask = (unsigned char *)malloc(n);
    memset(s.mask, 1, n);
    if (solver == 1) {
        s.JtJ = (double *)calloc(NSLOT * NSLOT, sizeof(double));
        s.Bi = (double *)calloc((size_t)N * NSLOT * 6, sizeof(double));
        s.Ci = (double *)calloc((size_t)N * 36, sizeof(double));
    } else
        s.JtJ = (double *)calloc((size_t)n * n, sizeof(double));
    double *param = s.param;
    memcpy(param, intr, sizeof intr);
    /* SURVEY.md A.1 mask */
    unsigned char *mask = s.mask;
    for (int i = ORC_NINTR; i < NSLOT; i++) mask[i] = 0;
    if (fix_aspect) mask[0] = 0;
    if (flags & ORC_FIX_FOCAL_LENGTH) mask[0] = mask[1] = 0;
    if (flags & ORC_FIX_PRINCIPAL_POINT) mask[2] = mask[3] = 0;
    if (flags & ORC_ZERO_TANGENT_DIST) { param[6] = param[7] = 0; mask[6] = mask[7] = 0; }
    if (flags & ORC_FIX_TANGENT_DIST) mask[6] = mask[7] = 0;
    mask[4] = !(flags & ORC_FIX_K1);
    mask[5] = !(flags & ORC_FIX_K2);
    mask[8] = !(flags & ORC_FIX_K3);

    /* 2. initial extrinsics */
    int rc = 0;
    if (init_rt) memcpy(param + NSLOT, init_rt, sizeof(double) * 6 * N);
    else {
        double in0[ORC_NINTR];
        memcpy(in0, param, sizeof in0);
#pragma omp parallel for schedule(dynamic, 8)
        for (int i = 0; i < N; i++) {
            int e = orc_find_extrinsics(P, obj, img + (size_t)2 * P * i, in0, param + NSLOT + 6 * i,
                                        param + NSLOT + 6 * i + 3);
            if (e) {
#pragma omp critical
                rc = e;
            }
        }
    }
    if (stats) {
        memcpy(stats->init_intr, param, sizeof intr);
        stats->t_init_s = now_s() - t0;
    }
    if (rc) goto done;

    /* 3. global LM  [OCV-K cvCalibrateCamera2Internal step 3]; SURVEY.md A.6 */
    {
        double t1 = now_s();
        double reprojErr = 0;
        int jsw = 0, esw = 0, ntr = 0, trial = 0;
        double *perA = NULL, *perE = (double *)malloc(sizeof(double) * N);
        double *perg = NULL;
        if (1) {
            perA = (double *)malloc(sizeof(double) * (size_t)N * NSLOT * NSLOT);
            perg = (double *)malloc(sizeof(double) * (size_t)N * NSLOT);
        }
        for (;;) {
            int dec = 0; double rel = -1;
            int proceed = lm_update_alt(&s, &dec, &rel);
            if (fix_aspect) { s.param[0] = s.param[1] * aspect; s.prev[0] = s.prev[1] * aspect; }
            if (dec && trace && ntr < max_trace) {
                orc_trace_t *tr = &trace[ntr++];
                tr->trial = trial; tr->accepted = dec == 1; tr->lambda_lg10 = s.lambdaLg10;
                tr->iters = s.iters; tr->err = reprojErr; tr->rel_step = dec == 1 ? rel : -1;
            }
            if (!proceed) break;
            int calcJ = s.state == ST_CALC_J;
            trial++;
            if (calcJ) jsw++; else esw++;
#pragma omp parallel for schedule(static)
            for (int i = 0; i < N; i++) {
                const double *ri = s.param + NSLOT + 6 * i, *ti = ri + 3;
                const double *mi = img + (size_t)2 * P * i;
                if (!calcJ) {
                    double *proj = (double *)malloc(sizeof(double) * 2 * P);
                    orc_project_points(P, obj, ri, ti, s.param, fix_aspect ? aspect : 0, proj, NULL, NULL, NULL,
                                       NULL, NULL);
                    double e2 = 0;
                    for (int k = 0; k < 2 * P; k++) e2 += (proj[k] - mi[k]) * (proj[k] - mi[k]);
                    perE[i] = e2;
                    free(proj);
                    continue;
                }
                double G[256];
                orc_image_gram(P, obj, mi, ri, ti, s.param, fix_aspect ? aspect : 0, G);
                perE[i] = G[255];
                /* JtJ blocks.  FIX_FOCAL_LENGTH / FIX_PRINCIPAL_POINT drop
                 * dpdf / dpdc upstream; the mask removes those rows/cols anyway. */
                double *Ai = perA + (size_t)i * NSLOT * NSLOT, *gi = perg + (size_t)i * NSLOT;
                memset(Ai, 0, sizeof(double) * NSLOT * NSLOT);
                memset(gi, 0, sizeof(double) * NSLOT);
                for (int a = 0; a < ORC_NINTR; a++) {
                    gi[a] = G[(6 + a) * 16 + 15];
                    for (int b = 0; b < ORC_NINTR; b++) Ai[a * NSLOT + b] = G[(6 + a) * 16 + 6 + b];
                }
                if (solver == 1) {
                    for (int a = 0; a < ORC_NINTR; a++)
                        for (int k = 0; k < 6; k++) s.Bi[(size_t)i * NSLOT * 6 + a * 6 + k] = G[(6 + a) * 16 + k];
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) s.Ci[36 * i + a * 6 + b] = G[a * 16 + b];
                } else {
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++)
                            s.JtJ[(size_t)(NSLOT + 6 * i + a) * n + NSLOT + 6 * i + b] = G[a * 16 + b];
                    for (int a = 0; a < ORC_NINTR; a++)
                        for (int k = 0; k < 6; k++) s.JtJ[(size_t)a * n + NSLOT + 6 * i + k] = G[(6 + a) * 16 + k];
                }
                for (int k = 0; k < 6; k++) s.JtErr[NSLOT + 6 * i + k] = G[k * 16 + 15];
            }
            reprojErr = 0;
            for (int i = 0; i < N; i++) reprojErr += perE[i];
            if (calcJ) {
                const int ld = solver == 1 ? NSLOT : n;
                for (int i = 0; i < N; i++) { /* fixed-order accumulation of the intrinsic corner */
                    const double *Ai = perA + (size_t)i * NSLOT * NSLOT, *gi = perg + (size_t)i * NSLOT;
                    for (int a = 0; a < ORC_NINTR; a++) {
                        s.JtErr[a] += gi[a];
                        for (int b = 0; b < ORC_NINTR; b++) s.JtJ[(size_t)a * ld + b] += Ai[a * NSLOT + b];
                    }
                }
            }
            s.errNorm = reprojErr;
        }
        free(perA); free(perg); free(perE);
        if (stats) {
            stats->rms = sqrt(reprojErr / ((double)N * P));
            stats->iters = s.iters; stats->j_sweeps = jsw; stats->e_sweeps = esw;
            stats->lambda_lg10 = s.lambdaLg10; stats->n_trace = ntr;
            stats->t_lm_s = now_s() - t1;
        }
    }
    memcpy(intr_out, s.param, sizeof intr);
    for (int i = 0; i < N; i++) {
        memcpy(rvecs + 3 * i, s.param + NSLOT + 6 * i, 3 * sizeof(double));
        memcpy(tvecs + 3 * i, s.param + NSLOT + 6 * i + 3, 3 * sizeof(double));
    }
done:
    free(s.param); free(s.prev); free(s.JtErr); free(s.mask); free(s.JtJ); free(s.Bi); free(s.Ci);
    return rc;
}

/* ------------------------------------------------------------------ Julia-side maps */

/* src/meta.jl:27-33,39-44 + src/buildcalibrations.jl:1-6:
 * intrinsic o distort o PerspectiveMap o extrinsic o scale */
void orc_map_forward(int M, const double *xyz, const double f[2], const double c[2], double k,
                     const double r[3], const double t[3], double cs, double *rc) {
    double R[9];
    orc_rodrigues(r, R, NULL);
    for (int i = 0; i < M; i++) {
        double s[3] = {xyz[3 * i] / cs, xyz[3 * i + 1] / cs, xyz[3 * i + 2] / cs};
        double X = R[0] * s[0] + R[1] * s[1] + R[2] * s[2] + t[0];
        double Y = R[3] * s[0] + R[4] * s[1] + R[5] * s[2] + t[1];
        double Z = R[6] * s[0] + R[7] * s[1] + R[8] * s[2] + t[2];
        double v[2] = {X / Z, Y / Z};
        if (k != 0) {
            double radial = 1 + k * (v[0] * v[0] + v[1] * v[1]);
            v[0] *= radial; v[1] *= radial;
        }
        rc[2 * i] = f[0] * v[0] + c[0];
        rc[2 * i + 1] = f[1] * v[1] + c[1];
    }
}

/* largest real root of c0 + x^2 - x^3 (src/meta.jl:50-57: roots of Polynomial([c,0,1,-1]),
 * keep |imag| < 1e-10, take the maximum real part) */
static double max_real_root_cubic(double c0) {
    /* x^3 - x^2 - c0 = 0; substitute x = y + 1/3: y^3 + p y + q = 0 */
    const double p = -1.0 / 3.0, q = -2.0 / 27.0 - c0;
    double disc = q * q / 4 + p * p * p / 27;
    double x;
    if (disc > 0) {
        double sq = sqrt(disc);
        x = cbrt(-q / 2 + sq) + cbrt(-q / 2 - sq) + 1.0 / 3.0;
    } else { /* three real roots: the k = 0 branch gives the largest */
        double rho = 2 * sqrt(-p / 3);
        double arg = 3 * q / (p * rho); /* = cos(3 phi) */
        arg = arg > 1 ? 1 : (arg < -1 ? -1 : arg);
        x = rho * cos(acos(arg) / 3) + 1.0 / 3.0;
    }
    for (int it = 0; it < 3; it++) { /* Newton polish */
        double fx = ((x - 1) * x) * x - c0, dfx = (3 * x - 2) * x;
        if (dfx != 0) x -= fx / dfx;
    }
    return x;
}

/* src/meta.jl:60-76,82: inv_scale o inv_extrinsic o inv_perspective o inv_distort o inv_intrinsic */
void orc_map_inverse(int M, const double *rc, const double f[2], const double c[2], double k,
                     const double r[3], const double t[3], double cs, double *xyz) {
    double R[9];
    orc_rodrigues(r, R, NULL);
    /* inverse extrinsic: linear = R^T, translation = -R^T t */
    double ti[3];
    for (int a = 0; a < 3; a++) ti[a] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    const double l[3] = {R[2], R[5], R[8]}; /* last row of R^T */
    for (int i = 0; i < M; i++) {
        double v[2] = {(rc[2 * i] - c[0]) / f[0], (rc[2 * i + 1] - c[1]) / f[1]};
        if (k != 0) {
            double rho = max_real_root_cubic(k * (v[0] * v[0] + v[1] * v[1]));
            v[0] /= rho; v[1] /= rho;
        }
        double rc1[3] = {v[0], v[1], 1};
        double d = -ti[2] / (l[0] * rc1[0] + l[1] * rc1[1] + l[2] * rc1[2]);
        double p[3] = {d * rc1[0], d * rc1[1], d * rc1[2]};
        for (int a = 0; a < 3; a++) xyz[3 * i + a] = cs * (R[a] * p[0] + R[3 + a] * p[1] + R[6 + a] * p[2] + ti[a]);
    }
}

/* src/buildcalibrations.jl:28-31,37-67 */
void orc_errors(int N, int n1, int n2, const double *obj, const double *img, const double f[2],
                const double c[2], double k, const double *rvecs, const double *tvecs, double cs, int S,
                const double *inv_pts, double out[4]) {
    const int P = n1 * n2;
    double reproj = 0, projn = 0, dist = 0, inv = 0;
    double *rcbuf = (double *)malloc(sizeof(double) * 2 * (P > S ? P : S));
    double *xyzbuf = (double *)malloc(sizeof(double) * 3 * (P > S ? P : S));
    for (int i = 0; i < N; i++) {
        const double *r = rvecs + 3 * i, *t = tvecs + 3 * i, *mi = img + (size_t)2 * P * i;
        orc_map_forward(P, obj, f, c, k, r, t, cs, rcbuf);
        for (int q = 0; q < 2 * P; q++) reproj += (rcbuf[q] - mi[q]) * (rcbuf[q] - mi[q]);
        orc_map_inverse(P, mi, f, c, k, r, t, cs, xyzbuf);
        for (int q = 0; q < 3 * P; q++) projn += (xyzbuf[q] - obj[q]) * (xyzbuf[q] - obj[q]);
        for (int b = 0; b < n2; b++) /* diff along dim 1 (first index fastest) */
            for (int a = 0; a + 1 < n1; a++) {
                const double *u = xyzbuf + 3 * (a + n1 * b), *w = u + 3;
                double d = sqrt((w[0] - u[0]) * (w[0] - u[0]) + (w[1] - u[1]) * (w[1] - u[1]) +
                                (w[2] - u[2]) * (w[2] - u[2])) - cs;
                dist += d * d;
            }
        for (int b = 0; b + 1 < n2; b++)
            for (int a = 0; a < n1; a++) {
                const double *u = xyzbuf + 3 * (a + n1 * b), *w = u + 3 * n1;
                double d = sqrt((w[0] - u[0]) * (w[0] - u[0]) + (w[1] - u[1]) * (w[1] - u[1]) +
                                (w[2] - u[2]) * (w[2] - u[2])) - cs;
                dist += d * d;
            }
        if (S > 0 && inv_pts) {
            const double *pts = inv_pts + (size_t)2 * S * i;
            orc_map_inverse(S, pts, f, c, k, r, t, cs, xyzbuf);
            orc_map_forward(S, xyzbuf, f, c, k, r, t, cs, rcbuf);
            for (int q = 0; q < 2 * S; q++) inv += (pts[q] - rcbuf[q]) * (pts[q] - rcbuf[q]);
        }
    }
    double n = (double)P * N;
    out[0] = sqrt(reproj / n);
    out[1] = sqrt(projn / n);
    out[2] = sqrt(dist / ((double)(n1 - 1) * (n2 - 1)) / N);
    out[3] = S > 0 ? sqrt(inv / S / N) : 0;
    free(rcbuf); free(xyzbuf);
}
