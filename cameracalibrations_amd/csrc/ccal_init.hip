// ccal_init.hip -- closed-form initialisation of calibrateCamera on gfx950.
//
// What OpenCV does before its LM loop, behind the reference's call at src/detect_fit.jl:47
// (the reference passes no intrinsic guess, src/detect_fit.jl:40, so both stages always run):
//   A4  cvInitIntrinsicParams2D : per-image homography board -> image, focal lengths from the
//       orthogonality constraints of its columns
//   A5  cvFindExtrinsicCameraParams2, planar branch: homography on normalised points,
//       decomposition, orthonormalisation, 6-parameter LM polish; the homography's DLT is
//       taken from A4's pixel-space DLT by default (k_init_extrinsics HMODE 1: same system, OpenCV's trajectory)
//
// Mapping: ONE 64-lane workgroup (= one wavefront) per image, one lane per corner; the normal equations of these
// small problems (24-30 distinct sums over the corners) are formed per lane and summed over the lanes by a
// transpose-reduce through LDS (wave_sums_lds, ccal_wave.hpp: a 16-wide MFMA Gram computes 256 entries to deliver
// them, which cost a tile store, 14 matrix instructions and a read-back per pass), single sums by wave butterflies;
// the small dense algebra (9x9 inverse iteration, 8x8 / 6x6 Cholesky, 3x3 polar) is wave-uniform register
// arithmetic.  Per-image iteration counts differ, hence one wave per workgroup.
#include <hip/hip_runtime.h>

#include "ccal_acc.hpp"
#include "ccal_kernels.hpp"
#include "ccal_math.hpp"
#include "ccal_state.hpp"
#include "ccal_wave.hpp"

namespace ccal {

#ifdef CCAL_STAMPS
__device__ unsigned long long g_init_stamps[32];
__device__ unsigned int g_init_blk[4096 * 4];   // per block: total cycles, homography polish iters, pose LM iters, Jacobian passes
#define IBLK(j, v) do { if (blockIdx.x < 4096 && threadIdx.x == 0) g_init_blk[blockIdx.x * 4 + (j)] = (unsigned int)(v); } while (0)
static __device__ __forceinline__ unsigned long long memtime_now() {
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    return t_;
}
#define ISTAMP(i)                                                                             \
    do {                                                                                      \
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                            \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_init_stamps[i] = t_;                                                            \
        }                                                                                     \
    } while (0)
#define JSUB(i)                                                                               \
    do {                                                                                      \
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                            \
            __builtin_amdgcn_sched_barrier(0);                                                \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_init_stamps[11 + (i)] = t_;                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)
#define ICOUNT(i, v) do { if (blockIdx.x == 7 && threadIdx.x == 0) g_init_stamps[i] = (unsigned long long)(v); } while (0)
#define HSTAMP(i)                                                                             \
    do {                                                                                      \
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                            \
            __builtin_amdgcn_sched_barrier(0);                                                \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_init_stamps[16 + (i)] = t_;                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)
hipError_t debug_read_init_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_init_stamps), sizeof(unsigned long long) * 32);
}
hipError_t debug_read_init_blocks(unsigned int *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_init_blk), sizeof(unsigned int) * 4096 * 4);
}
#else
#define IBLK(j, v) do { } while (0)
#define JSUB(i) do { } while (0)
#define ISTAMP(i) do { } while (0)
#define ICOUNT(i, v) do { } while (0)
#define HSTAMP(i) do { } while (0)
#endif

constexpr double DBL_EPS = 2.220446049250313e-16;
constexpr double FLT_EPS = 1.1920928955078125e-07;

// ---------------------------------------------------------------------------------------
// point sources: (model xy, observed xy), both rounded to float32 as findHomography does
// ---------------------------------------------------------------------------------------
struct RawPoints {               // A4: board template xy -> pixel coordinates
    const float *obj;            // [3,P]
    const float2 *img;           // this image's [P]
    __device__ __forceinline__ void get(int p, double &Mx, double &My, double &mx, double &my) const {
        Mx = (double)obj[3 * p]; My = (double)obj[3 * p + 1];
        const float2 uv = img[p];
        mx = (double)uv.x; my = (double)uv.y;
    }
};

struct NormPoints {              // A5: centred template xy -> normalised (undistorted) image coordinates
    const float *obj;
    const float2 *img;
    double cMx, cMy;             // template centroid
    double in[NINTR];
    bool distorted;
    __device__ __forceinline__ void get(int p, double &Mx, double &My, double &mx, double &my) const {
        Mx = (double)(float)((double)obj[3 * p] - cMx);
        My = (double)(float)((double)obj[3 * p + 1] - cMy);
        const float2 uv = img[p];
        double x = ((double)uv.x - in[2]) / in[0], y = ((double)uv.y - in[3]) / in[1];
        if (distorted) {         // [OCV-K cvUndistortPoints]: 5 fixed-point iterations
            const double x0 = x, y0 = y;
            for (int j = 0; j < 5; j++) {
                const double r2 = x * x + y * y;
                const double icd = 1.0 / (1.0 + ((in[8] * r2 + in[5]) * r2 + in[4]) * r2);
                if (icd < 0.0) { x = x0; y = y0; break; }
                const double dx = 2.0 * in[6] * x * y + in[7] * (r2 + 2.0 * x * x);
                const double dy = in[6] * (r2 + 2.0 * y * y) + 2.0 * in[7] * x * y;
                x = (x0 - dx) * icd; y = (y0 - dy) * icd;
            }
        }
        mx = (double)(float)x; my = (double)(float)y;
    }
};

// The correspondences of one image staged in LDS ([4][P]: Mx, My, mx, my), filled once per image: the homography makes
// a dozen passes over the same points, and NormPoints::get (two divisions, five undistortion iterations) inlined at
// every one of them made k_init_extrinsics 54 KB of code for a one-wave kernel that lives on instruction fetch.
struct LdsPoints {
    const double *s;
    int P;
    __device__ __forceinline__ void get(int p, double &Mx, double &My, double &mx, double &my) const {
        Mx = s[p]; My = s[P + p]; mx = s[2 * P + p]; my = s[3 * P + p];
    }
};

// A point source with this lane's first-pass point (index min(lane, P - 1): what every pass over the points asks for
// first) kept in registers: one image is a dozen to several dozen passes over the same 54 points, and every pass
// otherwise starts with a round trip to memory (and, for NormPoints, two divisions).  Same values, same results.
template <class Inner>
struct CachedPoints {
    Inner inner;
    int cidx;
    double cMx, cMy, cmx, cmy;
    __device__ __forceinline__ void init(const Inner &src, int lane, int P) {
        inner = src;
        cidx = lane < P ? lane : P - 1;
        inner.get(cidx, cMx, cMy, cmx, cmy);
    }
    __device__ __forceinline__ void get(int p, double &Mx, double &My, double &mx, double &my) const {
        if (p == cidx) { Mx = cMx; My = cMy; mx = cmx; my = cmy; }
        else inner.get(p, Mx, My, mx, my);
    }
};

// ---------------------------------------------------------------------------------------
// homography residual (+ Jacobian rows) of one correspondence, h[8] = 1
// [OCV-K HomographyRefineCallback::compute]
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void homog_point(const double h[8], double Mx, double My, double mx, double my,
                                            double &eu, double &ev, double &ww, double &xi, double &yi) {
    ww = h[6] * Mx + h[7] * My + 1.0;
    ww = fabs(ww) > DBL_EPS ? 1.0 / ww : 0.0;
    xi = (h[0] * Mx + h[1] * My + h[2]) * ww;
    yi = (h[3] * Mx + h[4] * My + h[5]) * ww;
    eu = xi - mx;
    ev = yi - my;
}

// residual sum and largest |residual| at h (the trial evaluation of LMSolver; used here after a rejected step)
template <class Pts>
__device__ double homog_error(const Pts &pts, int P, int lane, const double h[8], double &rmax) {
    double s = 0.0, m = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        if (p < P) {
            double Mx, My, mx, my, eu, ev, ww, xi, yi;
            pts.get(p, Mx, My, mx, my);
            homog_point(h, Mx, My, mx, my, eu, ev, ww, xi, yi);
            s += eu * eu + ev * ev;
            m = fmax(m, fmax(fabs(eu), fabs(ev)));
        }
    }
    rmax = wave_max(m);
    return wave_sum(s);
}

// ---- SPD solves with ONE MATRIX ROW PER LANE (round 5) ----------------------------------------------------------------------
// The step of the per-view polishes solves a damped 6 x 6 (pose) system.  As wave-uniform arithmetic (every lane the same
// Cholesky factor + two substitutions) that is a 1.0 k-cycle dependent chain per iteration of the image the launch waits for; with
// row i of [A | b] in lane i of every row of 16 lanes and the pivot row broadcast by DPP (the scaled Gauss-Jordan elimination of
// the LM advance, ccal_math.hpp gj_row_step: raw v_rcp_f64 estimate, no refined reciprocal on the chain) it is N pivots of four
// dependent operations.  The rows come straight out of the lane sums' LDS totals (wave_sums_lds_reduce): 7 reads per lane
// instead of 28 broadcast reads.
#ifndef CCAL_POSE_GJ
#define CCAL_POSE_GJ 1
#endif
#ifndef CCAL_HOMOG_BLOCKS
#define CCAL_HOMOG_BLOCKS 1
#endif
__device__ __forceinline__ double rcp_refined(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}
template <int N, int J>
__device__ __forceinline__ void gj_rows_steps(double (&a)[N], double &b, double &diag, bool &ok, int la) {
    const double d = row_bcast16<J>(a[J]);          // pivot
    ok = ok && (d > 0.0) && (d < 1.0e300);
    double prow[N];
#pragma unroll
    for (int k = 0; k < N; k++) prow[k] = k > J ? row_bcast16<J>(a[k]) : 0.0;
    const double pb = row_bcast16<J>(b);
    gj_row_step<N>(J, la == J, d, __builtin_amdgcn_rcp(d), prow, pb, a, b, diag);
    if constexpr (J + 1 < N) gj_rows_steps<N, J + 1>(a, b, diag, ok, la);
}
template <int N, int K>
__device__ __forceinline__ void bcast_rows(double xv, double (&x)[N]) {
    x[K] = row_bcast16<K>(xv);
    if constexpr (K + 1 < N) bcast_rows<N, K + 1>(xv, x);
}
// a = row `la` of the SPD matrix, b = its right-hand side entry (la = min(lane & 15, N - 1): every row of 16 lanes holds the whole
// system, lanes beyond N - 1 mirror the last row) -> x[N] in every lane; false: a pivot was not positive (x unusable)
template <int N>
__device__ __forceinline__ bool gj_solve_rows(double (&a)[N], double b, int la, double (&x)[N]) {
    bool ok = true;
    double diag = 1.0;
    gj_rows_steps<N, 0>(a, b, diag, ok, la);
    const double rd = rcp_refined(diag);
    double xv = b * rd;
    xv = fma(fma(-diag, xv, b), rd, xv);            // x_la = b / a_la,la, one correction step on the quotient
    bcast_rows<N, 0>(xv, x);
    return (__ballot(ok) & 1ull) != 0;              // (lane 0's view: every lane made the same pivot tests)
}

// A (8x8 packed lower) = J^T J, v = J^T r, S = r^T r at h.  The u-row of a correspondence touches only parameters
// {0,1,2,6,7} and the v-row only {3,4,5,6,7}, and both carry the same first three entries (Mx w, My w, w): of the 45
// entries 30 are distinct non-zero sums -- formed per lane, accumulated over the passes and summed over the lanes once
// (wave_sums_lds), instead of a 16-wide MFMA Gram per pass that computes 256 entries to deliver them.
constexpr int HSUMS = 30;
template <class Pts>
__device__ void homog_normal_eq(const Pts &pts, int P, int lane, const double h[8], double *scratch,
                                double A[36], double v[8], double &S, double &rmax) {
    double acc[HSUMS], emax = 0.0;
#pragma unroll
    for (int k = 0; k < HSUMS; k++) acc[k] = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int pi = c0 + lane;
        const bool valid = pi < P;
        double Mx, My, mx, my, eu, ev, ww, xi, yi;
        pts.get(valid ? pi : P - 1, Mx, My, mx, my);
        homog_point(h, Mx, My, mx, my, eu, ev, ww, xi, yi);
        const double vm = valid ? 1.0 : 0.0;                  // padding lanes: zero rows
        const double w = ww * vm, p = Mx * w, q = My * w;
        eu *= vm; ev *= vm;
        emax = fmax(emax, fmax(fabs(eu), fabs(ev)));
        const double ux = -p * xi, uy = -q * xi, vx = -p * yi, vy = -q * yi;
        acc[0] += p * p; acc[1] += p * q; acc[2] += q * q; acc[3] += p * w; acc[4] += q * w; acc[5] += w * w;
        acc[6] += ux * p; acc[7] += ux * q; acc[8] += ux * w; acc[9] += uy * p; acc[10] += uy * q; acc[11] += uy * w;
        acc[12] += vx * p; acc[13] += vx * q; acc[14] += vx * w; acc[15] += vy * p; acc[16] += vy * q; acc[17] += vy * w;
        acc[18] += ux * ux + vx * vx; acc[19] += ux * uy + vx * vy; acc[20] += uy * uy + vy * vy;
        acc[21] += p * eu; acc[22] += q * eu; acc[23] += w * eu; acc[24] += p * ev; acc[25] += q * ev; acc[26] += w * ev;
        acc[27] += ux * eu + vx * ev; acc[28] += uy * eu + vy * ev;
        acc[29] += eu * eu + ev * ev;
    }
    double t[HSUMS];
    rmax = wave_max(emax);
    wave_sums_lds<HSUMS>(acc, scratch, lane, t);
    // (0..2)^2 and (3..5)^2 blocks are the same sums; (3..5) x (0..2) is structurally zero
    A[tri(0, 0)] = t[0]; A[tri(1, 0)] = t[1]; A[tri(1, 1)] = t[2]; A[tri(2, 0)] = t[3]; A[tri(2, 1)] = t[4]; A[tri(2, 2)] = t[5];
    A[tri(3, 3)] = t[0]; A[tri(4, 3)] = t[1]; A[tri(4, 4)] = t[2]; A[tri(5, 3)] = t[3]; A[tri(5, 4)] = t[4]; A[tri(5, 5)] = t[5];
#pragma unroll
    for (int a = 3; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) A[tri(a, b)] = 0.0;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        A[tri(6, b)] = t[6 + b]; A[tri(7, b)] = t[9 + b];
        A[tri(6, 3 + b)] = t[12 + b]; A[tri(7, 3 + b)] = t[15 + b];
        v[b] = t[21 + b]; v[3 + b] = t[24 + b];
    }
    A[tri(6, 6)] = t[18]; A[tri(7, 6)] = t[19]; A[tri(7, 7)] = t[20];
    v[6] = t[27]; v[7] = t[28];
    S = t[29];
}

// d = (A + lambda diag(D))^-1 v for the homography polish's 8 x 8 normal equations, by their block structure (round 5):
//     A = [T 0 U^T; 0 T V^T; U V W]   T 3 x 3 (the SAME block twice: both image coordinates see the same template moments, and
//                                     D = the first iteration's diagonal damps both alike), U, V 2 x 3, W 2 x 2
// One 3 x 3 Cholesky factor serves both blocks, the 2 x 2 Schur complement S = W - U T^-1 U^T - V T^-1 V^T closes the system:
// five pivots on the dependent chain instead of the eight of a dense 8 x 8 factor, half the arithmetic.  false: not positive
// definite (the dense factor's test, on the same pivots up to elimination order).
__device__ __forceinline__ bool homog_solve_blocks(const double A[36], const double v[8], double lambda, const double D[8], double d[8]) {
    double T[6], it[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) T[tri(a, b)] = A[tri(a, b)] + (a == b ? lambda * D[a] : 0.0);
    bool ok = chol_packed<3>(T, it);
    // Y = L^-1 [U^T | V^T | v_p | v_q]: columns u0 u1 (rows 6, 7 of A, columns 0..2), w0 w1 (columns 3..5), v_p, v_q
    double y[6][3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        y[0][b] = A[tri(6, b)]; y[1][b] = A[tri(7, b)]; y[2][b] = A[tri(6, 3 + b)]; y[3][b] = A[tri(7, 3 + b)];
        y[4][b] = v[b]; y[5][b] = v[3 + b];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) fwd_subst<3>(T, it, y[c]);
    auto dot3 = [](const double (&p)[3], const double (&q)[3]) { return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]; };
    double S[3], is[2], r[2];
    S[0] = A[tri(6, 6)] + lambda * D[6] - dot3(y[0], y[0]) - dot3(y[2], y[2]);
    S[1] = A[tri(7, 6)] - dot3(y[1], y[0]) - dot3(y[3], y[2]);
    S[2] = A[tri(7, 7)] + lambda * D[7] - dot3(y[1], y[1]) - dot3(y[3], y[3]);
    r[0] = v[6] - dot3(y[0], y[4]) - dot3(y[2], y[5]);
    r[1] = v[7] - dot3(y[1], y[4]) - dot3(y[3], y[5]);
    ok = chol_packed<2>(S, is) && ok;
    fwd_subst<2>(S, is, r);
    bwd_subst<2>(S, is, r);
    double p[3], q[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        p[b] = y[4][b] - y[0][b] * r[0] - y[1][b] * r[1];
        q[b] = y[5][b] - y[2][b] * r[0] - y[3][b] * r[1];
    }
    bwd_subst<3>(T, it, p);
    bwd_subst<3>(T, it, q);
#pragma unroll
    for (int b = 0; b < 3; b++) { d[b] = p[b]; d[3 + b] = q[b]; }
    d[6] = r[0]; d[7] = r[1];
    return ok;
}

// [OCV-K LMSolverImpl::run] with maxIters = 10, eps = FLT_EPSILON on the 8 free entries
// SPEC = false: every trial by the error-only pass (LMSolver's own sequence): 44 fewer live doubles, which keeps
// k_init_homography at two waves per SIMD where that matters (more images than SIMDs).
template <bool SPEC, class Pts>
__device__ void homog_refine(const Pts &pts, int P, int lane, double *scratch, double x[8]) {
    // While steps are being accepted a trial point is evaluated with the normal-equations pass straight away (its residual
    // sum and largest residual are what the gain ratio and the stop rule need): on acceptance those normal equations are the
    // next iteration's, and LMSolver's separate error-only evaluation is saved.  After a rejected step -- near convergence
    // they come in runs: the polish on pixel coordinates spends its last iterations there -- the trial goes back to the
    // cheap error-only pass, and the normal equations follow only on acceptance.  Same decisions, same points.
    // (The lane-parallel solve of the pose step -- gj_solve_rows -- was tried here too, round 5, with the rows of the 8 x 8 system
    //  read out of the lane sums' LDS totals through an index map: slower -- C1's pose stage 30.5 -> 47.8 us, C5's intrinsics stage
    //  183 -> 258 us: both kernels sit at their register limit and the extra row state spills.)
    double A[36], v[8], D[8], S, rmax;
    homog_normal_eq(pts, P, lane, x, scratch, A, v, S, rmax);
#pragma unroll
    for (int a = 0; a < 8; a++) D[a] = A[tri(a, a)];
    const double Rlo = 0.25, Rhi = 0.75;
    double lambda = 1.0, lc = 0.75;
    bool last_accepted = SPEC;
    HSTAMP(6);
    for (int iter = 0;;) {
        HSTAMP(7);
        double d[8], xd[8];
        // (latency form only: the throughput form runs at its register limit, and C5's intrinsics stage measured 181 -> 235 us with
        //  the block solve's extra live values; C3: intrinsics stage 33.2 -> 31.8 us, full fit -1.8 us, tools/init_timing.py)
        if constexpr (CCAL_HOMOG_BLOCKS != 0 && SPEC) {
            if (!homog_solve_blocks(A, v, lambda, D, d)) break;   // degenerate normal equations: keep the current H
        } else {
        double Ap[36];
#pragma unroll
        for (int k = 0; k < 36; k++) Ap[k] = A[k];
#pragma unroll
        for (int a = 0; a < 8; a++) { Ap[tri(a, a)] += lambda * D[a]; d[a] = v[a]; }
        double ipv[8];
        if (!chol_packed<8>(Ap, ipv)) break;            // degenerate normal equations: keep the current H
        fwd_subst<8>(Ap, ipv, d);
        bwd_subst<8>(Ap, ipv, d);
        }
#pragma unroll
        for (int a = 0; a < 8; a++) xd[a] = x[a] - d[a];
        HSTAMP(8);
        double Ac[36], vc[8], Sd, rdmax;
        if (SPEC && last_accepted) homog_normal_eq(pts, P, lane, xd, scratch, Ac, vc, Sd, rdmax);
        else Sd = homog_error(pts, P, lane, xd, rdmax);
        HSTAMP(9);
        double dS = 0.0, dv = 0.0, dmax = 0.0;          // dS = d . (2 v - A d)
#pragma unroll
        for (int a = 0; a < 8; a++) {
            double Ad = 0.0;
#pragma unroll
            for (int b = 0; b < 8; b++) Ad += A[a >= b ? tri(a, b) : tri(b, a)] * d[b];
            dS += d[a] * (2.0 * v[a] - Ad);
            dv += d[a] * v[a];
            dmax = fmax(dmax, fabs(d[a]));
        }
        const double Rr = (S - Sd) / (fabs(dS) > DBL_EPS ? dS : 1.0);
        if (Rr > Rhi) {
            lambda *= 0.5;
            if (lambda < lc) lambda = 0.0;
        } else if (Rr < Rlo) {
            double nu = (Sd - S) / (fabs(dv) > DBL_EPS ? dv : 1.0) + 2.0;
            nu = nu < 2.0 ? 2.0 : (nu > 10.0 ? 10.0 : nu);
            if (lambda == 0.0) {
                // lambda = lc = 1 / max |diag(A^-1)|;  diag(A^-1)[a] = || L^-1 e_a ||^2
                double L[36];
#pragma unroll
                for (int k = 0; k < 36; k++) L[k] = A[k];
                double maxval = DBL_EPS, ipl[8];
                if (chol_packed<8>(L, ipl)) {
#pragma unroll
                    for (int a = 0; a < 8; a++) {
                        double e[8];
#pragma unroll
                        for (int b = 0; b < 8; b++) e[b] = (a == b) ? 1.0 : 0.0;
                        fwd_subst<8>(L, ipl, e);
                        double s2 = 0.0;
#pragma unroll
                        for (int b = 0; b < 8; b++) s2 += e[b] * e[b];
                        maxval = fmax(maxval, s2);
                    }
                }
                lambda = lc = 1.0 / maxval;
                nu *= 0.5;
            }
            lambda *= nu;
        }
        HSTAMP(10);
        if (Sd < S) {
#pragma unroll
            for (int a = 0; a < 8; a++) x[a] = xd[a];
            if (SPEC && last_accepted) {
#pragma unroll
                for (int a = 0; a < 8; a++) v[a] = vc[a];
#pragma unroll
                for (int k = 0; k < 36; k++) A[k] = Ac[k];
                S = Sd;
            } else {
                double r_unused;
                homog_normal_eq(pts, P, lane, x, scratch, A, v, S, r_unused);
            }
            rmax = rdmax;
            last_accepted = SPEC;
        } else {
            last_accepted = false;
        }
        HSTAMP(11);
        iter++;
        if (!(iter < 10 && dmax >= FLT_EPS && rmax >= FLT_EPS)) { ICOUNT(7, iter); IBLK(1, iter); break; }
    }
}

// [OCV-K findHomography(method 0)]: per-axis normalised DLT (smallest eigenvector of LtL by
// shifted inverse iteration) + LM polish.  Every lane returns the same H.  false = degenerate.
template <bool SPEC, class Pts>
__device__ bool wave_homography(const Pts &pts, int P, int lane, double *scratch, double H[9], double *H_dlt = nullptr) {
    HSTAMP(0);
    double cM0 = 0, cM1 = 0, cm0 = 0, cm1 = 0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        if (p < P) {
            double Mx, My, mx, my;
            pts.get(p, Mx, My, mx, my);
            cM0 += Mx; cM1 += My; cm0 += mx; cm1 += my;
        }
    }
    cM0 = wave_sum(cM0) / P; cM1 = wave_sum(cM1) / P; cm0 = wave_sum(cm0) / P; cm1 = wave_sum(cm1) / P;
    double sM0 = 0, sM1 = 0, sm0 = 0, sm1 = 0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        if (p < P) {
            double Mx, My, mx, my;
            pts.get(p, Mx, My, mx, my);
            sM0 += fabs(Mx - cM0); sM1 += fabs(My - cM1); sm0 += fabs(mx - cm0); sm1 += fabs(my - cm1);
        }
    }
    sM0 = wave_sum(sM0); sM1 = wave_sum(sM1); sm0 = wave_sum(sm0); sm1 = wave_sum(sm1);
    if (fabs(sm0) < DBL_EPS || fabs(sm1) < DBL_EPS || fabs(sM0) < DBL_EPS || fabs(sM1) < DBL_EPS) return false;
    sm0 = P / sm0; sm1 = P / sm1; sM0 = P / sM0; sM1 = P / sM1;

    HSTAMP(1);
    // LtL = sum Lx Lx^T + Ly Ly^T with Lx = [X Y 1 | 0 0 0 | -xX -xY -x], Ly = [0 0 0 | X Y 1 | -yX -yY -y]: with
    // t = (X, Y, 1) the 45 entries are 24 distinct sums -- t (x) t (twice on the diagonal), -x t (x) t, -y t (x) t,
    // (x^2 + y^2) t (x) t -- formed per lane and summed over the lanes once (wave_sums_lds)
    constexpr int LSUMS = 24;
    double lacc[LSUMS];
#pragma unroll
    for (int k = 0; k < LSUMS; k++) lacc[k] = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < P;
        double Mx, My, mx, my;
        pts.get(valid ? p : P - 1, Mx, My, mx, my);
        const double x = (mx - cm0) * sm0, y = (my - cm1) * sm1;
        const double one = valid ? 1.0 : 0.0;                 // padding lanes: zero rows
        const double X = (Mx - cM0) * sM0 * one, Y = (My - cM1) * sM1 * one;
        const double tt[6] = {X * X, X * Y, Y * Y, X, Y, one};   // t (x) t, packed lower (XX XY YY X1 Y1 11)
        const double r2 = x * x + y * y;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            lacc[k] += tt[k];
            lacc[6 + k] -= x * tt[k];
            lacc[12 + k] -= y * tt[k];
            lacc[18 + k] += r2 * tt[k];
        }
    }
    double ls[LSUMS];
    wave_sums_lds<LSUMS>(lacc, scratch, lane, ls);
    double A[45], trace = 0.0;
    // packed-lower index of t_i t_j: (0,0) (1,0) (1,1) (2,0) (2,1) (2,2) == tri(i, j)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int k = i >= j ? tri(i, j) : tri(j, i);
            if (i >= j) { A[tri(i, j)] = ls[k]; A[tri(3 + i, 3 + j)] = ls[k]; A[tri(6 + i, 6 + j)] = ls[18 + k]; }
            A[tri(3 + i, j)] = 0.0;
            A[tri(6 + i, j)] = ls[6 + k];
            A[tri(6 + i, 3 + j)] = ls[12 + k];
        }
#pragma unroll
    for (int a = 0; a < 9; a++) trace += A[tri(a, a)];
    HSTAMP(2);
    // smallest eigenvector: inverse iteration on LtL + mu I (mu keeps the factor positive
    // definite for noise-free data, where the smallest eigenvalue is 0 up to rounding)
    const double mu_shift = 1e-11 * trace;
#pragma unroll
    for (int a = 0; a < 9; a++) A[tri(a, a)] += mu_shift;
    double ipa[9];
    if (!chol_packed<9>(A, ipa)) return false;
    double hv[9];
#pragma unroll
    for (int a = 0; a < 9; a++) hv[a] = 1.0 / 3.0;
    for (int it = 0; it < 12; it++) {
        double prev[9];
#pragma unroll
        for (int a = 0; a < 9; a++) prev[a] = hv[a];
        fwd_subst<9>(A, ipa, hv);
        bwd_subst<9>(A, ipa, hv);
        double nrm = 0.0;
#pragma unroll
        for (int a = 0; a < 9; a++) nrm += hv[a] * hv[a];
        nrm = rsqrt_pos(nrm);
        double change = 0.0;
#pragma unroll
        for (int a = 0; a < 9; a++) {
            hv[a] *= nrm;
            change = fmax(change, fabs(hv[a] - prev[a]));
        }
        if (change < 1.0e-15) break;          // converged to rounding (typically 3-4 iterations)
    }
    HSTAMP(3);
    // H = invHnorm * H0 * Hnorm2, scaled so that H[8] = 1
    const double i0 = 1.0 / sm0, i1 = 1.0 / sm1;
    double T[9];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        T[j] = i0 * hv[j] + cm0 * hv[6 + j];
        T[3 + j] = i1 * hv[3 + j] + cm1 * hv[6 + j];
        T[6 + j] = hv[6 + j];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        H[a * 3] = T[a * 3] * sM0;
        H[a * 3 + 1] = T[a * 3 + 1] * sM1;
        H[a * 3 + 2] = -T[a * 3] * cM0 * sM0 - T[a * 3 + 1] * cM1 * sM1 + T[a * 3 + 2];
    }
    const double s = 1.0 / H[8];
#pragma unroll
    for (int k = 0; k < 9; k++) H[k] *= s;
    HSTAMP(4);
    if (H_dlt != nullptr && lane < 9) {           // the DLT's result before the polish (every lane holds the same H)
        double v = H[0];
#pragma unroll
        for (int k = 1; k < 9; k++) v = (lane == k) ? H[k] : v;
        H_dlt[lane] = v;
    }
    if (P > 4) {
        homog_refine<SPEC>(pts, P, lane, scratch, H);
        H[8] = 1.0;
    }
    HSTAMP(5);
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 9; k++) finite = finite && (fabs(H[k]) < 1.0e300);
    return finite;
}

// ---- the focal least squares summed EXACTLY across the images of one GPU (ccal_acc.hpp's fixed-point limbs, as the LM's reduced
//      system): k_init_homography adds its image's five sums + failure flag with integer atomics into replica (image mod 8) of
//      [FOC_REPL][ACC_LIMBS][NFOC] int64 words; whoever needs the totals -- k_init_focal_acc, or every workgroup of the pose
//      stage in its prologue when the stages run as one pipeline (no focal kernel, no boundary in between) -- reads the eight
//      replicas, decodes and solves: the same bits everywhere, whatever the order the images finished in.  Two buffers rotate:
//      the reader of one clears the other for the next fit. ----
constexpr int FOC_REPL = 8;
constexpr int FOC_WORDS = ACC_LIMBS * NFOC;          // 40 int64 words per replica
static_assert(FOC_REPL * FOC_WORDS == LM_FOC_ACC_WORDS, "host-side size of one focal accumulator buffer");

__device__ __forceinline__ void foc_acc_add(long long *__restrict__ rep, int col, double v) {
    int j, flag;
    int64_t lo, hi;
    acc_split(v, j, lo, hi, flag);
    if (flag) {
        __hip_atomic_fetch_add(rep + (flag - 1) * NFOC + (NFOC - 1), 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (lo != 0) __hip_atomic_fetch_add(rep + j * NFOC + col, (long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hi != 0) __hip_atomic_fetch_add(rep + (j + 1) * NFOC + col, (long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------
// A4: homography per image + its two rows of the focal-length least squares
// ---------------------------------------------------------------------------------------
// H_out [N][2][9]: the polished homography (what the focal-length least squares uses) and, behind it, the DLT's result before
// the polish (what k_init_extrinsics starts OpenCV's second polish from, HMODE 1)
// (the kernel's body, image i: shared by the ordinary launch -- image = workgroup -- and the merged launch of several fits)
template <bool SPEC>
__device__ __forceinline__ void init_homography_body(const float *__restrict__ obj, const float *__restrict__ img,
                                                     double *__restrict__ H_out, double *__restrict__ foc,
                                                     long long *__restrict__ foc_acc, int *__restrict__ fail0, int N,
                                                     int P, int width, int height, const int i) {
    __shared__ __attribute__((aligned(16))) double s_sums[sums_scratch_doubles(HSUMS)];
    const int lane = threadIdx.x;
    (void)N;
    if (fail0 != nullptr && i == 0 && lane == 0) *fail0 = 0;     // (pipeline form: no focal kernel will zero the pose stage's failure counter)
    const RawPoints raw{obj, reinterpret_cast<const float2 *>(img) + (size_t)i * P};
    CachedPoints<RawPoints> pts;
    pts.init(raw, lane, P);
    double H[9];
    if (lane < 9) H_out[(size_t)18 * i + 9 + lane] = 0.0;       // (a homography that fails before the polish leaves zeros)
    const bool ok = wave_homography<SPEC>(pts, P, lane, s_sums, H, H_out + (size_t)18 * i + 9);
    if (lane < 9) {
        H_out[(size_t)18 * i + lane] = ok ? H[lane] : 0.0;
        if (!ok) H_out[(size_t)18 * i + 9 + lane] = 0.0;
    }
    // [OCV-K cvInitIntrinsicParams2D]
    const double a2 = !width ? 0.5 : (width - 1) * 0.5, a5 = !height ? 0.5 : (height - 1) * 0.5;
    H[0] -= H[6] * a2; H[1] -= H[7] * a2; H[2] -= H[8] * a2;
    H[3] -= H[6] * a5; H[4] -= H[7] * a5; H[5] -= H[8] * a5;
    double hh[3], vv[3], d1[3], d2[3], n[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const double t0 = H[j * 3], t1 = H[j * 3 + 1];
        hh[j] = t0; vv[j] = t1; d1[j] = (t0 + t1) * 0.5; d2[j] = (t0 - t1) * 0.5;
        n[0] += t0 * t0; n[1] += t1 * t1; n[2] += d1[j] * d1[j]; n[3] += d2[j] * d2[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) n[j] = 1.0 / sqrt(n[j]);
#pragma unroll
    for (int j = 0; j < 3; j++) { hh[j] *= n[0]; vv[j] *= n[1]; d1[j] *= n[2]; d2[j] *= n[3]; }
    const double Ap[4] = {hh[0] * vv[0], hh[1] * vv[1], d1[0] * d2[0], d1[1] * d2[1]};
    const double bp[2] = {-hh[2] * vv[2], -d1[2] * d2[2]};
    double out[NFOC];
    out[0] = Ap[0] * Ap[0] + Ap[2] * Ap[2];
    out[1] = Ap[0] * Ap[1] + Ap[2] * Ap[3];
    out[2] = Ap[1] * Ap[1] + Ap[3] * Ap[3];
    out[3] = Ap[0] * bp[0] + Ap[2] * bp[1];
    out[4] = Ap[1] * bp[0] + Ap[3] * bp[1];
    out[5] = ok ? 0.0 : 1.0;
    out[6] = out[7] = 0.0;
    if (!ok) { out[0] = out[1] = out[2] = out[3] = out[4] = 0.0; }
    if (lane < NFOC) {
        double v = out[0];
#pragma unroll
        for (int k = 1; k < NFOC; k++) v = (lane == k) ? out[k] : v;
        foc[(size_t)NFOC * i + lane] = v;
        // this GPU holds every image: the sums are formed right here, exactly (see foc_acc_add)
        if (foc_acc != nullptr && lane < NFOC - 2) foc_acc_add(foc_acc + (i & (FOC_REPL - 1)) * FOC_WORDS, lane, v);
    }
}

template <bool SPEC>
__global__ __launch_bounds__(64) void k_init_homography(const float *__restrict__ obj, const float *__restrict__ img,
                                                         double *__restrict__ H_out, double *__restrict__ foc,
                                                         long long *__restrict__ foc_acc, int *__restrict__ fail0, int N,
                                                         int P, int width, int height) {
    init_homography_body<SPEC>(obj, img, H_out, foc, foc_acc, fail0, N, P, width, height, (int)blockIdx.x);
}

// merged launch (ccal_fit_many): the images of several independent fits in one grid; workgroup -> fit -> that fit's arguments
struct HomArgs {
    const float *obj, *img;
    double *H_out, *foc;
    long long *foc_acc;
    int *fail0;
    int N, P, width, height;
};
template <bool SPEC>
__global__ __launch_bounds__(64) void k_init_homography_multi(const HomArgs *__restrict__ table, const int *__restrict__ map,
                                                               const int *__restrict__ first) {
    const int fit = map[blockIdx.x];
    const HomArgs a = table[fit];
    init_homography_body<SPEC>(a.obj, a.img, a.H_out, a.foc, a.foc_acc, a.fail0, a.N, a.P, a.width, a.height, (int)blockIdx.x - first[fit]);
}

// [OCV-K cvInitIntrinsicParams2D]: cvSolve(A, b, f, CV_NORMAL + CV_SVD) on the 2x2 normal equations (one thread)
__device__ __forceinline__ void focal_from_sums(const double *red, int width, int height, double aspect, double intr[NINTR], int &bad) {
    double A[3] = {red[0], red[1], red[2]}, f[2] = {red[3], red[4]};
    bad = red[5] > 0.0 ? 1 : 0;
    double ip2[2];
    if (chol_packed<2>(A, ip2)) {
        fwd_subst<2>(A, ip2, f);
        bwd_subst<2>(A, ip2, f);
    } else {
        bad = 1;
    }
    double fx = sqrt(fabs(1.0 / f[0])), fy = sqrt(fabs(1.0 / f[1]));
    if (aspect != 0.0) {
        const double tf = (fx + fy) / (aspect + 1.0);
        fx = aspect * tf;
        fy = tf;
    }
    intr[0] = fx;
    intr[1] = fy;
    intr[2] = !width ? 0.5 : (width - 1) * 0.5;
    intr[3] = !height ? 0.5 : (height - 1) * 0.5;
#pragma unroll
    for (int a = 4; a < NINTR; a++) intr[a] = 0.0;
}

__device__ void focal_solve(const double *red, double *__restrict__ intr0, int *__restrict__ fail_count, int width, int height,
                            double aspect) {
    double intr[NINTR];
    int bad;
    focal_from_sums(red, width, height, aspect, intr, bad);
    for (int a = 0; a < NINTR; a++) intr0[a] = intr[a];
    fail_count[1] = bad;
    fail_count[0] = 0;                 // the pose stage that follows counts its failed views here
}

// one wave: the totals of one buffer -> s_red[NFOC] (LDS); s_part: FOC_WORDS AccPart of LDS scratch.  A NaN / out-of-range
// contribution anywhere turns the sums into NaN (the solve then reports a failed focal estimate).
__device__ __forceinline__ void foc_acc_read(const long long *__restrict__ acc, AccPart *s_part, double *s_red, int lane) {
    if (lane < FOC_WORDS) {
        // every replica is split on its own: one address takes up to N / FOC_REPL adds of 52-bit payloads (< 2^63 for
        // N <= LM_FOC_MAX_IMAGES), but the total over the replicas would not fit an int64 beyond ~4096 images; the binary64
        // halves stay below 2^36 in magnitude, so their sums are exact
        AccPart p;
        p.hi = p.lo = 0.0;
#pragma unroll
        for (int r = 0; r < FOC_REPL; r++) acc_part_add(p, acc[r * FOC_WORDS + lane]);
        acc_part_norm(p);
        s_part[lane] = p;
    }
    wave_lds_sync();
    if (lane < NFOC) {
        AccPart part[ACC_LIMBS];
#pragma unroll
        for (int j = 0; j < ACC_LIMBS; j++) part[j] = s_part[j * NFOC + lane];
        double v = acc_decode(part);
        const AccPart n_nan = s_part[0 * NFOC + NFOC - 1], n_big = s_part[1 * NFOC + NFOC - 1];
        if (lane >= NFOC - 2) v = 0.0;                // (the two pad columns carry no value; the last one carries the flags)
        if (n_nan.hi != 0.0 || n_nan.lo != 0.0 || n_big.hi != 0.0 || n_big.lo != 0.0) v = __longlong_as_double(0x7ff8000000000000ll);
        s_red[lane] = v;
    }
    wave_lds_sync();
}

// red_out != NULL: an image-sharded fit -- this rank's sums go out as NFOC doubles (summed across the ranks before the solve)
__global__ __launch_bounds__(64) void k_init_focal_acc(const long long *__restrict__ acc, long long *__restrict__ acc_clear,
                                                        double *__restrict__ red_out, double *__restrict__ intr0,
                                                        int *__restrict__ fail_count, int width, int height, double aspect) {
    __shared__ AccPart s_part[FOC_WORDS];
    __shared__ double s_red[NFOC];
    const int lane = threadIdx.x;
    foc_acc_read(acc, s_part, s_red, lane);
    if (red_out != nullptr) {
        if (lane < NFOC) red_out[lane] = s_red[lane];
    } else if (lane == 0) {
        focal_solve(s_red, intr0, fail_count, width, height, aspect);
    }
    for (int k = lane; k < FOC_REPL * FOC_WORDS; k += 64) acc_clear[k] = 0;
}

// intr0 != NULL: this GPU holds every image, the solve follows in the same launch (no all-reduce in between)
__global__ __launch_bounds__(1024) void k_init_focal_reduce(const double *__restrict__ foc, double *__restrict__ red,
                                                             int N, double *__restrict__ intr0, int *__restrict__ fail_count,
                                                             int width, int height, double aspect) {
    // 128 slices x NFOC columns; slice sums are combined in a fixed order (deterministic)
    __shared__ double part[128][NFOC];
    __shared__ double fin[NFOC];
    const int j = threadIdx.x % NFOC, c = threadIdx.x / NFOC;
    double s = 0.0;
    for (int i = c; i < N; i += 128) s += foc[(size_t)NFOC * i + j];
    part[c][j] = s;
    __syncthreads();
    if (threadIdx.x < NFOC) {
        double t = 0.0;
        for (int k = 0; k < 128; k++) t += part[k][threadIdx.x];
        red[threadIdx.x] = t;
        fin[threadIdx.x] = t;
    }
    if (intr0 == nullptr) return;
    __syncthreads();
    if (threadIdx.x == 0) focal_solve(fin, intr0, fail_count, width, height, aspect);
}

__global__ __launch_bounds__(64) void k_init_focal_solve(const double *__restrict__ red, double *__restrict__ intr0,
                                                          int *__restrict__ fail_count, int width, int height,
                                                          double aspect) {
    if (threadIdx.x != 0) return;
    focal_solve(red, intr0, fail_count, width, height, aspect);
}

// ---------------------------------------------------------------------------------------
// A5: planar pose from the homography on normalised points + 6-parameter LM polish
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void polar3(double X[9]) {
    // orthogonal polar factor (== U V^T of the SVD) by Newton iteration X <- (X + X^-T) / 2
    // (quadratic convergence: from the near-orthonormal matrices of a homography decomposition four or five steps reach
    //  the fixed point; the loop stops once a step moves no entry by more than a few ulp)
    for (int it = 0; it < 10; it++) {
        const double c0 = X[4] * X[8] - X[5] * X[7], c1 = X[5] * X[6] - X[3] * X[8], c2 = X[3] * X[7] - X[4] * X[6];
        const double c3 = X[2] * X[7] - X[1] * X[8], c4 = X[0] * X[8] - X[2] * X[6], c5 = X[1] * X[6] - X[0] * X[7];
        const double c6 = X[1] * X[5] - X[2] * X[4], c7 = X[2] * X[3] - X[0] * X[5], c8 = X[0] * X[4] - X[1] * X[3];
        const double idet = 1.0 / (X[0] * c0 + X[1] * c1 + X[2] * c2);
        const double C[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8};   // cofactors: X^-T = C / det
        double moved = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const double xn = 0.5 * (X[k] + C[k] * idet);
            moved = fmax(moved, fabs(xn - X[k]));
            X[k] = xn;
        }
        if (!(moved > 4.0e-16)) break;       // (also leaves on NaN)
    }
}

__device__ __forceinline__ void rodrigues_inv(const double R[9], double r[3]) {
    // [OCV-K cvRodrigues2, matrix -> vector] for an orthonormal R
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1.0) * 0.5;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0.0) { r[0] = r[1] = r[2] = 0.0; return; }
        double t;
        t = (R[0] + 1.0) * 0.5; r[0] = sqrt(fmax(t, 0.0));
        t = (R[4] + 1.0) * 0.5; r[1] = sqrt(fmax(t, 0.0)) * (R[1] < 0.0 ? -1.0 : 1.0);
        t = (R[8] + 1.0) * 0.5; r[2] = sqrt(fmax(t, 0.0)) * (R[2] < 0.0 ? -1.0 : 1.0);
        if (fabs(r[0]) < fabs(r[1]) && fabs(r[0]) < fabs(r[2]) && ((R[5] > 0.0) != (r[1] * r[2] > 0.0))) r[2] = -r[2];
        theta /= sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        r[0] *= theta; r[1] *= theta; r[2] *= theta;
        return;
    }
    const double vth = theta / (2.0 * s);
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

// residual + Jacobian pass of the per-view LM
struct FirstPoint { float m[3]; float2 uv; };      // this lane's point of the first pass (index min(lane, P - 1)), kept in registers

// ||e||^2 of one view (error-only evaluation of a trial point: the lean form of the pose polish)
__device__ double view_error(const float *obj, const float2 *img, int P, int lane, const double rt[6],
                             const double R[9], const double in[NINTR], const FirstPoint &fp) {
    double s = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        if (p < P) {
            const bool first = c0 == 0;
            const double M[3] = {(double)(first ? fp.m[0] : obj[3 * p]), (double)(first ? fp.m[1] : obj[3 * p + 1]),
                                 (double)(first ? fp.m[2] : obj[3 * p + 2])};
            const float2 uv = first ? fp.uv : img[p];
            double eu, ev;
            point_residual(M, (double)uv.x, (double)uv.y, R, rt + 3, in, eu, ev);
            s += eu * eu + ev * ev;
        }
    }
    return wave_sum(s);
}

// JtJ (6x6 packed), Jt e and ||e||^2 of one view: 28 sums over the corners, formed per lane, accumulated over the
// passes and summed over the lanes once (wave_sums_lds)
constexpr int VSUMS = 28;
// ROWS: JtJ[0..5] = row min(lane & 15, 5) of JtJ and JtErr[0] = that row's entry of Jt e (gj_solve_rows' layout) instead of the
// packed matrix and the whole vector in every lane
template <bool ROWS = false>
__device__ double view_normal_eq(const float *obj, const float2 *img, int P, int lane, const double rt[6],
                                 const double R[9], const double K[9], const double in[NINTR], double *scratch,
                                 double *JtJ, double *JtErr, const FirstPoint &fp) {
    double acc[VSUMS];
#pragma unroll
    for (int k = 0; k < VSUMS; k++) acc[k] = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < P;
        const int pc = valid ? p : P - 1;
        const bool first = c0 == 0;
        const double M[3] = {(double)(first ? fp.m[0] : obj[3 * pc]), (double)(first ? fp.m[1] : obj[3 * pc + 1]),
                             (double)(first ? fp.m[2] : obj[3 * pc + 2])};
        const float2 uv = first ? fp.uv : img[pc];
        double mu[NCOL], mv[NCOL];
        JSUB(0);
        point_rows_K(M, (double)uv.x, (double)uv.y, R, K, rt + 3, in, 0.0, mu, mv, valid ? 1.0 : 0.0);   // padding lanes: zero rows
        JSUB(1);
        // (two fused multiply-adds per sum: written out, since contraction follows the source expression in this file and
        //  `acc += a b + c d` would be a product, a fused multiply-add and an addition)
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b <= a; b++) acc[tri(a, b)] = fma(mv[a], mv[b], fma(mu[a], mu[b], acc[tri(a, b)]));
            acc[21 + a] = fma(mv[a], mv[COL_E], fma(mu[a], mu[COL_E], acc[21 + a]));
        }
        acc[27] = fma(mv[COL_E], mv[COL_E], fma(mu[COL_E], mu[COL_E], acc[27]));
    }
    JSUB(2);
    if constexpr (ROWS) {
        const double *res = wave_sums_lds_reduce<VSUMS>(acc, scratch, lane);
        const int la = min(lane & 15, 5);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int hi = max(la, k), lo = min(la, k);
            JtJ[k] = res[hi * (hi + 1) / 2 + lo];
        }
        JtErr[0] = res[21 + la];
        const double e2 = res[27];
        wave_lds_sync();                              // the scratch may be overwritten by the next call
        JSUB(3);
        return e2;
    } else {
        double t[VSUMS];
        wave_sums_lds<VSUMS>(acc, scratch, lane, t);
#pragma unroll
        for (int k = 0; k < 21; k++) JtJ[k] = t[k];
#pragma unroll
        for (int a = 0; a < 6; a++) JtErr[a] = t[21 + a];
        JSUB(3);
        return t[27];
    }
}

// SPEC: trial points of the two polishes by their normal-equations passes (shortest chain per image: up to one image per
// SIMD the launch is as long as its slowest image); !SPEC: CvLevMarq's / LMSolver's own error-only trial evaluations --
// 98 fewer live doubles (what more images than SIMDs want: registers for a second wave per SIMD).
// HMODE: where the normalised homography OpenCV computes before the polish comes from.
//   0  [OCV-K] findHomography on the normalised points: DLT + LMSolver polish (the only form that works without a stored
//      pixel-space homography, or with a distorted intrinsic guess)
//   1  OpenCV's sequence with the DLT taken from k_init_homography: findHomography normalises both point sets per axis (centroid,
//      mean absolute deviation) before the DLT, and the normalised image points (u - cx) / fx are a per-axis affine map of the
//      pixel coordinates, so the DLT solves the SAME system for both and its result on normalised points is K^-1 H_dlt T_c
//      (H_dlt: the pixel-space DLT result before its polish; T_c: the template's centring) up to the float32 rounding of the
//      normalised points.  LMSolver's polish then runs on the normalised points exactly as in form 0.
//   2  the polished pixel-space homography H_pix stands in for the whole second homography: K^-1 H_pix T_c is decomposed
//      directly and the 6-parameter pose polish that follows absorbs the difference.  Views whose pose polish converges end at
//      the same optimum (1e-11); a view that exhausts CvLevMarq's 20 iterations ends where 20 iterations take it from the
//      slightly different start -- opt-in (ccal_set_pose_init), not OpenCV's trajectory for such views
// LDSP = the normalised correspondences are staged in LDS (32 P bytes); false: read through NormPoints (any P).
// FOC: the stages run as one pipeline on a GPU that holds every image -- the initial intrinsics do not exist yet: every
// workgroup reads the exact focal sums k_init_homography has just formed (FocalArgs), solves the 2 x 2 system itself (the same bits
// in all of them) and workgroup 0 stores the intrinsics for the LM and clears the other accumulator buffer.
struct FocalArgs {
    const long long *acc;        // this fit's accumulator buffer
    long long *acc_clear;        // the other one, cleared for the next fit
    double *intr0_out;
    int width, height;
    double aspect;
};
template <bool SPEC, int HMODE, bool LDSP, bool FOC = false>
__device__ __forceinline__ void init_extrinsics_body(const float *__restrict__ obj, const float *__restrict__ img,
                                                     const double *__restrict__ intr0, const double *__restrict__ H_pix,
                                                     double *__restrict__ ext,
                                                     int *__restrict__ fail_count, int N, int P, const FocalArgs &F, const int i) {
    __shared__ __attribute__((aligned(16))) double s_sums[sums_scratch_doubles(HSUMS)];
    const int lane = threadIdx.x;
    (void)N;
    const float2 *img_i = reinterpret_cast<const float2 *>(img) + (size_t)i * P;

    NormPoints pts;
    pts.obj = obj;
    pts.img = img_i;
    bool distorted = false;
    if constexpr (FOC) {
        static_assert(HMODE != 0, "the fused focal solve belongs to the pipeline, where the homographies exist");
        AccPart *s_part = reinterpret_cast<AccPart *>(s_sums);            // (the sums scratch is not in use yet)
        double *s_red = s_sums + 2 * FOC_WORDS;
        foc_acc_read(F.acc, s_part, s_red, lane);
        double red[NFOC];
#pragma unroll
        for (int k = 0; k < NFOC; k++) red[k] = s_red[k];
        int bad;
        focal_from_sums(red, F.width, F.height, F.aspect, pts.in, bad);
        // From here on the kernel must compile to what the stage-by-stage form (intrinsics loaded from memory) compiles to, so that
        // the pipeline and the stages give the same bits: the values are made opaque (the five zeros of the distortion would
        // otherwise be folded into the rows' arithmetic, which changes which products the compiler fuses).
#pragma unroll
        for (int a = 0; a < NINTR; a++) {
            asm volatile("" : "+v"(pts.in[a]));
            if (a >= 4 && pts.in[a] != 0.0) distorted = true;
        }
        wave_lds_sync();
        if (i == 0) {
            if (lane < NINTR) {
                double v = pts.in[0];
#pragma unroll
                for (int a = 1; a < NINTR; a++) v = (lane == a) ? pts.in[a] : v;
                F.intr0_out[lane] = v;
            }
            if (lane == 0) fail_count[1] = bad;
            for (int k = lane; k < FOC_REPL * FOC_WORDS; k += 64) F.acc_clear[k] = 0;
        }
    } else {
#pragma unroll
    for (int a = 0; a < NINTR; a++) {
        pts.in[a] = intr0[a];
        if (a >= 4 && pts.in[a] != 0.0) distorted = true;
    }
    }
    pts.distorted = distorted;
    // template centroid (R_transform = I for a z = 0 board: T_transform = -Mc)
    double c0s = 0.0, c1s = 0.0;
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        if (p < P) { c0s += (double)obj[3 * p]; c1s += (double)obj[3 * p + 1]; }
    }
    pts.cMx = wave_sum(c0s) / P;
    pts.cMy = wave_sum(c1s) / P;

    double H[9], rt[6];
#ifdef CCAL_STAMPS
    const unsigned long long t_begin = memtime_now();
#endif
    ISTAMP(0);
    extern __shared__ __attribute__((aligned(16))) double s_pts[];   // [4][P]
    bool ok;
    if constexpr (HMODE != 0) {
        // H = K^-1 H_pix T_c, scaled to H[8] = 1 (every lane the same bits); a failed pixel-space homography was stored as zeros
        double Hp[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Hp[k] = H_pix[(size_t)18 * i + (HMODE == 1 ? 9 : 0) + k];
        const double ifx = 1.0 / pts.in[0], ify = 1.0 / pts.in[1];
        double G[9];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            G[j] = (Hp[j] - pts.in[2] * Hp[6 + j]) * ifx;
            G[3 + j] = (Hp[3 + j] - pts.in[3] * Hp[6 + j]) * ify;
            G[6 + j] = Hp[6 + j];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) {
            H[a * 3] = G[a * 3];
            H[a * 3 + 1] = G[a * 3 + 1];
            H[a * 3 + 2] = G[a * 3] * pts.cMx + G[a * 3 + 1] * pts.cMy + G[a * 3 + 2];
        }
        ok = Hp[8] != 0.0 && fabs(H[8]) > DBL_EPS;
        const double s = ok ? 1.0 / H[8] : 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) H[k] *= s;
    }
    if constexpr (HMODE != 2) {
        auto run = [&](const auto &cpts) {
            if constexpr (HMODE == 0) {
                ok = wave_homography<SPEC>(cpts, P, lane, s_sums, H);
            } else if (ok && P > 4) {
                homog_refine<SPEC>(cpts, P, lane, s_sums, H);
                H[8] = 1.0;
            }
        };
        if constexpr (LDSP) {
            for (int p = lane; p < P; p += 64) {
                double Mx, My, mx, my;
                pts.get(p, Mx, My, mx, my);
                s_pts[p] = Mx; s_pts[P + p] = My; s_pts[2 * P + p] = mx; s_pts[3 * P + p] = my;
            }
            wave_lds_sync();
            const LdsPoints lpts{s_pts, P};
            CachedPoints<LdsPoints> cpts;
            cpts.init(lpts, lane, P);
            run(cpts);
        } else {
            CachedPoints<NormPoints> cpts;
            cpts.init(pts, lane, P);
            run(cpts);
        }
    }
    if constexpr (HMODE != 0) {
#pragma unroll
        for (int k = 0; k < 9; k++) ok = ok && (fabs(H[k]) < 1.0e300);
    }
    ISTAMP(1);
    if (ok) {
        double h1[3] = {H[0], H[3], H[6]}, h2[3] = {H[1], H[4], H[7]}, h3[3] = {H[2], H[5], H[8]};
        const double n1 = sqrt(h1[0] * h1[0] + h1[1] * h1[1] + h1[2] * h1[2]);
        const double n2 = sqrt(h2[0] * h2[0] + h2[1] * h2[1] + h2[2] * h2[2]);
        double tt[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            h1[a] /= fmax(n1, DBL_EPS);
            h2[a] /= fmax(n2, DBL_EPS);
            tt[a] = h3[a] * 2.0 / fmax(n1 + n2, DBL_EPS);
        }
        h3[0] = h1[1] * h2[2] - h1[2] * h2[1];
        h3[1] = h1[2] * h2[0] - h1[0] * h2[2];
        h3[2] = h1[0] * h2[1] - h1[1] * h2[0];
        double Rm[9] = {h1[0], h2[0], h3[0], h1[1], h2[1], h3[1], h1[2], h2[2], h3[2]};
        polar3(Rm);
        rodrigues_inv(Rm, rt);
        double R[9];
        rodrigues_R(rt, R);
        // t += R * T_transform,  T_transform = -(cMx, cMy, 0)
#pragma unroll
        for (int a = 0; a < 3; a++) rt[3 + a] = tt[a] - (R[a * 3] * pts.cMx + R[a * 3 + 1] * pts.cMy);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) rt[k] = 0.0;
        if (lane == 0) atomicAdd(fail_count, 1);
    }

    // ---- [OCV-K CvLevMarq(6, 2P, (COUNT+EPS, 20, FLT_EPSILON))::update] ----
    // (CCAL_POSE_GJ: JtJ / JtErr hold ONE ROW per lane -- JtJ[0..5], JtErr[0] -- for the lane-parallel solve of the step)
    // (latency form only: in the throughput form -- more images than SIMDs -- the DPP chain is no shorter than the wave-uniform
    //  Cholesky's issue time and C5's pose stage measured 320 -> 330 us with it; C3: 69.8 -> 65.5 us, tools/init_timing.py)
    constexpr bool GJ = CCAL_POSE_GJ != 0 && SPEC;
    constexpr int NJ = GJ ? 6 : 21, NE = GJ ? 1 : 6;
    const int la6 = min(lane & 15, 5);
    double prev[6], JtJ[NJ], JtErr[NE];
#pragma unroll
    for (int k = 0; k < 6; k++) prev[k] = rt[k];
#pragma unroll
    for (int k = 0; k < NE; k++) JtErr[k] = 0.0;
#pragma unroll
    for (int k = 0; k < NJ; k++) JtJ[k] = 0.0;
    ISTAMP(2);
    // lambda = 10^lambdaLg10 as OpenCV forms it, exp(lambdaLg10 * log(10)): tabulated once (lane l holds entry l)
    // instead of one double-precision exp per step
    const double pow10_l = exp((double)((lane < 33 ? lane : 32) - 16) * log(10.0));
    double R[9], K[9];                           // rotation of the current rt and its left Jacobian: one evaluation per
    rodrigues_RK(rt, R, K);                      // new point, shared by the error pass and the Jacobian pass that follows it
    int lambdaLg10 = -3, iters = 0, state = 1;   // 1 STARTED, 2 CALC_J, 3 CHECK_ERR
    int n_j = 0, n_e = 0;
    (void)n_j; (void)n_e;
#ifdef CCAL_STAMPS
    unsigned long long t_step = 0, t_jac = 0, t_err = 0;
#endif
    double errNorm = 0.0, prevErrNorm = 0.0;
    const double *in = pts.in;
    FirstPoint fp;
    {
        const int p0 = lane < P ? lane : P - 1;
        fp.m[0] = obj[3 * p0]; fp.m[1] = obj[3 * p0 + 1]; fp.m[2] = obj[3 * p0 + 2];
        fp.uv = img_i[p0];
    }
    // A trial point is evaluated with the Jacobian pass straight away (CvLevMarq evaluates the error alone and, on
    // acceptance, the Jacobian at the same point in a second pass): steps are accepted almost without exception (2 rejections
    // in the 1000 views of C3), so the trial point's normal equations are the next step's and the error-only pass is saved
    // -- 0.6 k cycles per iteration of the views the launch waits for (the ones that run OpenCV's 20-iteration cap).
    // Same decisions, same points; a rejected trial simply discards its normal equations.  Since the normal equations
    // became per-lane products + one LDS reduce the Jacobian pass is 2.7 k cycles; with round 2's 4.1 k this lost.
    // (tools/init_timing.py, one box: extrinsics stage 95.1 -> 87.5 us on C3, 530 -> 499 us on C5.)
    // After a REJECTED step the re-damped trial points go back to CvLevMarq's own error-only evaluation, and the Jacobian
    // follows on acceptance: a view whose third or fourth step lands on the optimum to rounding rejects the next one six to
    // nine times in a row (lambda has to climb from 1e-6 past 1 before the step shrinks), and a launch of up to one image per
    // SIMD is as long as its slowest image.
    double JtJ_c[NJ], JtErr_c[NE];
    bool have_cand = false, after_reject = false;
    for (int guard = 0; ok && guard < 20 * 24; guard++) {
        bool do_step = false;
        if (state == 1) {
            state = 2;
        } else if (state == 2) {
#pragma unroll
            for (int k = 0; k < 6; k++) prev[k] = rt[k];
            do_step = true;
            prevErrNorm = errNorm;
            state = 3;
        } else {
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
                do_step = true;
                have_cand = false;
                after_reject = true;
            } else {
                after_reject = false;
                lambdaLg10 = max(lambdaLg10 - 1, -16);
                double num = 0.0, den = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) { num += (rt[k] - prev[k]) * (rt[k] - prev[k]); den += prev[k] * prev[k]; }
                // ||param - prevParam|| / ||prevParam|| < FLT_EPSILON, compared squared (no square roots, no division)
                if (++iters >= 20 || num < FLT_EPS * FLT_EPS * den) break;
                prevErrNorm = errNorm;
                if (SPEC && have_cand) {
                    // the accepted point's normal equations are already here: CALC_J's bookkeeping and the step at once
#pragma unroll
                    for (int k = 0; k < NJ; k++) JtJ[k] = JtJ_c[k];
#pragma unroll
                    for (int k = 0; k < NE; k++) JtErr[k] = JtErr_c[k];
#pragma unroll
                    for (int k = 0; k < 6; k++) prev[k] = rt[k];
                    have_cand = false;
                    do_step = true;
                    state = 3;
                } else {
                    state = 2;
                }
            }
        }
#ifdef CCAL_STAMPS
        const unsigned long long ts0 = memtime_now();
#endif
        if (do_step) {
            // CvLevMarq::step: diag *= 1 + lambda; solve; param = prev - delta
            const double lam = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pow10_l), lambdaLg10 + 16),
                                                __builtin_amdgcn_readlane(__double2loint(pow10_l), lambdaLg10 + 16));
            double d[6];
            bool pd;
            if constexpr (GJ) {
                double A6[6];
#pragma unroll
                for (int k = 0; k < 6; k++) A6[k] = (la6 == k) ? JtJ[k] * (1.0 + lam) : JtJ[k];
                pd = gj_solve_rows<6>(A6, JtErr[0], la6, d);
            } else {
                double A[21];
#pragma unroll
                for (int k = 0; k < 21; k++) A[k] = JtJ[k];
#pragma unroll
                for (int k = 0; k < 6; k++) d[k] = JtErr[k];
                damp_diag<6>(A, lam);
                double ip6[6];
                pd = chol_packed<6>(A, ip6);
                fwd_subst<6>(A, ip6, d);
                bwd_subst<6>(A, ip6, d);
            }
#pragma unroll
            for (int k = 0; k < 6; k++) rt[k] = prev[k] - (pd ? d[k] : 0.0);
            rodrigues_RK(rt, R, K);
        }
#ifdef CCAL_STAMPS
        const unsigned long long ts1 = memtime_now();
        if (do_step) t_step += ts1 - ts0;
#endif
        if (state == 2) { errNorm = sqrt(view_normal_eq<GJ>(obj, img_i, P, lane, rt, R, K, in, s_sums, JtJ, JtErr, fp)); n_j++; }
        else if (SPEC && !after_reject) { errNorm = sqrt(view_normal_eq<GJ>(obj, img_i, P, lane, rt, R, K, in, s_sums, JtJ_c, JtErr_c, fp)); have_cand = true; n_e++; }
        else { errNorm = sqrt(view_error(obj, img_i, P, lane, rt, R, in, fp)); n_e++; }
#ifdef CCAL_STAMPS
        const unsigned long long ts2 = memtime_now();
        if (state == 2) t_jac += ts2 - ts1; else t_err += ts2 - ts1;
#endif
    }
    ICOUNT(8, t_step); ICOUNT(9, t_jac); ICOUNT(10, t_err);
    ISTAMP(3);
    ICOUNT(4, n_j); ICOUNT(5, n_e); ICOUNT(6, iters);
#ifdef CCAL_STAMPS
    IBLK(0, memtime_now() - t_begin); IBLK(2, iters); IBLK(3, n_e);
#endif
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 6; k++) finite = finite && (fabs(rt[k]) < 1.0e300);
    if (ok && !finite && lane == 0) atomicAdd(fail_count, 1);
    if (i == 0 && lane == 0) fail_count[2] = 1;      // "fresh initialisation results": the next solve's first launch reports them to the host
    if (lane < 6) {
        double v = rt[0];
#pragma unroll
        for (int k = 1; k < 6; k++) v = (lane == k) ? rt[k] : v;
        ext[(size_t)i * EXT + lane] = v;
    }
}

template <bool SPEC, int HMODE, bool LDSP, bool FOC = false>
__global__ __launch_bounds__(64, SPEC ? 1 : 2) void k_init_extrinsics(const float *__restrict__ obj, const float *__restrict__ img,
                                                         const double *__restrict__ intr0, const double *__restrict__ H_pix,
                                                         double *__restrict__ ext,
                                                         int *__restrict__ fail_count, int N, int P, const FocalArgs F) {
    init_extrinsics_body<SPEC, HMODE, LDSP, FOC>(obj, img, intr0, H_pix, ext, fail_count, N, P, F, (int)blockIdx.x);
}

struct ExtArgs {
    const float *obj, *img;
    const double *intr0, *H_pix;
    double *ext;
    int *fail_count;
    int N, P;
    FocalArgs F;
};
template <bool SPEC, int HMODE, bool LDSP, bool FOC>
__global__ __launch_bounds__(64, SPEC ? 1 : 2) void k_init_extrinsics_multi(const ExtArgs *__restrict__ table, const int *__restrict__ map,
                                                                             const int *__restrict__ first) {
    const int fit = map[blockIdx.x];
    const ExtArgs a = table[fit];
    init_extrinsics_body<SPEC, HMODE, LDSP, FOC>(a.obj, a.img, a.intr0, a.H_pix, a.ext, a.fail_count, a.N, a.P, a.F, (int)blockIdx.x - first[fit]);
}

// ---------------------------------------------------------------------------------------
static_assert(sizeof(HomArgs) <= sizeof(MergeRecord::args) && sizeof(ExtArgs) <= sizeof(MergeRecord::args), "a recorded launch keeps its arguments by value");
template <class Args>
static void merge_add(MergeRecorder *r, int kernel, int key, int grid, size_t dyn, const Args &a) {
    MergeRecord m;
    m.kernel = kernel; m.key = key; m.grid = grid; m.block = 64; m.dyn_lds = dyn;
    memcpy(m.args, &a, sizeof a);
    m.args_bytes = sizeof a;
    r->recs.push_back(m);
}

// key = SPEC
hipError_t launch_init_homography_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, int grid) {
    const HomArgs *t = static_cast<const HomArgs *>(table);
    if (key & 1) hipLaunchKernelGGL(k_init_homography_multi<true>, dim3(grid), dim3(64), 0, stream, t, map, first);
    else hipLaunchKernelGGL(k_init_homography_multi<false>, dim3(grid), dim3(64), 0, stream, t, map, first);
    return hipGetLastError();
}
// key = SPEC | HMODE << 1 (pipeline form with the staged correspondences only: FOC, LDSP)
hipError_t launch_init_extrinsics_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, int grid, size_t dyn_lds) {
    const ExtArgs *t = static_cast<const ExtArgs *>(table);
    const int spec = key & 1, hmode = key >> 1;
    if (hmode == 2 && spec) hipLaunchKernelGGL((k_init_extrinsics_multi<true, 2, true, true>), dim3(grid), dim3(64), dyn_lds, stream, t, map, first);
    else if (hmode == 2) hipLaunchKernelGGL((k_init_extrinsics_multi<false, 2, true, true>), dim3(grid), dim3(64), dyn_lds, stream, t, map, first);
    else if (hmode == 1 && spec) hipLaunchKernelGGL((k_init_extrinsics_multi<true, 1, true, true>), dim3(grid), dim3(64), dyn_lds, stream, t, map, first);
    else if (hmode == 1) hipLaunchKernelGGL((k_init_extrinsics_multi<false, 1, true, true>), dim3(grid), dim3(64), dyn_lds, stream, t, map, first);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_init_homography(hipStream_t stream, const float *obj, const float *img, double *H_out,
                                  double *foc_partials, long long *foc_acc, int *fail0, int N, int P, int width, int height, bool spec) {
    if (MergeRecorder *r = merge_recorder()) {
        const HomArgs a{obj, img, H_out, foc_partials, foc_acc, fail0, N, P, width, height};
        merge_add(r, MERGE_HOMOGRAPHY, spec ? 1 : 0, N, 0, a);
        return hipSuccess;
    }
    // up to one image per SIMD the launch is as long as its slowest image: trial points by the normal-equations pass;
    // beyond that it is throughput: the leaner form that keeps two waves per SIMD
    if (spec)
        hipLaunchKernelGGL(k_init_homography<true>, dim3(N), dim3(64), 0, stream, obj, img, H_out, foc_partials, foc_acc, fail0, N, P, width, height);
    else
        hipLaunchKernelGGL(k_init_homography<false>, dim3(N), dim3(64), 0, stream, obj, img, H_out, foc_partials, foc_acc, fail0, N, P, width, height);
    return hipGetLastError();
}
hipError_t launch_init_focal_acc(hipStream_t stream, const long long *acc, long long *acc_clear, double *intr0, int *fail_count,
                                 int width, int height, double aspect) {
    hipLaunchKernelGGL(k_init_focal_acc, dim3(1), dim3(64), 0, stream, acc, acc_clear, (double *)nullptr, intr0, fail_count, width, height, aspect);
    return hipGetLastError();
}
hipError_t launch_init_focal_acc_row(hipStream_t stream, const long long *acc, long long *acc_clear, double *red_out) {
    hipLaunchKernelGGL(k_init_focal_acc, dim3(1), dim3(64), 0, stream, acc, acc_clear, red_out, (double *)nullptr, (int *)nullptr, 0, 0, 0.0);
    return hipGetLastError();
}
hipError_t launch_init_focal_reduce(hipStream_t stream, const double *foc_partials, double *foc_red, int N) {
    hipLaunchKernelGGL(k_init_focal_reduce, dim3(1), dim3(1024), 0, stream, foc_partials, foc_red, N, (double *)nullptr, (int *)nullptr,
                       0, 0, 0.0);
    return hipGetLastError();
}
hipError_t launch_init_focal_reduce_solve(hipStream_t stream, const double *foc_partials, double *foc_red, int N, double *intr0,
                                          int *fail_count, int width, int height, double aspect) {
    hipLaunchKernelGGL(k_init_focal_reduce, dim3(1), dim3(1024), 0, stream, foc_partials, foc_red, N, intr0, fail_count, width,
                       height, aspect);
    return hipGetLastError();
}
hipError_t launch_init_focal_solve(hipStream_t stream, const double *foc_red, double *intr0, int *fail_count,
                                   int width, int height, double aspect) {
    hipLaunchKernelGGL(k_init_focal_solve, dim3(1), dim3(64), 0, stream, foc_red, intr0, fail_count, width, height,
                       aspect);
    return hipGetLastError();
}
// One image per SIMD or fewer (decided by the caller from the GLOBAL image count, so that every shard of a fit and the
// unsharded fit run the same arithmetic): the launch is as long as its slowest image -> speculative form; beyond that it is
// throughput -> the lean form that keeps two waves per SIMD.
int init_simd_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] > 0) return cached[dev];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    return cached[dev] = 4 * cus;
}

hipError_t launch_init_extrinsics(hipStream_t stream, const float *obj, const float *img, const double *intr0, const double *H_pix,
                                  int hmode, bool spec, double *ext, int *fail_count, int N, int P, const InitFocal *focal) {
    // the staged correspondences (32 P bytes) next to the static scratch must fit a workgroup's 64 KB of LDS; bigger boards
    // read them through NormPoints (lean form only: that instantiation exists for throughput anyway)
    const size_t lds_pts = sizeof(double) * 4 * (size_t)P;
    const bool ldsp = lds_pts + sizeof(double) * sums_scratch_doubles(HSUMS) + 256 <= 64 * 1024;
    if (H_pix == nullptr) hmode = 0;
    if (!ldsp) spec = false;
    FocalArgs F{};
    const bool foc = focal != nullptr;
    if (foc) {
        if (hmode == 0) return hipErrorInvalidValue;      // (the pipeline form needs the intrinsics stage's homographies)
        F.acc = focal->acc; F.acc_clear = focal->acc_clear; F.intr0_out = focal->intr0_out;
        F.width = focal->width; F.height = focal->height; F.aspect = focal->aspect;
    }
    if (MergeRecorder *r = merge_recorder()) {
        if (!foc || !ldsp) { r->unsupported = true; return hipErrorInvalidValue; }   // (merged form: the pipeline with staged correspondences)
        const ExtArgs a{obj, img, intr0, H_pix, ext, fail_count, N, P, F};
        merge_add(r, MERGE_EXTRINSICS, (spec ? 1 : 0) | hmode << 1, N, hmode == 2 ? 0 : lds_pts, a);
        return hipSuccess;
    }
    auto go = [&](auto kernel, size_t dyn) {
        hipLaunchKernelGGL(kernel, dim3(N), dim3(64), dyn, stream, obj, img, intr0, H_pix, ext, fail_count, N, P, F);
    };
    if (foc) {                              // stages as one pipeline: the focal solve in the prologue
        if (hmode == 2) {
            if (spec) go(k_init_extrinsics<true, 2, true, true>, 0); else go(k_init_extrinsics<false, 2, true, true>, 0);
        } else {
            if (!ldsp) go(k_init_extrinsics<false, 1, false, true>, 0);
            else if (spec) go(k_init_extrinsics<true, 1, true, true>, lds_pts);
            else go(k_init_extrinsics<false, 1, true, true>, lds_pts);
        }
    } else if (hmode == 2) {                // no pass over normalised points at all
        if (spec) go(k_init_extrinsics<true, 2, true>, 0); else go(k_init_extrinsics<false, 2, true>, 0);
    } else if (hmode == 1) {
        if (!ldsp) go(k_init_extrinsics<false, 1, false>, 0);
        else if (spec) go(k_init_extrinsics<true, 1, true>, lds_pts);
        else go(k_init_extrinsics<false, 1, true>, lds_pts);
    } else {
        if (!ldsp) go(k_init_extrinsics<false, 0, false>, 0);
        else if (spec) go(k_init_extrinsics<true, 0, true>, lds_pts);
        else go(k_init_extrinsics<false, 0, true>, lds_pts);
    }
    return hipGetLastError();
}

}  // namespace ccal
