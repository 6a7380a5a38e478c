// ccal_lm.hip -- the global Levenberg-Marquardt loop of calibrateCamera on gfx950.
//
// Replaces the loop OpenCV runs behind the reference's single FFI call
// (reference src/detect_fit.jl:47; algorithm: SURVEY.md rows A6-A9, Appendix A.3/A.6).
//
// Two kernels per LM trial, decisions made on the device (the host only reads `done`):
//
//   k_lm_sweep   one 64-lane wavefront per image, one lane per corner.
//                  1. back-substitute this image's extrinsic step from the factor stored by the
//                     previous sweep (delta_i = L^-T (Y_h - Y_B delta_int)) -> candidate rvec/tvec
//                  2. per corner: residual + the two 16-wide rows of [Je | Ji | e]  (FP64)
//                  3. rows -> LDS (padded, conflict-free) -> v_mfma_f64_16x16x4_f64:
//                     G = [Je|Ji|e]^T [Je|Ji|e]  (16x16; holds JtJ blocks, Jt e and ||e||^2)
//                  4. damp + Cholesky of the 6x6 extrinsic block, Y = L^-1 [B^T | h], Schur
//                     contribution Y^T Y, workgroup partial sums -> partials[block][112]
//   k_lm_solve   one workgroup: fixed-order sum of the partials (deterministic), then one
//                lane advances the CvLevMarq state machine: accept / reject, lambda
//                schedule, stop rule, and the 9x9 damped Schur system for the next step.
//
// A rejected step costs one extra "re-Schur" pass over the stored blocks (no projection).
// Multi-GPU: the 112-double reduced system is all-reduced between the reduce and solve
// halves of k_lm_solve (host hook), every rank then advances the same state.
#include <hip/hip_runtime.h>

#include "ccal_kernels.hpp"
#include "ccal_math.hpp"
#include "ccal_state.hpp"
#include "ccal_wave.hpp"

namespace ccal {

constexpr int WAVES = 4;             // images per workgroup

// packed-lower index tables
__device__ const unsigned char TRI6_A[21] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5};
__device__ const unsigned char TRI6_B[21] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5};
__device__ const unsigned char TRI10_A[55] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 6, 6, 6, 6, 6, 6, 6,
                                             7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9};
__device__ const unsigned char TRI10_B[55] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6,
                                             0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9};

// ---------------------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_sweep(const float *__restrict__ obj,
                                                          const float *__restrict__ img,
                                                          double *__restrict__ ext,
                                                          double *__restrict__ blk,
                                                          const LmState *__restrict__ st,
                                                          double *__restrict__ partials,
                                                          double *__restrict__ gram_out, int N, int P) {
    __shared__ __attribute__((aligned(16))) double s_tile[WAVES * TILE_D];
    __shared__ __attribute__((aligned(16))) double s_G[WAVES * 256];
    __shared__ __attribute__((aligned(16))) double s_Y[WAVES * 64];     // Y (60) | pad
    __shared__ __attribute__((aligned(16))) double s_L[WAVES * 32];     // L (21) | inv (6) | pd
    __shared__ __attribute__((aligned(16))) double s_red[WAVES * NRED];

    if (st->done) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img_i = blockIdx.x * WAVES + wave;
    const bool active = img_i < N;              // wave-uniform
    const int ii = active ? img_i : N - 1;      // clamped: idle waves mirror the last image, store nothing
    const int mode = st->mode;
    const int par = st->parity;

    double *tile = s_tile + wave * TILE_D;
    double *Gs = s_G + wave * 256;
    double *Ys = s_Y + wave * 64;
    double *Ls = s_L + wave * 32;
    double *red = s_red + wave * NRED;

    double dnum = 0.0, dden = 0.0;
    const int cand = (mode == MODE_EVAL) ? 1 - par : par;   // buffer this sweep evaluates / writes

    if (mode != MODE_RESCHUR) {
        // ---- 1. candidate extrinsics (wave-uniform arithmetic) ----
        double e6[6];
        const double *ep = ext + ((size_t)par * N + ii) * EXT;
#pragma unroll
        for (int k = 0; k < 6; k++) e6[k] = ep[k];
        double in[NINTR];
        if (mode == MODE_EVAL) {
            // delta_i = C_l^-1 (h - B^T delta_int) = L^-T (Y_h - Y_B delta_int), factor from the last sweep
            const double *bp = blk + ((size_t)par * N + ii) * BLK;
            double L[21], inv[6], rhs[6];
#pragma unroll
            for (int k = 0; k < 21; k++) L[k] = bp[BLK_L + k];
#pragma unroll
            for (int k = 0; k < 6; k++) { inv[k] = bp[BLK_INV + k]; rhs[k] = bp[BLK_Y + 54 + k]; }
#pragma unroll
            for (int a = 0; a < NINTR; a++) {
                const double d = st->dint[a];
#pragma unroll
                for (int k = 0; k < 6; k++) rhs[k] -= bp[BLK_Y + a * 6 + k] * d;
            }
            bwd_subst<6>(L, inv, rhs);
            const bool ok = bp[BLK_PD] != 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const double d = ok ? rhs[k] : 0.0;   // degenerate view: leave its pose unchanged
                dden += e6[k] * e6[k];
                dnum += d * d;
                e6[k] -= d;
            }
            if (active && lane < 6) ext[((size_t)cand * N + ii) * EXT + lane] = e6[lane];
#pragma unroll
            for (int a = 0; a < NINTR; a++) in[a] = st->intr_cand[a];
        } else {
#pragma unroll
            for (int a = 0; a < NINTR; a++) in[a] = st->intr_prev[a];
        }

        // ---- 2./3. rows -> LDS -> f64 MFMA Gram ----
        double R[9], dR[27];
        rodrigues(e6, R, dR);
        const double aspect = st->aspect;
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const float2 *img2 = reinterpret_cast<const float2 *>(img) + (size_t)ii * P;
        for (int c0 = 0; c0 < P; c0 += 64) {
            const int p = c0 + lane;
            const bool valid = p < P;
            const int pc = valid ? p : P - 1;
            const double M[3] = {(double)obj[3 * pc], (double)obj[3 * pc + 1], (double)obj[3 * pc + 2]};
            const float2 uv = img2[pc];             // coalesced 8 B per lane
            double mu[NCOL], mv[NCOL];
            point_rows(M, (double)uv.x, (double)uv.y, R, dR, e6 + 3, in, aspect, mu, mv);
            tile_store_rows(tile, lane, mu, mv, valid);
            __syncthreads();
            tile_gram(tile, lane, min(64, P - c0), acc0, acc1);
            __syncthreads();
        }
        gram_to_lds(Gs, lane, acc0, acc1);
    } else {
        // ---- re-Schur pass: rebuild the C / B / h entries of G from the stored raw blocks ----
#pragma unroll
        for (int r = 0; r < 4; r++) Gs[lane + 64 * r] = 0.0;
        __syncthreads();
        const double *bp = blk + ((size_t)par * N + ii) * BLK;
        for (int o = lane; o < BLK_RAW; o += 64) {
            const double v = bp[o];
            if (o < BLK_B) Gs[TRI6_A[o] * 16 + TRI6_B[o]] = v;
            else if (o < BLK_H) Gs[((o - BLK_B) % 6) * 16 + 6 + (o - BLK_B) / 6] = v;
            else Gs[(o - BLK_H) * 16 + COL_E] = v;
        }
    }
    __syncthreads();

    if (gram_out != nullptr && active) {
#pragma unroll
        for (int r = 0; r < 4; r++) gram_out[(size_t)ii * 256 + lane + 64 * r] = Gs[lane + 64 * r];
    }

    // ---- 4. damped factor and Schur contribution at lam_next ----
    double C[21], inv[6];
#pragma unroll
    for (int k = 0; k < 21; k++) C[k] = Gs[TRI6_A[k] * 16 + TRI6_B[k]];   // LDS broadcast reads
    damp_diag<6>(C, st->lam_next);
    const bool pd = chol_packed<6>(C, inv);
    {
        // lane j < 10 owns column j of [B^T | h] and forward-substitutes it: Y_j = L^-1 col_j
        const int j = lane < 10 ? lane : 9;
        const int col = j < 9 ? 6 + j : COL_E;
        double y[6];
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] = Gs[k * 16 + col];
        fwd_subst<6>(C, inv, y);
        if (lane < 10) {
#pragma unroll
            for (int k = 0; k < 6; k++) Ys[j * 6 + k] = y[k];
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 21; k++) Ls[k] = C[k];
#pragma unroll
            for (int k = 0; k < 6; k++) Ls[21 + k] = inv[k];
            Ls[27] = pd ? 1.0 : 0.0;
        }
    }
    __syncthreads();

    // ---- keep this image's blocks: raw part for re-damping, factor part for back-substitution ----
    if (active) {
        double *bo = blk + ((size_t)cand * N + ii) * BLK;
        for (int o = lane; o < BLK_USED; o += 64) {
            double v;
            if (o < BLK_RAW) {
                if (mode == MODE_RESCHUR) continue;      // raw blocks unchanged
                if (o < BLK_B) v = Gs[TRI6_A[o] * 16 + TRI6_B[o]];
                else if (o < BLK_H) v = Gs[((o - BLK_B) % 6) * 16 + 6 + (o - BLK_B) / 6];
                else v = Gs[(o - BLK_H) * 16 + COL_E];
            } else if (o < BLK_Y) {
                v = (o < BLK_L + 28) ? Ls[o - BLK_L] : 0.0;
            } else {
                v = Ys[o - BLK_Y];
            }
            bo[o] = v;
        }
    }

    const double err_i = Gs[COL_E * 16 + COL_E];
    const bool finite = (err_i == err_i) && (fabs(err_i) < 1.0e300);
    for (int o = lane; o < NRED; o += 64) {
        double v = 0.0;
        if (o < RED_SC) {                           // A, g straight from G
            if (mode != MODE_RESCHUR) {
                if (o < RED_G) v = Gs[(6 + TRI10_A[o]) * 16 + 6 + TRI10_B[o]];
                else v = Gs[(6 + (o - RED_G)) * 16 + COL_E];
            }
        } else if (o < RED_ERR) {                   // SC, sg from Y^T Y
            int a, b;
            if (o < RED_SG) { a = TRI10_A[o - RED_SC]; b = TRI10_B[o - RED_SC]; }
            else { a = o - RED_SG; b = 9; }
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) acc += Ys[a * 6 + k] * Ys[b * 6 + k];
            v = pd ? acc : 0.0;
        } else if (o == RED_ERR) v = (mode != MODE_RESCHUR) ? err_i : 0.0;
        else if (o == RED_DNUM) v = dnum;
        else if (o == RED_DDEN) v = dden;
        else v = (pd && finite) ? 0.0 : 1.0;
        red[o] = active ? v : 0.0;
    }
    __syncthreads();
    if (tid < NRED) {
        double s = s_red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; w++) s += s_red[w * NRED + tid];   // fixed order
        partials[(size_t)blockIdx.x * NRED + tid] = s;
    }
}

// ---------------------------------------------------------------------------------------
// reduce + CvLevMarq state machine + 9x9 Schur solve
// ---------------------------------------------------------------------------------------
__device__ void lm_trace(LmState *st, TraceRec *trace, double err, double rel, int accepted) {
    if (trace != nullptr && st->n_trace < st->max_trace) {
        TraceRec &t = trace[st->n_trace];
        t.err = err; t.rel_step = rel; t.accepted = accepted;
        t.lambda_lg10 = st->lambda_lg10; t.iters = st->iters; t.pad = 0;
    }
    st->n_trace++;
}

__device__ __forceinline__ double pow10_lg(const LmState *st, int lg) { return st->pow10[lg + 16]; }

// [OCV-K CvLevMarq::updateAlt], SURVEY.md A.6, fused with the next step()
__device__ void lm_advance(LmState *st, const double *rs, TraceRec *trace) {
    const int mode = st->mode;
    const double err = rs[RED_ERR];
    if (st->trials < MAX_KINDS) st->kinds[st->trials] = 1 + mode;
    st->trials++;
    if (mode == MODE_INIT) {
        st->sweeps++;
        st->prev_err = err;
        st->n_bad = (int)rs[RED_BAD];
        for (int k = 0; k < 45; k++) st->A_prev[k] = rs[RED_A + k];
        for (int k = 0; k < 9; k++) st->g_prev[k] = rs[RED_G + k];
        lm_trace(st, trace, err, -1.0, 2);
        if (!(fabs(err) < 1.0e300)) { st->status = -4; st->done = 1; return; }
    } else if (mode == MODE_EVAL) {
        st->sweeps++;
        if (err > st->prev_err && ++st->lambda_lg10 <= 16) {
            // rejected: same blocks, larger lambda
            st->rejected++;
            st->mode = MODE_RESCHUR;
            st->lg_partials = st->lambda_lg10;
            st->lam_next = pow10_lg(st, st->lambda_lg10);
            lm_trace(st, trace, err, -1.0, 0);
            return;
        }
        st->lambda_lg10 = max(st->lambda_lg10 - 1, -16);
        st->iters++;
        double dnum = rs[RED_DNUM], dden = rs[RED_DDEN];
        for (int a = 0; a < 9; a++) {
            const double d = st->intr_cand[a] - st->intr_prev[a];
            dnum += d * d;
            dden += st->intr_prev[a] * st->intr_prev[a];
            st->intr_prev[a] = st->intr_cand[a];
        }
        const double rel = sqrt(dnum) / sqrt(dden);
        st->rel_step = rel;
        st->parity ^= 1;
        st->prev_err = err;
        st->n_bad = (int)rs[RED_BAD];
        for (int k = 0; k < 45; k++) st->A_prev[k] = rs[RED_A + k];
        for (int k = 0; k < 9; k++) st->g_prev[k] = rs[RED_G + k];
        lm_trace(st, trace, err, rel, 1);
        if (!(fabs(err) < 1.0e300)) { st->status = -4; st->done = 1; return; }
        if (st->iters >= st->max_iter || rel < st->eps) {
            if (!(rel < st->eps)) st->status = 1;   // CCAL_W_MAX_ITER
            st->done = 1;
            return;
        }
        if (st->lg_partials != st->lambda_lg10) {   // lambda saturated: partials are for another lambda
            st->mode = MODE_RESCHUR;
            st->lg_partials = st->lambda_lg10;
            st->lam_next = pow10_lg(st, st->lambda_lg10);
            return;
        }
    } else {
        st->reschur++;
    }

    // ---- step(): damped Schur system  S = A_l - SC,  rhs = g - sg   ----
    const double lam = pow10_lg(st, st->lambda_lg10);
    double S[45], rhs[9], inv[9];
    bool freep[9];
#pragma unroll
    for (int a = 0; a < 9; a++) {
        const double Aaa = st->A_prev[tri(a, a)];
        freep[a] = st->mask[a] != 0 && Aaa > 0.0;   // fixed or unobservable parameter: delta = 0
    }
#pragma unroll
    for (int a = 0; a < 9; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) {
            double v = st->A_prev[tri(a, b)] - rs[RED_SC + tri(a, b)];
            if (a == b) v = st->A_prev[tri(a, a)] * (1.0 + lam) - rs[RED_SC + tri(a, a)];
            if (!(freep[a] && freep[b])) v = (a == b) ? 1.0 : 0.0;
            S[tri(a, b)] = v;
        }
        rhs[a] = freep[a] ? st->g_prev[a] - rs[RED_SG + a] : 0.0;
    }
    const bool ok = chol_packed<9>(S, inv);
    if (!ok) {
        // damped system not positive definite (degenerate geometry): damp harder
        if (++st->lambda_lg10 <= 16) {
            st->mode = MODE_RESCHUR;
            st->lg_partials = st->lambda_lg10;
            st->lam_next = pow10_lg(st, st->lambda_lg10);
        } else {
            st->status = -3;
            st->done = 1;
        }
        return;
    }
    fwd_subst<9>(S, inv, rhs);
    bwd_subst<9>(S, inv, rhs);
#pragma unroll
    for (int a = 0; a < 9; a++) {
        const double d = freep[a] ? rhs[a] : 0.0;
        st->dint[a] = d;
        st->intr_cand[a] = st->intr_prev[a] - d;
    }
    if (st->aspect > 0.0) st->intr_cand[0] = st->aspect * st->intr_cand[1];
    st->lam_step = lam;
    const int lgn = max(st->lambda_lg10 - 1, -16);
    st->lg_partials = lgn;
    st->lam_next = pow10_lg(st, lgn);
    st->mode = MODE_EVAL;
}

constexpr int SOLVE_THREADS = 1024;
constexpr int SOLVE_SLICES = SOLVE_THREADS / 128;
constexpr int SOLVE_UNROLL = 32;
constexpr int STATE_WORDS = sizeof(LmState) / 8;
static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as 8-byte words");

// One workgroup.  The partial sums are added by 8 slices x 112 columns with independent
// (pipelined) loads and combined in a fixed order; the state is staged through LDS so the
// serial state machine never waits on global memory.  When the state machine terminates, the
// final state is also written straight into the caller's pinned host mirror (no copy node).
__global__ __launch_bounds__(SOLVE_THREADS) void k_lm_solve(LmState *st, const double *__restrict__ partials,
                                                             double *red, TraceRec *trace, LmState *host_result,
                                                             int nblocks, int do_reduce, int do_solve) {
    __shared__ double part[SOLVE_SLICES][128];
    __shared__ double rs[NRED];
    __shared__ LmState ls;
    if (st->done) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < STATE_WORDS; i += SOLVE_THREADS)
        reinterpret_cast<double *>(&ls)[i] = reinterpret_cast<const double *>(st)[i];
    if (do_reduce) {
        // slice c adds blocks c, c + 8, ... of column j: SOLVE_UNROLL loads are put in flight
        // together (one L2 round trip per group instead of one per block), then added in order
        const int j = tid & 127, c = tid >> 7;
        double s = 0.0;
        if (j < NRED) {
            for (int b0 = c; b0 < nblocks; b0 += SOLVE_SLICES * SOLVE_UNROLL) {
                double v[SOLVE_UNROLL];
#pragma unroll
                for (int k = 0; k < SOLVE_UNROLL; k++) {
                    const int b = b0 + k * SOLVE_SLICES;
                    v[k] = b < nblocks ? partials[(size_t)b * NRED + j] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < SOLVE_UNROLL; k++) s += v[k];
            }
        }
        part[c][j] = s;
        __syncthreads();
        if (tid < NRED) {
            double t = part[0][tid];
#pragma unroll
            for (int k = 1; k < SOLVE_SLICES; k++) t += part[k][tid];   // fixed order
            red[tid] = t;
            rs[tid] = t;
        }
    } else if (tid < NRED) {
        rs[tid] = red[tid];
    }
    __syncthreads();
    if (!do_solve) return;
    if (tid == 0) lm_advance(&ls, rs, trace);
    __syncthreads();
    const bool finished = ls.done != 0 && host_result != nullptr;
    for (int i = tid; i < STATE_WORDS; i += SOLVE_THREADS) {
        const double w = reinterpret_cast<const double *>(&ls)[i];
        reinterpret_cast<double *>(st)[i] = w;
        if (finished) reinterpret_cast<double *>(host_result)[i] = w;
    }
}

// ---------------------------------------------------------------------------------------
// begin: (re)build the LM state on the device and restore the initial poses
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_begin(LmState *st, LmInit init, const double *__restrict__ intr0,
                                                   const double *__restrict__ ext0, double *__restrict__ ext) {
    const int tid = threadIdx.x;
    const size_t n = (size_t)init.N_local * EXT;
    if (ext != ext0)
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) ext[i] = ext0[i];
    if (blockIdx.x != 0) return;
    __shared__ LmState ls;
    for (int i = tid; i < STATE_WORDS; i += 256) reinterpret_cast<double *>(&ls)[i] = 0.0;
    __syncthreads();
    // lambda = 10^lambdaLg10 as OpenCV forms it: exp(lambdaLg10 * log(10))
    if (tid < 33) ls.pow10[tid] = exp((double)(tid - 16) * log(10.0));
    __syncthreads();
    if (tid == 0) {
        ls.mode = MODE_INIT;
        ls.lambda_lg10 = -3;
        ls.lg_partials = -3;
        ls.lam_next = ls.pow10[13];
        ls.lam_step = ls.lam_next;
        ls.max_iter = init.max_iter;
        ls.eps = init.eps;
        ls.max_trace = init.max_trace;
        ls.P = init.P;
        ls.N_local = init.N_local;
        ls.nblocks = init.nblocks;
        ls.N_global = init.N_global;
        for (int a = 0; a < 9; a++) { ls.mask[a] = init.mask[a]; ls.intr_prev[a] = intr0[a]; }
        ls.aspect = init.aspect_from_guess ? intr0[0] / intr0[1] : init.aspect;
        // [OCV-K cvCalibrateCamera2Internal]: p1 = p2 = 0 under ZERO_TANGENT_DIST; fx = aspect * fy
        if (init.zero_tangent) ls.intr_prev[6] = ls.intr_prev[7] = 0.0;
        if (ls.aspect > 0.0) ls.intr_prev[0] = ls.aspect * ls.intr_prev[1];
        for (int a = 0; a < 9; a++) ls.intr_cand[a] = ls.intr_prev[a];
    }
    __syncthreads();
    for (int i = tid; i < STATE_WORDS; i += 256)
        reinterpret_cast<double *>(st)[i] = reinterpret_cast<const double *>(&ls)[i];
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
int lm_sweep_blocks(int N) { return (N + WAVES - 1) / WAVES; }

hipError_t launch_lm_begin(hipStream_t stream, LmState *st, const LmInit &init, const double *intr0, const double *ext0,
                           double *ext) {
    const int blocks = (int)((((size_t)init.N_local * EXT) + 255) / 256);
    hipLaunchKernelGGL(k_lm_begin, dim3(blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks)), dim3(256), 0, stream, st, init,
                       intr0, ext0, ext);
    return hipGetLastError();
}

hipError_t launch_lm_sweep(hipStream_t stream, const float *obj, const float *img, double *ext, double *blk,
                           const LmState *st, double *partials, double *gram_out, int N, int P) {
    hipLaunchKernelGGL(k_lm_sweep, dim3(lm_sweep_blocks(N)), dim3(WAVES * 64), 0, stream, obj, img, ext, blk, st,
                       partials, gram_out, N, P);
    return hipGetLastError();
}

hipError_t launch_lm_solve(hipStream_t stream, LmState *st, const double *partials, double *red, TraceRec *trace,
                           LmState *host_result, int nblocks, int do_reduce, int do_solve) {
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(SOLVE_THREADS), 0, stream, st, partials, red, trace, host_result,
                       nblocks, do_reduce, do_solve);
    return hipGetLastError();
}

}  // namespace ccal
