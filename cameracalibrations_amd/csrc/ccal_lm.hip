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
//                     contribution Y^T Y, workgroup partial sums -> partials[block][68]
//   k_lm_solve   one workgroup: fixed-order sum of the partials (deterministic), then one
//                lane advances the CvLevMarq state machine: accept / reject, lambda
//                schedule, stop rule, and the 9x9 damped Schur system for the next step.
//
// A rejected step costs one extra sweep of the accepted point with the larger lambda.
// Multi-GPU: the 68-double reduced system is all-reduced between the reduce and solve
// halves of k_lm_solve (host hook), every rank then advances the same state.
#include <hip/hip_runtime.h>

#include "ccal_kernels.hpp"
#include "ccal_math.hpp"
#include "ccal_state.hpp"
#include "ccal_wave.hpp"

namespace ccal {

constexpr int WAVES = 4;             // images per workgroup

// Diagnostic build only (-DCCAL_STAMPS): s_memtime stamps of one wave at the phase boundaries
// of the sweep, read back through ccal_debug_stamps().  No stamp executes in the product build.
#ifdef CCAL_STAMPS
__device__ unsigned long long g_stamps[32];
#define STAMP(i)                                                                              \
    do {                                                                                      \
        if (blockIdx.x == 7 && tid == 0) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_stamps[i] = t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)
#define STAMP2(i)                                                                             \
    do {                                                                                      \
        if (threadIdx.x == 0) {                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_stamps[16 + (i)] = t_;                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP2(i) do { } while (0)
#endif

// packed-lower decode o -> (a, b), a >= b, for o < 55 (no memory table: arithmetic in the
// shadow of the first global loads)
__device__ __forceinline__ void tri_decode(int o, int &a, int &b) {
    a = (int)((sqrtf(8.0f * (float)o + 1.0f) - 1.0f) * 0.5f);
    if ((a + 1) * (a + 2) / 2 <= o) a++;
    if (a * (a + 1) / 2 > o) a--;
    b = o - a * (a + 1) / 2;
}

// wave-private LDS carve-up (doubles).  The Gram tile is dead once the MFMAs have issued, so
// everything after it re-uses the same 17 KB: 4 waves x 17 KB + 3.5 KB = 73 KB per workgroup.
constexpr int W_G = 0;        // 16 x 16 Gram matrix
constexpr int W_HEAD = 256;   // 72-double block head (prologue staging, then the new head)
constexpr int W_Y = 328;      // Y[j][k] = (L^-1 col_j)[k], 10 x 6
static_assert(W_Y + 60 <= TILE_D, "wave scratch exceeds the tile");

// ---------------------------------------------------------------------------------------
// sweep
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_sweep(const float *__restrict__ obj,
                                                          const float *__restrict__ img,
                                                          double *__restrict__ blk,
                                                          const LmState *__restrict__ st,
                                                          double *__restrict__ partials,
                                                          double *__restrict__ gram_out, int N, int P) {
    __shared__ __attribute__((aligned(16))) double s_wave[WAVES * TILE_D];
    __shared__ __attribute__((aligned(16))) double s_red[WAVES * NRED];

    const int tid = threadIdx.x;
    STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img_i = blockIdx.x * WAVES + wave;
    const bool active = img_i < N;              // wave-uniform
    const int ii = active ? img_i : N - 1;      // clamped: idle waves mirror the last image, store nothing

    // ---- 0. every load that does not depend on the LM state is issued up front: both parities
    //         of the block head (rvec,tvec | U | pd) and the first 64 corners ----
    const double *b0p = blk + (size_t)ii * BLK, *b1p = blk + ((size_t)N + ii) * BLK;
    const double h0a = b0p[lane], h1a = b1p[lane];
    const double h0b = lane < 8 ? b0p[64 + lane] : 0.0, h1b = lane < 8 ? b1p[64 + lane] : 0.0;
    const float2 *img2 = reinterpret_cast<const float2 *>(img) + (size_t)ii * P;
    const int pc0 = lane < P ? lane : P - 1;
    float2 uv = img2[pc0];                      // coalesced 8 B per lane
    float m0 = obj[3 * pc0], m1 = obj[3 * pc0 + 1], m2 = obj[3 * pc0 + 2];
    int oa, ob;                                 // (a, b) of the packed S0 entry this lane assembles
    tri_decode(lane < 45 ? lane : 0, oa, ob);

    if (st->done) return;
    STAMP(1);
    const int mode = st->mode;
    const int par = st->parity;
    const int cand = (mode == MODE_EVAL) ? 1 - par : par;   // buffer this sweep evaluates / writes

    double *W = s_wave + wave * TILE_D;
    double *Gs = W + W_G, *Hs = W + W_HEAD, *Ys = W + W_Y;
    double *red = s_red + wave * NRED;

    // ---- 1. block head of the accepted point -> wave-uniform registers ----
    Hs[lane] = par ? h1a : h0a;
    if (lane < 8) Hs[64 + lane] = par ? h1b : h0b;
    wave_lds_sync();
    double e6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) e6[k] = Hs[BLK_EXT + k];    // LDS broadcast reads
    double dnum = 0.0, dden = 0.0;
    double in[NINTR];
    if (mode == MODE_EVAL) {
        // delta_i = C_l^-1 (h - B^T delta_int) = u_h - U_B delta_int, U kept by the last sweep
        double d[6];
#pragma unroll
        for (int k = 0; k < 6; k++) d[k] = Hs[BLK_U + 54 + k];
#pragma unroll
        for (int a = 0; a < NINTR; a++) {
            const double da = st->dint[a];
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] -= Hs[BLK_U + a * 6 + k] * da;
        }
        const bool ok = Hs[BLK_PD] != 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double dk = ok ? d[k] : 0.0;    // degenerate view: leave its pose unchanged
            dden += e6[k] * e6[k];
            dnum += dk * dk;
            e6[k] -= dk;
        }
#pragma unroll
        for (int a = 0; a < NINTR; a++) in[a] = st->intr_cand[a];
    } else {
        // MODE_INIT / MODE_RESCHUR: (re-)evaluate the accepted point itself
#pragma unroll
        for (int a = 0; a < NINTR; a++) in[a] = st->intr_prev[a];
    }
    wave_lds_sync();                              // head consumed: the tile may overwrite it

    // ---- 2./3. rows -> LDS -> f64 MFMA Gram ----
    STAMP(2);
    double R[9], K[9];
    rodrigues_RK(e6, R, K);
    STAMP(3);
    const double aspect = st->aspect;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < P;
        if (c0 > 0) {
            const int pc = valid ? p : P - 1;
            uv = img2[pc];
            m0 = obj[3 * pc]; m1 = obj[3 * pc + 1]; m2 = obj[3 * pc + 2];
        }
        const double M[3] = {(double)m0, (double)m1, (double)m2};
        double mu[NCOL], mv[NCOL];
        point_rows_K(M, (double)uv.x, (double)uv.y, R, K, e6 + 3, in, aspect, mu, mv);
        STAMP(4);
        tile_store_rows(W, lane, mu, mv, valid);
        wave_lds_sync();
        STAMP(5);
        tile_gram(W, lane, min(64, P - c0), acc0, acc1);
        wave_lds_sync();                          // operands are in registers: the tile is free
    }
    gram_to_lds(Gs, lane, acc0, acc1);
    STAMP(6);
    wave_lds_sync();

    if (gram_out != nullptr && active) {
#pragma unroll
        for (int r = 0; r < 4; r++) gram_out[(size_t)ii * 256 + lane + 64 * r] = Gs[lane + 64 * r];
    }

    // ---- 4. damped factor, U = C_l^-1 [B^T | h] and Y = L^-1 [B^T | h] at lam_next ----
    STAMP(7);
    double C[21], inv[6];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) C[tri(a, b)] = Gs[a * 16 + b];   // LDS broadcast reads
    damp_diag<6>(C, st->lam_next);
    const bool pd = chol_packed<6>(C, inv);
    {
        // lane j < 10 owns column j of [B^T | h]
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
        bwd_subst<6>(C, inv, y);
        if (lane < 10) {
#pragma unroll
            for (int k = 0; k < 6; k++) Hs[BLK_U + j * 6 + k] = y[k];
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) Hs[BLK_EXT + k] = e6[k];
            Hs[6] = Hs[7] = 0.0;
            Hs[BLK_PD] = pd ? 1.0 : 0.0;
            Hs[69] = Hs[70] = Hs[71] = 0.0;
        }
    }
    wave_lds_sync();

    // ---- keep this image's block for the next sweep's prologue ----
    STAMP(8);
    if (active) {
        double *bo = blk + ((size_t)cand * N + ii) * BLK;
        bo[lane] = Hs[lane];
        if (lane < 8) bo[64 + lane] = Hs[64 + lane];
    }

    // ---- this image's contributions to the reduced system: S0 | dA | gs | err dnum dden bad ----
    STAMP(9);
    const double err_i = Gs[COL_E * 16 + COL_E];
    const bool finite = (err_i == err_i) && (fabs(err_i) < 1.0e300);
    {
        double v;
        if (lane < RED_GS) {
            // lanes 0..44: A(a,b) - Y_a . Y_b;  45..53: A(a,a)
            const int a = lane < RED_DA ? oa : lane - RED_DA, b = lane < RED_DA ? ob : lane - RED_DA;
            v = Gs[(6 + a) * 16 + 6 + b];
            if (lane < RED_DA) {
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) acc += Ys[a * 6 + k] * Ys[b * 6 + k];
                v -= pd ? acc : 0.0;
            }
        } else if (lane < RED_ERR) {
            const int a = lane - RED_GS;                  // g(a) - Y_a . Y_h
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) acc += Ys[a * 6 + k] * Ys[54 + k];
            v = Gs[(6 + a) * 16 + COL_E] - (pd ? acc : 0.0);
        } else {
            v = err_i;                                    // lane 63
        }
        red[lane] = active ? v : 0.0;
        if (lane < 4) {
            const double w = lane == 0 ? dnum : (lane == 1 ? dden : (lane == 2 ? ((pd && finite) ? 0.0 : 1.0) : 0.0));
            red[64 + lane] = active ? w : 0.0;
        }
    }
    __syncthreads();
    if (tid < NRED) {
        double s = s_red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; w++) s += s_red[w * NRED + tid];   // fixed order
        partials[(size_t)blockIdx.x * NRED + tid] = s;
    }
    STAMP(10);
}

#ifdef CCAL_STAMPS
hipError_t debug_read_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
#endif

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
        lm_trace(st, trace, err, -1.0, 2);
        if (!(fabs(err) < 1.0e300)) { st->status = -4; st->done = 1; return; }
    } else if (mode == MODE_EVAL) {
        st->sweeps++;
        if (err > st->prev_err && ++st->lambda_lg10 <= 16) {
            // rejected: sweep the accepted point again with the larger lambda
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
        st->reschur++;                              // fresh system of the accepted point at the new lambda
    }

    // ---- step(): damped Schur system  S = A_l - SC = S0 + lambda diag(A),  rhs = g - sg  (row A8) ----
    const double lam = pow10_lg(st, st->lambda_lg10);
    double S[45], rhs[9], inv[9];
    bool freep[9];
#pragma unroll
    for (int a = 0; a < 9; a++) freep[a] = st->mask[a] != 0 && rs[RED_DA + a] > 0.0;   // fixed / unobservable: delta = 0
#pragma unroll
    for (int a = 0; a < 9; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) {
            double v = rs[RED_S + tri(a, b)];
            if (a == b) v += lam * rs[RED_DA + a];
            if (!(freep[a] && freep[b])) v = (a == b) ? 1.0 : 0.0;
            S[tri(a, b)] = v;
        }
        rhs[a] = freep[a] ? rs[RED_GS + a] : 0.0;
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
constexpr int SOLVE_COLS = 64 + 8;                  // >= NRED, partial rows are read as 72-wide
constexpr int SOLVE_SLICES = 14;                    // 14 x 72 = 1008 threads reduce
constexpr int SOLVE_UNROLL = 24;
static_assert(NRED <= SOLVE_COLS && SOLVE_SLICES * SOLVE_COLS <= SOLVE_THREADS, "reduce geometry");
constexpr int STATE_WORDS = sizeof(LmState) / 8;
static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as 8-byte words");

// One workgroup.  The partial sums are added by 8 slices x 112 columns with independent
// (pipelined) loads and combined in a fixed order; the state is staged through LDS so the
// serial state machine never waits on global memory.  When the state machine terminates, the
// final state is also written straight into the caller's pinned host mirror (no copy node).
__global__ __launch_bounds__(SOLVE_THREADS) void k_lm_solve(LmState *st, const double *__restrict__ partials,
                                                             double *red, TraceRec *trace, LmState *host_result,
                                                             unsigned long long *host_progress, int nblocks,
                                                             int do_reduce, int do_solve) {
    __shared__ double part[SOLVE_SLICES][SOLVE_COLS];
    __shared__ double rs[NRED];
    __shared__ LmState ls;
    STAMP2(0);
    if (st->done) return;
    STAMP2(1);
    const int tid = threadIdx.x;
    for (int i = tid; i < STATE_WORDS; i += SOLVE_THREADS)
        reinterpret_cast<double *>(&ls)[i] = reinterpret_cast<const double *>(st)[i];
    if (do_reduce) {
        // slice c adds blocks c, c + 14, ... of column j: SOLVE_UNROLL loads are put in flight
        // together (one L2 round trip per group instead of one per block), then added in order
        const int j = tid % SOLVE_COLS, c = tid / SOLVE_COLS;
        double s = 0.0;
        if (j < NRED && c < SOLVE_SLICES) {
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
        if (c < SOLVE_SLICES) part[c][j] = s;
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
    STAMP2(2);
    if (!do_solve) return;
    if (tid == 0) lm_advance(&ls, rs, trace);
    STAMP2(3);
    __syncthreads();
    for (int i = tid; i < STATE_WORDS; i += SOLVE_THREADS)
        reinterpret_cast<double *>(st)[i] = reinterpret_cast<const double *>(&ls)[i];
    if (host_progress != nullptr && tid < 64) {
        // wave 0 alone talks to the host: the state header when the solve terminates, then the
        // progress word (trials << 32) | done that the host spins on, released at system scope
        static_assert(HOST_WORDS <= 64 && HOST_WORDS <= STATE_WORDS, "host mirror is written by one wave");
        if (ls.done != 0 && host_result != nullptr && tid < HOST_WORDS)
            reinterpret_cast<double *>(host_result)[tid] = reinterpret_cast<const double *>(&ls)[tid];
        if (tid == 0)
            __hip_atomic_store(host_progress, ((unsigned long long)(unsigned)ls.trials << 32) | (ls.done ? 1ull : 0ull),
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    STAMP2(4);
}

// ---------------------------------------------------------------------------------------
// begin: (re)build the LM state on the device and restore the initial poses
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_begin(LmState *st, LmInit init, const double *__restrict__ intr0,
                                                   const double *ext0, int ext0_stride, double *blk) {
    const int tid = threadIdx.x;
    // initial rvec,tvec of image i -> head of its parity-0 block (the source may be that block)
    const size_t n = (size_t)init.N_local * 8;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < n; i += (size_t)gridDim.x * 256) {
        const size_t im = i >> 3, k = i & 7;
        const double v = k < 6 ? ext0[im * ext0_stride + k] : 0.0;
        blk[im * BLK + BLK_EXT + k] = v;
    }
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
                           int ext0_stride, double *blk) {
    const int blocks = (int)((((size_t)init.N_local * 8) + 255) / 256);
    hipLaunchKernelGGL(k_lm_begin, dim3(blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks)), dim3(256), 0, stream, st, init,
                       intr0, ext0, ext0_stride, blk);
    return hipGetLastError();
}

hipError_t launch_lm_sweep(hipStream_t stream, const float *obj, const float *img, double *blk, const LmState *st,
                           double *partials, double *gram_out, int N, int P) {
    hipLaunchKernelGGL(k_lm_sweep, dim3(lm_sweep_blocks(N)), dim3(WAVES * 64), 0, stream, obj, img, blk, st, partials,
                       gram_out, N, P);
    return hipGetLastError();
}

hipError_t launch_lm_solve(hipStream_t stream, LmState *st, const double *partials, double *red, TraceRec *trace,
                           LmState *host_result, unsigned long long *host_progress, int nblocks, int do_reduce,
                           int do_solve) {
    hipLaunchKernelGGL(k_lm_solve, dim3(1), dim3(SOLVE_THREADS), 0, stream, st, partials, red, trace, host_result,
                       host_progress, nblocks, do_reduce, do_solve);
    return hipGetLastError();
}

}  // namespace ccal
