// ccal_lm.hip -- the global Levenberg-Marquardt loop of calibrateCamera on gfx950.
//
// Replaces the loop OpenCV runs behind the reference's single FFI call
// (reference src/detect_fit.jl:47; algorithm: SURVEY.md rows A6-A9, Appendix A.3/A.6).
//
// ONE kernel per LM trial, decisions made on the device (the host only reads `done`):
//
//   k_lm_trial   prologue, every workgroup redundantly: fixed-order sum of the previous trial's
//                workgroup partials (deterministic: identical bits in every workgroup), then one lane
//                advances the CvLevMarq state machine on an LDS copy of the state: accept / reject,
//                lambda schedule, stop rule, and the 9x9 damped Schur system for this trial's step.
//                Workgroup 0 persists the state (double-buffered by trial parity) and reports to the
//                host.  Then the sweep: one 64-lane wavefront per image, one lane per corner.
//                  1. back-substitute this image's extrinsic step from the factor stored by the
//                     previous sweep (delta_i = L^-T (Y_h - Y_B delta_int)) -> candidate rvec/tvec
//                  2. per corner: residual + the two 16-wide rows of [Je | Ji | e]  (FP64)
//                  3. rows -> LDS (padded, conflict-free) -> v_mfma_f64_16x16x4_f64:
//                     G = [Je|Ji|e]^T [Je|Ji|e]  (16x16; holds JtJ blocks, Jt e and ||e||^2)
//                  4. damp + Cholesky of the 6x6 extrinsic block, Y = L^-1 [B^T | h], Schur
//                     contribution Y^T Y, workgroup partial sums -> partials[block][68]
//   k_lm_reduce  only when the partials are too many to be summed by every workgroup (large N) or a
//                collective library sums them across GPUs (RCCL / host hook): partial rows -> G rows.
//   k_lm_exchange  multi-GPU with many workgroups: partial rows -> one row -> direct xGMI exchange.
//
// A solve of T trials is T + 1 launches of k_lm_trial (the last one runs the prologue only).
// A rejected step costs one extra sweep of the accepted point with the larger lambda.
// Multi-GPU (images sharded): every rank sums its own rows, the 68-double row is summed over the ranks
// and every rank advances the same state.  Default transport: the direct row exchange fused into the
// prologue (k_lm_trial<true>, ccal_p2p.hpp) -- no extra launch; alternatives: k_lm_reduce -> RCCL or
// a host hook -> next k_lm_trial.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstring>

#include "ccal_kernels.hpp"
#include "ccal_math.hpp"
#include "ccal_p2p_dev.hpp"
#include "ccal_state.hpp"
#include "ccal_wave.hpp"

namespace ccal {

static_assert(P2P_MAX_DOUBLES == NRED, "the exchanged row is one reduced system");

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
        if (blockIdx.x == 7 && threadIdx.x == 0) {                                                               \
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

// [OCV-K CvLevMarq::updateAlt], SURVEY.md A.6, fused with the next step().
// Every workgroup runs this on its own LDS copy of the state and reaches the same decisions;
// BOOK = true (workgroup 0 only, whose copy is persisted) also keeps the books the host reads:
// trace, per-trial kinds, counters, the reported relative step.
template <bool BOOK>
__device__ __forceinline__ void lm_advance(LmState *__restrict__ st, const double *__restrict__ rs, TraceRec *trace) {
    // Everything the decision needs is pulled out of LDS in ONE burst of independent reads (the
    // reduced row, the scalars of the state, both intrinsic vectors): one lane runs this, so every
    // dependent LDS round trip in the middle of the logic would be paid in full.
    double r[NRED];
#pragma unroll
    for (int o = 0; o < NRED; o++) r[o] = rs[o];
    const int mode = st->mode;
    const int lam_lg0 = st->lambda_lg10;
    int lam_lg = lam_lg0;
    const int lg_part = st->lg_partials, iters0 = st->iters, max_iter = st->max_iter, trials0 = st->trials;
    const double prev_err = st->prev_err, eps = st->eps, aspect = st->aspect;
    double ip[9], ic[9];
    bool freep[9];
#pragma unroll
    for (int a = 0; a < 9; a++) {
        ip[a] = st->intr_prev[a];
        ic[a] = st->intr_cand[a];
        freep[a] = st->mask[a] != 0 && r[RED_DA + a] > 0.0;   // fixed / unobservable: delta = 0
    }
    // the three lambdas a decision can ask for (dynamic table reads issued with the burst)
    const double lam_up = pow10_lg(st, min(lam_lg0 + 1, 16)), lam_dn = pow10_lg(st, max(lam_lg0 - 1, -16)),
                 lam_dn2 = pow10_lg(st, max(lam_lg0 - 2, -16)), lam_same = pow10_lg(st, lam_lg0);
    auto pow10_sel = [&](int lg) -> double {       // 10^lg from the pre-read values when it is one of them
        return lg == lam_lg0 ? lam_same : (lg == lam_lg0 - 1 ? lam_dn : (lg == lam_lg0 - 2 ? lam_dn2 : (lg == lam_lg0 + 1 ? lam_up : pow10_lg(st, lg))));
    };
    const double err = r[RED_ERR];
    if (BOOK) {
        if (trials0 < MAX_KINDS) st->kinds[trials0] = 1 + mode;
        st->n_bad = (int)r[RED_BAD];
    }
    st->trials = trials0 + 1;
    double lam = lam_same;                          // lambda of the step() below
    if (mode == MODE_INIT) {
        if (BOOK) { st->sweeps++; lm_trace(st, trace, err, -1.0, 2); }
        st->prev_err = err;
        if (!(fabs(err) < 1.0e300)) { st->status = -4; st->done = 1; return; }
    } else if (mode == MODE_EVAL) {
        if (BOOK) st->sweeps++;
        if (err > prev_err && lam_lg + 1 <= 16) {
            // rejected: sweep the accepted point again with the larger lambda
            lam_lg++;
            st->lambda_lg10 = lam_lg;
            if (BOOK) { st->rejected++; lm_trace(st, trace, err, -1.0, 0); }
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg;
            st->lam_next = lam_up;
            return;
        }
        if (err > prev_err) lam_lg++;               // lambda already at 10^16: the ++ happened, OpenCV accepts the step
        lam_lg = max(lam_lg - 1, -16);
        st->lambda_lg10 = lam_lg;
        st->iters = iters0 + 1;
        double dnum = r[RED_DNUM], dden = r[RED_DDEN];
#pragma unroll
        for (int a = 0; a < 9; a++) {
            const double d = ic[a] - ip[a];
            dnum += d * d;
            dden += ip[a] * ip[a];
            ip[a] = ic[a];
            st->intr_prev[a] = ic[a];
        }
        // stop rule ||param - prevParam|| / ||prevParam|| < eps, decided without sqrt / divide
        const bool small = dnum < eps * eps * dden;
        st->parity ^= 1;
        st->prev_err = err;
        if (BOOK) {
            const double rel = sqrt(dnum / dden);
            st->rel_step = rel;
            lm_trace(st, trace, err, rel, 1);
        }
        if (!(fabs(err) < 1.0e300)) { st->status = -4; st->done = 1; return; }
        if (iters0 + 1 >= max_iter || small) {
            if (!small) st->status = 1;             // CCAL_W_MAX_ITER
            st->done = 1;
            return;
        }
        lam = pow10_sel(lam_lg);
        if (lg_part != lam_lg) {                    // lambda saturated: partials are for another lambda
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg;
            st->lam_next = lam;
            return;
        }
    } else {
        if (BOOK) st->reschur++;                    // fresh system of the accepted point at the new lambda
    }

    // ---- step(): damped Schur system  S = A_l - SC = S0 + lambda diag(A),  rhs = g - sg  (row A8) ----
    STAMP2(0);
    double S[45], rhs[9], inv[9];
#pragma unroll
    for (int a = 0; a < 9; a++) {
#pragma unroll
        for (int b = 0; b <= a; b++) {
            double v = r[RED_S + tri(a, b)];
            if (a == b) v += lam * r[RED_DA + a];
            if (!(freep[a] && freep[b])) v = (a == b) ? 1.0 : 0.0;
            S[tri(a, b)] = v;
        }
        rhs[a] = freep[a] ? r[RED_GS + a] : 0.0;
    }
    STAMP2(1);
    const bool ok = chol_packed<9>(S, inv);
    STAMP2(2);
    if (!ok) {
        // damped system not positive definite (degenerate geometry): damp harder
        if (lam_lg + 1 <= 16) {
            st->lambda_lg10 = lam_lg + 1;
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg + 1;
            st->lam_next = pow10_sel(lam_lg + 1);
        } else {
            st->lambda_lg10 = lam_lg + 1;
            st->status = -3;
            st->done = 1;
        }
        return;
    }
    fwd_subst<9>(S, inv, rhs);
    bwd_subst<9>(S, inv, rhs);
    STAMP2(3);
    double cand[9];
#pragma unroll
    for (int a = 0; a < 9; a++) {
        const double d = freep[a] ? rhs[a] : 0.0;
        st->dint[a] = d;
        cand[a] = ip[a] - d;
    }
    if (aspect > 0.0) cand[0] = aspect * cand[1];
#pragma unroll
    for (int a = 0; a < 9; a++) st->intr_cand[a] = cand[a];
    st->lam_step = lam;
    const int lgn = max(lam_lg - 1, -16);
    st->lg_partials = lgn;
    st->lam_next = pow10_sel(lgn);
    st->mode = MODE_EVAL;
}

// wave-private LDS carve-up (doubles).  The Gram tile is dead once the MFMAs have issued, so
// everything after it re-uses the same 17 KB.
constexpr int W_G = 0;        // 16 x 16 Gram matrix
constexpr int W_HEAD = 256;   // 72-double block head (prologue staging, then the new head)
constexpr int W_Y = 328;      // Y[j][k] = (L^-1 col_j)[k], 10 x 6
static_assert(W_Y + 60 <= TILE_D, "wave scratch exceeds the tile");

constexpr int STATE_WORDS = sizeof(LmState) / 8;
static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as 8-byte words");
static_assert(HOST_WORDS <= 64 && HOST_WORDS <= STATE_WORDS, "host mirror is written by one wave");
constexpr int RED_PAIRS = NRED / 2;                 // partial rows are read as 34 double2
constexpr int RED_SLICES = (WAVES * 64) / RED_PAIRS; // 7 slices of rows per workgroup
constexpr int RED_UNROLL = 40;                      // double2 loads in flight per thread (280 rows in one trip)

// Fixed-order sum of `n_rows` rows of NRED doubles by one 256-thread workgroup -> rs[NRED].
// Slice c adds rows c, c + 7, ...; the first RED_UNROLL rows of every slice are fetched by
// reduce_issue() as independent 16-byte loads BEFORE anything waits on memory, so the prologue
// costs one memory round trip for up to 7 * RED_UNROLL rows; slices are combined in order.
// Same instruction order on the same data in every workgroup: identical bits everywhere.
struct RedRegs { double2 v[RED_UNROLL]; };

__device__ __forceinline__ void reduce_issue(const double *__restrict__ rows, int n_rows, int tid, RedRegs &q) {
    const int j2 = tid % RED_PAIRS, c = tid / RED_PAIRS;
    const double2 *src = reinterpret_cast<const double2 *>(rows) + j2;
#pragma unroll
    for (int k = 0; k < RED_UNROLL; k++) {
        const int r = c + k * RED_SLICES;
        q.v[k] = (c < RED_SLICES && r < n_rows) ? src[(size_t)r * RED_PAIRS] : make_double2(0.0, 0.0);
    }
}

__device__ __forceinline__ void reduce_finish(const double *__restrict__ rows, int n_rows, double *part, double *rs,
                                              int tid, const RedRegs &q) {
    const int j2 = tid % RED_PAIRS, c = tid / RED_PAIRS;
    if (c < RED_SLICES) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < RED_UNROLL; k++) { s0 += q.v[k].x; s1 += q.v[k].y; }
        const double2 *src = reinterpret_cast<const double2 *>(rows) + j2;
        for (int r0 = c + RED_UNROLL * RED_SLICES; r0 < n_rows; r0 += RED_SLICES * 8) {   // beyond the first group
            double2 v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int r = r0 + k * RED_SLICES;
                v[k] = r < n_rows ? src[(size_t)r * RED_PAIRS] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int k = 0; k < 8; k++) { s0 += v[k].x; s1 += v[k].y; }
        }
        part[c * NRED + 2 * j2] = s0;
        part[c * NRED + 2 * j2 + 1] = s1;
    }
    __syncthreads();
    if (tid < NRED) {
        double t = part[tid];
#pragma unroll
        for (int k = 1; k < RED_SLICES; k++) t += part[k * NRED + tid];
        rs[tid] = t;
    }
    __syncthreads();
}

// Everything one launch needs, by value (a solve is a graph of these launches).
struct TrialArgs {
    const float *obj, *img;
    double *blk;
    LmState *st;                  // [2], double-buffered by trial parity: S_k lives in st[k & 1]
    double *partials;             // [2][nblocks][NRED], this trial writes buffer k & 1
    const double *rows_in;        // reduction input of the prologue (previous partials, or reduced rows)
    int n_in;
    TraceRec *trace;
    LmState *host_result;
    unsigned long long *host_progress;
    const double *intr0, *ext0;   // start values (first trial only)
    int ext0_stride;
    LmInit init;
    double *gram_out;
    int N, P, k;
    int first;                    // k == 0: build S_0 instead of advancing
    int final_only;               // prologue only (detect termination, report to the host)
    int xchg_fused;               // image-sharded fit: the prologue exchanges the local row with the peer GPUs
    P2PDev p2p;                   // used by the XCHG instantiation only
};

// S_0: the state OpenCV's loop starts from
__device__ void lm_boot(LmState *ls, const LmInit &init, const double *intr0, int tid) {
    for (int i = tid; i < STATE_WORDS; i += WAVES * 64) reinterpret_cast<double *>(ls)[i] = 0.0;
    __syncthreads();
    // lambda = 10^lambdaLg10 as OpenCV forms it: exp(lambdaLg10 * log(10))
    if (tid < 33) ls->pow10[tid] = exp((double)(tid - 16) * log(10.0));
    __syncthreads();
    if (tid == 0) {
        ls->mode = MODE_INIT;
        ls->lambda_lg10 = -3;
        ls->lg_partials = -3;
        ls->lam_next = ls->pow10[13];
        ls->lam_step = ls->lam_next;
        ls->max_iter = init.max_iter;
        ls->eps = init.eps;
        ls->max_trace = init.max_trace;
        ls->P = init.P;
        ls->N_local = init.N_local;
        ls->nblocks = init.nblocks;
        ls->N_global = init.N_global;
        for (int a = 0; a < 9; a++) { ls->mask[a] = init.mask[a]; ls->intr_prev[a] = intr0[a]; }
        ls->aspect = init.aspect_from_guess ? intr0[0] / intr0[1] : init.aspect;
        // [OCV-K cvCalibrateCamera2Internal]: p1 = p2 = 0 under ZERO_TANGENT_DIST; fx = aspect * fy
        if (init.zero_tangent) ls->intr_prev[6] = ls->intr_prev[7] = 0.0;
        if (ls->aspect > 0.0) ls->intr_prev[0] = ls->aspect * ls->intr_prev[1];
        for (int a = 0; a < 9; a++) ls->intr_cand[a] = ls->intr_prev[a];
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// one LM trial
// ---------------------------------------------------------------------------------------
// XCHG = image-sharded fit over the GPUs of one node (ccal_p2p.hpp): between the local row sum and
// the LM advance the prologue stores the row into the peers' inboxes (workgroup 0) and adds the
// peers' rows from the local inbox in rank order (every workgroup) -- the whole all-reduce, with
// no launch of its own.  The first launch of a solve takes the next sequence number.
template <bool XCHG>
__global__ __launch_bounds__(WAVES * 64) void k_lm_trial(const TrialArgs A) {
    __shared__ __attribute__((aligned(16))) double s_wave[WAVES * TILE_D];
    __shared__ __attribute__((aligned(16))) double s_red[WAVES * NRED];
    __shared__ __attribute__((aligned(16))) double s_part[RED_SLICES * NRED];
    __shared__ __attribute__((aligned(16))) double s_rs[NRED];
    __shared__ LmState ls;
    double *s_all = s_wave;       // XCHG: the peers' rows are staged in the (not yet used) tile space
    static_assert(P2P_MAX_RANKS * NRED <= WAVES * TILE_D, "exchange staging exceeds the tile space");

    const int tid = threadIdx.x;
    STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N, P = A.P;
    const int img_i = blockIdx.x * WAVES + wave;
    const bool active = img_i < N;              // wave-uniform
    const int ii = active ? img_i : N - 1;      // clamped: idle waves mirror the last image, store nothing

    // ---- 0. loads that do not depend on the LM state are issued up front: both parities of the
    //         block head (rvec,tvec | U | pd) -- or the start pose -- and the first 64 corners ----
    double h0a = 0.0, h1a = 0.0, h0b = 0.0, h1b = 0.0;
    const float2 *img2 = reinterpret_cast<const float2 *>(A.img) + (size_t)ii * P;
    const int pc0 = lane < P ? lane : P - 1;
    float2 uv = make_float2(0.f, 0.f);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (!A.final_only) {
        if (A.first) {
            h0a = lane < 6 ? A.ext0[(size_t)ii * A.ext0_stride + lane] : 0.0;
        } else {
            const double *b0p = A.blk + (size_t)ii * BLK, *b1p = A.blk + ((size_t)N + ii) * BLK;
            h0a = b0p[lane]; h1a = b1p[lane];
            if (lane < 8) { h0b = b0p[64 + lane]; h1b = b1p[64 + lane]; }
        }
        uv = img2[pc0];                         // coalesced 8 B per lane
        m0 = A.obj[3 * pc0]; m1 = A.obj[3 * pc0 + 1]; m2 = A.obj[3 * pc0 + 2];
    }
    int oa, ob;                                 // (a, b) of the packed S0 entry this lane assembles
    tri_decode(lane < 45 ? lane : 0, oa, ob);

    // ---- 1. the state of this trial, S_k, in LDS (every workgroup computes the same bits) ----
    unsigned int seq = 0;
    if constexpr (XCHG) seq = __hip_atomic_load(A.p2p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (A.first) {
        lm_boot(&ls, A.init, A.intr0, tid);
        // no other workgroup of this launch reads the sequence number; the launches after it see seq + 1
        if constexpr (XCHG)
            if (blockIdx.x == 0 && tid == 0)
                __hip_atomic_store(A.p2p.epoch, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const LmState *prev = A.st + ((A.k - 1) & 1);
        RedRegs q;
        reduce_issue(A.rows_in, A.n_in, tid, q);   // in flight together with the state and the heads
        for (int i = tid; i < STATE_WORDS; i += WAVES * 64)
            reinterpret_cast<double *>(&ls)[i] = reinterpret_cast<const double *>(prev)[i];
        __syncthreads();
        STAMP(11);
        if (ls.done) {
            // solve already terminated: surplus launch.  The terminal state is carried along, so every
            // later launch of the batch sees it too (none of them sweeps or exchanges again).
            if (blockIdx.x == 0) {
                LmState *cur = A.st + (A.k & 1);
                for (int i = tid; i < STATE_WORDS; i += WAVES * 64)
                    reinterpret_cast<double *>(cur)[i] = reinterpret_cast<const double *>(&ls)[i];
            }
            return;
        }
        reduce_finish(A.rows_in, A.n_in, s_part, s_rs, tid, q);
        if constexpr (XCHG)
            if (A.xchg_fused)
                p2p_exchange_rows(A.p2p, p2p_tag(seq, (unsigned)A.k), p2p_slot(seq, (unsigned)A.k), s_rs, s_all, NRED, tid,
                                  blockIdx.x == 0);
        STAMP(12);
        if (tid == 0) {
            if (blockIdx.x == 0) lm_advance<true>(&ls, s_rs, A.trace);
            else lm_advance<false>(&ls, s_rs, nullptr);
        }
        __syncthreads();
    }
    STAMP(1);
    if (blockIdx.x == 0) {
        LmState *cur = A.st + (A.k & 1);
        for (int i = tid; i < STATE_WORDS; i += WAVES * 64)
            reinterpret_cast<double *>(cur)[i] = reinterpret_cast<const double *>(&ls)[i];
        if (A.host_progress != nullptr && tid < 64) {
            // wave 0 alone talks to the host: the state header when the solve terminates, then the
            // progress word (trials << 32) | done that the host spins on, released at system scope
            if (ls.done != 0 && A.host_result != nullptr && tid < HOST_WORDS)
                reinterpret_cast<double *>(A.host_result)[tid] = reinterpret_cast<const double *>(&ls)[tid];
            if (tid == 0)
                __hip_atomic_store(A.host_progress,
                                   ((unsigned long long)(unsigned)ls.trials << 32) | (ls.done ? 1ull : 0ull),
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (ls.done || A.final_only) return;

    const int mode = ls.mode;
    const int par = ls.parity;
    const int cand = (mode == MODE_EVAL) ? 1 - par : par;   // buffer this sweep evaluates / writes

    double *W = s_wave + wave * TILE_D;
    double *Gs = W + W_G, *Hs = W + W_HEAD, *Ys = W + W_Y;
    double *red = s_red + wave * NRED;

    // ---- 2. block head of the accepted point -> wave-uniform registers ----
    // lane j holds head[j] (hsel) and lanes 0..7 head[64 + j] (hselb); uniform values are pulled out
    // with v_readlane, no LDS staging of the head
    const double hsel = par ? h1a : h0a, hselb = par ? h1b : h0b;
    auto bcast = [](double v, int src) -> double {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                                __builtin_amdgcn_readlane(__double2loint(v), src));
    };
    double e6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) e6[k] = bcast(hsel, BLK_EXT + k);
    double dnum = 0.0, dden = 0.0;
    double in[NINTR];
    if (mode == MODE_EVAL) {
        // delta_i = C_l^-1 (h - B^T delta_int) = u_h - U_B delta_int, U kept by the last sweep: lane
        // 8 + 6a + k multiplies U[a][k] by delta_int[a]; the products (and u_h, head[62..67]) are
        // transposed through LDS so that lane k < 6 adds up component k
        static_assert(BLK_U == 8 && BLK_PD == 68, "head layout");
        const bool isU = lane >= BLK_U && lane < BLK_U + 54;
        const double da = isU ? ls.dint[(lane - BLK_U) / 6] : 1.0;
        Hs[lane] = hsel * da;
        if (lane < 4) Hs[64 + lane] = hselb;
        wave_lds_sync();
        double dl = 0.0;
        if (lane < 6) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < NINTR; a++) acc += Hs[BLK_U + a * 6 + lane];
            dl = Hs[BLK_U + 54 + lane] - acc;
        }
        const bool ok = bcast(hselb, BLK_PD - 64) != 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double dk = ok ? bcast(dl, k) : 0.0;    // degenerate view: leave its pose unchanged
            dden += e6[k] * e6[k];
            dnum += dk * dk;
            e6[k] -= dk;
        }
#pragma unroll
        for (int a = 0; a < NINTR; a++) in[a] = ls.intr_cand[a];
    } else {
        // MODE_INIT / MODE_RESCHUR: (re-)evaluate the accepted point itself
#pragma unroll
        for (int a = 0; a < NINTR; a++) in[a] = ls.intr_prev[a];
    }
    const double aspect = ls.aspect, lam_next = ls.lam_next;
    wave_lds_sync();                              // head consumed: the tile may overwrite it

    // ---- 3. rows -> LDS -> f64 MFMA Gram ----
    STAMP(2);
    double R[9], K[9];
    rodrigues_RK(e6, R, K);
    STAMP(3);
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < P; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < P;
        if (c0 > 0) {
            const int pc = valid ? p : P - 1;
            uv = img2[pc];
            m0 = A.obj[3 * pc]; m1 = A.obj[3 * pc + 1]; m2 = A.obj[3 * pc + 2];
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

    if (A.gram_out != nullptr && active) {
#pragma unroll
        for (int r = 0; r < 4; r++) A.gram_out[(size_t)ii * 256 + lane + 64 * r] = Gs[lane + 64 * r];
    }

    // ---- 4. damped factor, U = C_l^-1 [B^T | h] and Y = L^-1 [B^T | h] at lam_next ----
    STAMP(7);
    double C[21], inv[6];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) C[tri(a, b)] = Gs[a * 16 + b];   // LDS broadcast reads
    damp_diag<6>(C, lam_next);
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
        double *bo = A.blk + ((size_t)cand * N + ii) * BLK;
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
        A.partials[((size_t)(A.k & 1) * gridDim.x + blockIdx.x) * NRED + tid] = s;
    }
    STAMP(10);
}

#ifdef CCAL_STAMPS
hipError_t debug_read_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
#endif

// ---------------------------------------------------------------------------------------
// partial rows -> G rows (row g = fixed-order sum of rows g, g + G, ...)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_reduce(const double *__restrict__ rows, int n_rows,
                                                           double *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) double s_part[RED_SLICES * NRED];
    __shared__ __attribute__((aligned(16))) double s_rs[NRED];
    const int G = gridDim.x, g = blockIdx.x;
    const int n_mine = (n_rows - g + G - 1) / G;
    // rows g, g + G, ... are a strided view: same kernel as the prologue with a row stride of G
    const int tid = threadIdx.x;
    const int j2 = tid % RED_PAIRS, c = tid / RED_PAIRS;
    if (c < RED_SLICES) {
        double s0 = 0.0, s1 = 0.0;
        const double2 *src = reinterpret_cast<const double2 *>(rows) + j2;
        for (int r0 = c; r0 < n_mine; r0 += RED_SLICES * RED_UNROLL) {
            double2 v[RED_UNROLL];
#pragma unroll
            for (int k = 0; k < RED_UNROLL; k++) {
                const int r = r0 + k * RED_SLICES;
                v[k] = r < n_mine ? src[((size_t)r * G + g) * RED_PAIRS] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int k = 0; k < RED_UNROLL; k++) { s0 += v[k].x; s1 += v[k].y; }
        }
        s_part[c * NRED + 2 * j2] = s0;
        s_part[c * NRED + 2 * j2 + 1] = s1;
    }
    __syncthreads();
    if (tid < NRED) {
        double t = s_part[tid];
#pragma unroll
        for (int k = 1; k < RED_SLICES; k++) t += s_part[k * NRED + tid];
        out[(size_t)g * NRED + tid] = t;
    }
    (void)s_rs;
}

// ---------------------------------------------------------------------------------------
// multi-GPU: partial rows -> one row -> sum over the ranks of the node (xGMI exchange,
// ccal_p2p.hpp) -> out[0..68).  ONE workgroup of 1024 threads: 30 slices x 34 column pairs pull
// the local rows with every load in flight at once (a hand-off between workgroups -- release
// fence + counter -- costs more than the 136 KB a single CU has to read), combine in fixed
// order and run the exchange.  Used only when there are too many partial rows for every
// workgroup of k_lm_trial to sum them (then the exchange is fused into its prologue instead).
// ---------------------------------------------------------------------------------------
constexpr int XCH_THREADS = 1024;
constexpr int XCH_SLICES = 30;      // 30 * RED_PAIRS = 1020 active threads
constexpr int XCH_UNROLL = 9;       // 270 rows per pass

__global__ __launch_bounds__(XCH_THREADS) void k_lm_exchange(const double *__restrict__ rows, int n_rows,
                                                              double *__restrict__ out, P2PDev p, int k) {
    __shared__ __attribute__((aligned(16))) double s_part[XCH_SLICES * NRED];
    __shared__ double s_val[NRED];
    __shared__ double s_all[P2P_MAX_RANKS * NRED];
    const int tid = threadIdx.x;
    const int j2 = tid % RED_PAIRS, c = tid / RED_PAIRS;
    if (c < XCH_SLICES) {
        double s0 = 0.0, s1 = 0.0;
        const double2 *src = reinterpret_cast<const double2 *>(rows) + j2;
        for (int r0 = c; r0 < n_rows; r0 += XCH_SLICES * XCH_UNROLL) {
            double2 v[XCH_UNROLL];
#pragma unroll
            for (int k = 0; k < XCH_UNROLL; k++) {
                const int r = r0 + k * XCH_SLICES;
                v[k] = r < n_rows ? src[(size_t)r * RED_PAIRS] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int k = 0; k < XCH_UNROLL; k++) { s0 += v[k].x; s1 += v[k].y; }
        }
        s_part[c * NRED + 2 * j2] = s0;
        s_part[c * NRED + 2 * j2 + 1] = s1;
    }
    __syncthreads();
    if (tid < NRED) {
        double t = s_part[tid];
#pragma unroll
        for (int k = 1; k < XCH_SLICES; k++) t += s_part[k * NRED + tid];
        s_val[tid] = t;
    }
    __syncthreads();
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p2p_exchange_rows(p, p2p_tag(seq, (unsigned)k), p2p_slot(seq, (unsigned)k), s_val, s_all, NRED, tid, true);
    if (tid < NRED) out[tid] = s_val[tid];
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
int lm_sweep_blocks(int N) { return (N + WAVES - 1) / WAVES; }

hipError_t launch_lm_trial(hipStream_t stream, const TrialLaunch &t) {
    TrialArgs a;
    a.obj = t.obj; a.img = t.img; a.blk = t.blk; a.st = t.st; a.partials = t.partials;
    a.rows_in = t.rows_in; a.n_in = t.n_in; a.trace = t.trace; a.host_result = t.host_result;
    a.host_progress = t.host_progress; a.intr0 = t.intr0; a.ext0 = t.ext0; a.ext0_stride = t.ext0_stride;
    a.init = t.init; a.gram_out = t.gram_out; a.N = t.N; a.P = t.P; a.k = t.k; a.first = t.first;
    a.final_only = t.final_only;
    a.xchg_fused = 0;
    memset(&a.p2p, 0, sizeof a.p2p);
    const int grid = t.final_only ? 1 : lm_sweep_blocks(t.N);
    if (t.p2p != nullptr) {
        a.p2p = *t.p2p;
        a.xchg_fused = t.xchg_fused;
        if (t.ev_start != nullptr && t.ev_stop != nullptr)
            hipExtLaunchKernelGGL(k_lm_trial<true>, dim3(grid), dim3(WAVES * 64), 0, stream, t.ev_start, t.ev_stop, 0, a);
        else
            hipLaunchKernelGGL(k_lm_trial<true>, dim3(grid), dim3(WAVES * 64), 0, stream, a);
        return hipGetLastError();
    }
    if (t.ev_start != nullptr && t.ev_stop != nullptr)   // events bound to this dispatch: its own begin / end
        hipExtLaunchKernelGGL(k_lm_trial<false>, dim3(grid), dim3(WAVES * 64), 0, stream, t.ev_start, t.ev_stop, 0, a);
    else
        hipLaunchKernelGGL(k_lm_trial<false>, dim3(grid), dim3(WAVES * 64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_lm_reduce(hipStream_t stream, const double *rows, int n_rows, double *out, int G) {
    hipLaunchKernelGGL(k_lm_reduce, dim3(G), dim3(WAVES * 64), 0, stream, rows, n_rows, out);
    return hipGetLastError();
}

hipError_t launch_lm_exchange(hipStream_t stream, const double *rows, int n_rows, double *out, const P2PDev &p, int k) {
    hipLaunchKernelGGL(k_lm_exchange, dim3(1), dim3(XCH_THREADS), 0, stream, rows, n_rows, out, p, k);
    return hipGetLastError();
}

}  // namespace ccal
