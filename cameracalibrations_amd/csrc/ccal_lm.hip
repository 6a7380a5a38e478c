// ccal_lm.hip -- the global Levenberg-Marquardt loop of calibrateCamera on gfx950.
//
// Replaces the loop OpenCV runs behind the reference's single FFI call
// (reference src/detect_fit.jl:47; algorithm: SURVEY.md rows A6-A9, Appendix A.3/A.6).
//
// ONE kernel per LM trial, decisions made on the device (the host only reads `done`):
//
//   k_lm_trial   prologue, every workgroup redundantly: the reduced intrinsic system of the previous
//                trial is read from the exact accumulators (R replicas x 5 limbs x 68 columns of int64,
//                22 KB at R = 8 -- not one row per workgroup), turned back into 68 doubles, and wave 0
//                advances the CvLevMarq state machine on an LDS copy of the state: accept / reject,
//                lambda schedule, stop rule, and the 9x9 damped Schur system for this trial's step,
//                solved with one matrix row per lane (scaled Gauss-Jordan, pivot row broadcast by DPP row_newbcast).
//                Workgroup 0 persists the state (double-buffered by trial parity) and reports to the
//                host.  Then the sweep: one 64-lane wavefront per image, one lane per corner.
//                  1. back-substitute this image's extrinsic step from the factor stored by the
//                     previous sweep (delta_i = L^-T (Y_h - Y_B delta_int)) -> candidate rvec/tvec
//                  2. per corner: residual + the two 16-wide rows of [Je | Ji | e]  (FP64)
//                  3. rows -> LDS (padded, conflict-free) -> v_mfma_f64_16x16x4_f64:
//                     G = [Je|Ji|e]^T [Je|Ji|e]  (16x16; holds JtJ blocks, Jt e and ||e||^2)
//                  4. damp + Cholesky of the 6x6 extrinsic block, Y = L^-1 [B^T | h], Schur
//                     contribution Y^T Y, workgroup sums -> fixed-point limbs -> integer atomics
//                     into acc[k % 3][blockIdx % R] (ccal_acc.hpp: exact, order-independent)
//   k_lm_row     only when a collective library sums the system across GPUs (RCCL / host hook):
//                accumulators -> one row of 68 doubles.
//   k_lm_exchange  multi-GPU with many workgroups, or ranks of several processes sharing one GPU: accumulators ->
//                one row -> direct xGMI exchange, by ONE workgroup (a waiting rank then holds one CU, not a grid).
//   k_lm_send / k_lm_recv  the same exchange in two stream-ordered halves for shards that share one in-order
//                stream (ccal_calibrate_multi with a device listed more than once).
//
//                Instantiations k_lm_trial<FLOW, LOOP, KIND, WPI>: looping grid or one image group per workgroup; first / later /
//                terminating launch; WPI = 2 or 4 waves per image for boards of more than 64 corners while the split grid is at
//                most one workgroup per compute unit (each wave takes an even share of the corners, the partial Grams meet in
//                LDS); FLOW = the prologue alone, or with the xGMI row exchange fused in (the tail-advance flows of round 4,
//                measured slower, are kept as tools/experiments/r04_tail_advance.patch: profiles/r04_tail_advance_probe.txt).
// A solve of T trials is T + 1 launches of k_lm_trial (the last one runs the prologue only and
// leaves the accumulators cleared for the next solve).
// A rejected step costs one extra sweep of the accepted point with the larger lambda.
// Multi-GPU (images sharded): every rank forms its own row, the 68-double row is summed over the ranks
// and every rank advances the same state.  Default transport: the direct row exchange fused into the
// prologue (k_lm_trial<true>, ccal_p2p.hpp) -- no extra launch; alternatives: k_lm_row -> RCCL or
// a host hook -> next k_lm_trial.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "ccal_acc.hpp"
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
#ifndef CCAL_STAMP_BLOCK
#define CCAL_STAMP_BLOCK 7
#endif
__device__ unsigned long long g_stamps[32];
// every workgroup's timeline on the chip-wide 100 MHz counter: entry, prologue done, exit, and where it ran
__device__ unsigned long long g_wg_times[1024 * 4];
#define WG_TIME(i)                                                                            \
    do {                                                                                      \
        if (KIND == 1 && threadIdx.x == 0 && blockIdx.x < 1024) {                             \
            unsigned long long t_;                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            g_wg_times[blockIdx.x * 4 + (i)] = t_;                                            \
        }                                                                                     \
    } while (0)
#define STAMP(i)                                                                              \
    do {                                                                                      \
        if (blockIdx.x == CCAL_STAMP_BLOCK && tid == 0) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_stamps[i] = t_;                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                \
        }                                                                                     \
    } while (0)
#define STAMP_RT(i)                                                                           \
    do {                                                                                      \
        if (blockIdx.x == CCAL_STAMP_BLOCK && tid == 0) {                                                    \
            unsigned long long t_;                                                            \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            g_stamps[i] = t_;                                                                 \
        }                                                                                     \
    } while (0)
#define STAMP2(i)                                                                             \
    do {                                                                                      \
        if (blockIdx.x == CCAL_STAMP_BLOCK && threadIdx.x == 0) {                                                               \
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
#define STAMP_RT(i) do { } while (0)
#define WG_TIME(i) do { } while (0)
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

// wave-uniform copy of lane `src`'s value (src is a compile-time constant after unrolling)
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}

constexpr int SROW = 10;       // row stride (doubles) of the symmetric copy of S0 in LDS: rows start 16-byte aligned

// 10^lg out of the four pre-read table entries (values, not references: stays in registers)
__device__ __forceinline__ double lam_pick(int lg, int lg0, int lg_dn, double same, double up, double dn, double dn2) {
    double v = dn2;
    v = (lg == lg_dn) ? dn : v;
    v = (lg == lg0 + 1) ? up : v;
    v = (lg == lg0) ? same : v;
    return v;
}

// 1 / d for a normal positive d: v_rcp_f64 + two Newton steps (no division sequence, no special cases)
__device__ __forceinline__ double rcp_pos(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}

// The nine elimination steps of the 9x9 solve, rows in lanes 0..8 (lanes 9..15 mirror row 8): pivot and pivot row
// reach the other lanes of the 16-lane row through DPP row broadcasts of lane J.  (Skipping the steps of fixed
// parameters -- identity rows -- behind a wave-uniform branch was measured: the nine basic blocks cost the
// straight-line version 500 cycles at K = 8, more than the one skipped step saves.)
template <int J>
__device__ __forceinline__ void gj9_steps(double (&a)[9], double &b, double &diag, bool &ok, int lane, std::integral_constant<int, J>) {
    const double d = row_bcast16<J>(a[J]);          // pivot
    ok = ok && (d > 0.0) && (d < 1.0e300);
    double prow[9];
#pragma unroll
    for (int k = 0; k < 9; k++) prow[k] = k > J ? row_bcast16<J>(a[k]) : 0.0;
    const double pb = row_bcast16<J>(b);
    gj_row_step<9>(J, lane == J, d, __builtin_amdgcn_rcp(d), prow, pb, a, b, diag);
    if constexpr (J < 8) gj9_steps(a, b, diag, ok, lane, std::integral_constant<int, J + 1>());
}

// [OCV-K CvLevMarq::updateAlt], SURVEY.md A.6, fused with the next step().
// Run by wave 0 of every workgroup on the workgroup's LDS copy of the state (every workgroup reaches
// the same decisions from the same bits); all 64 lanes execute it.  A trial's kernel time is one
// wavefront's dependent instruction chain, so the function is laid out for a short chain:
//   * everything it needs is read from LDS in one burst,
//   * the step of the next trial is solved SPECULATIVELY, in the same basic block as the accept /
//     reject / stop decisions (lambda of the step is known from the mode and the old lambda alone), so
//     the decision chain (error comparison, ||dp|| / ||p|| sums) hides in the solve's latency; a
//     rejected or terminating trial simply discards the step,
//   * the damped 9x9 Schur system is solved with row i in lane i by a scaled Gauss-Jordan elimination
//     (ccal_math.hpp gj_row_step): step j broadcasts pivot row j (one v_mov_b64_dpp row_newbcast per
//     value) and every other lane does  row_i <- row_i (d s) - (a_ij s) row_j  with s the raw v_rcp_f64
//     estimate of 1 / d -- four dependent FP64 operations per pivot, no refined reciprocal (square root),
//     and no back-substitution: after nine steps x_i = b_i / a_ii sits in lane i.
//     The system is symmetric positive definite (all pivots > 0 is the test Cholesky makes), so no
//     pivoting is needed; tests/test_acc_host.py checks the same arithmetic against exact rationals.
// rs = the 68 decoded sums; sS = S0 once more as a full symmetric 9 x 9 (row stride SROW) so that lane i fetches
// its row with five wide LDS reads.  Uniform values are stored by every lane to the same LDS address; per-parameter values by lanes 0..8.
// BOOK = true (workgroup 0 only, whose copy is persisted) also keeps the books the host reads:
// trace, per-trial kinds, counters, the reported relative step.
template <bool BOOK, bool FINAL>
__device__ __forceinline__ void lm_advance_wave(LmState *__restrict__ st, const double *__restrict__ rs,
                                                const double *__restrict__ sS, TraceRec *trace, const int lane, const int k) {
    // ---- one burst of LDS reads ----
    const int mode = st->mode;
    const int lam_lg0 = st->lambda_lg10;
    const int lg_part = st->lg_partials, iters0 = st->iters, max_iter = st->max_iter, trials0 = st->trials;
    const double prev_err = st->prev_err, eps = st->eps, aspect = st->aspect;
    const double err = rs[RED_ERR], r_dnum = rs[RED_DNUM], r_dden = rs[RED_DDEN], r_bad = rs[RED_BAD];
    const int la = lane < 9 ? lane : 8;
    const double ipl = st->intr_prev[la], icl = st->intr_cand[la], dAl = rs[RED_DA + la];
    // row `la` of S0 and of the right-hand side; fixed parameters arrive as zero rows / columns with dA = 0
    // (masked where they are produced, in the sweep), exactly like unobservable ones
    double a[9], b = rs[RED_GS + la];
    {
        const double2 *row = reinterpret_cast<const double2 *>(sS + la * SROW);
#pragma unroll
        for (int k = 0; k < 4; k++) { const double2 v = row[k]; a[2 * k] = v.x; a[2 * k + 1] = v.y; }
        a[8] = sS[la * SROW + 8];
    }
    const double lam_same = pow10_lg(st, lam_lg0), lam_up = pow10_lg(st, min(lam_lg0 + 1, 16)),
                 lam_dn = pow10_lg(st, max(lam_lg0 - 1, -16)), lam_dn2 = pow10_lg(st, max(lam_lg0 - 2, -16));
    const int lg_dn = max(lam_lg0 - 1, -16);
#define POW10_SEL(lg) lam_pick((lg), lam_lg0, lg_dn, lam_same, lam_up, lam_dn, lam_dn2)   /* 10^lg for the lambdas a decision can reach */
    int sweeps0 = 0, rejected0 = 0, reschur0 = 0;
    if (BOOK) { sweeps0 = st->sweeps; rejected0 = st->rejected; reschur0 = st->reschur; }

    // ---- decisions (wave-uniform, no stores yet) ----
    const bool is_init = mode == MODE_INIT, is_eval = mode == MODE_EVAL;
    const bool worse = err > prev_err;
    const bool reject = is_eval && worse && lam_lg0 + 1 <= 16;
    const bool accept = is_eval && !reject;
    // lambda after an accepted step (the ++ of a rejection that could not happen at 10^16 is undone by the --)
    const int lam_lg_acc = max(lam_lg0 + (worse ? 1 : 0) - 1, -16);
    const int lam_lg = accept ? lam_lg_acc : lam_lg0;           // lambda exponent of the step solved below
    const double lam = POW10_SEL(lam_lg);
    const bool nonfinite = !(fabs(err) < 1.0e300);
    // ||param - prevParam||^2 and ||prevParam||^2 over the whole parameter vector: the extrinsic part was
    // summed by the sweep, the nine intrinsic slots are added here (lanes 0..8 of the first row of 16)
    const double dl = lane < 9 ? icl - ipl : 0.0;
    const double dnum = r_dnum + lane_bcast(row_sum16(dl * dl), 0);
    const double dden = r_dden + lane_bcast(row_sum16(lane < 9 ? ipl * ipl : 0.0), 0);
    const bool small = dnum < eps * eps * dden;     // stop rule ||dp|| / ||p|| < eps without sqrt / divide
    double rel = 0.0;
    if (BOOK) rel = sqrt(dnum / dden);
    const double ip_new = accept ? icl : ipl;       // this lane's accepted intrinsic after the decision

    // ---- step(): damped Schur system  S = A_l - SC = S0 + lambda diag(A),  rhs = g - sg  (row A8) ----
    STAMP2(0);
    const bool fi = dAl > 0.0;                      // free and observable (delta = 0 otherwise: identity row)
    {
        const double dsel = fi ? fma(lam, dAl, sS[la * SROW + la]) : 1.0;
#pragma unroll
        for (int k = 0; k < 9; k++) a[k] = (la == k) ? dsel : a[k];
    }
    STAMP2(1);
    bool ok = true;                                 // (per lane; lane 0's copy decides)
    double diag = 1.0;                              // this lane's own pivot, rescaled along with its row
    // the terminating (prologue-only) launch of a solve has nothing to overlap the decisions with: it solves only when
    // a step will actually be taken (i.e. the solve goes on in another batch of launches)
    const bool will_step = !(reject || (nonfinite && (is_init || accept)) || (accept && (iters0 + 1 >= max_iter || small || lg_part != lam_lg)));
    if (!FINAL || will_step) gj9_steps(a, b, diag, ok, lane, std::integral_constant<int, 0>());
    ok = (__ballot(ok) & 1ull) != 0;
    STAMP2(2);
    // x_i = b_i / a_ii, one correction step on the quotient
    const double rd = rcp_pos(diag);
    double x = b * rd;
    x = fma(fma(-diag, x, b), rd, x);
    const double dlt = (lane < 9 && fi && ok) ? x : 0.0;
    double cand = ip_new - dlt;
    const double cand1 = row_bcast16<1>(cand);
    if (aspect > 0.0 && lane == 0) cand = aspect * cand1;
    STAMP2(3);

    // ---- the state machine proper: stores only ----
    const bool l0 = lane == 0;
    if (BOOK) {
        if (trials0 < MAX_KINDS) st->kinds[trials0] = 1 + mode;
        st->n_bad = (int)r_bad;
    }
    st->trials = trials0 + 1;
    // Trial k - 1 wrote its point's per-image blocks into buffer (k - 1) & 1.  Unless that point was a rejected
    // candidate it is the accepted point now; after a rejection the accepted point's blocks are still where they
    // were (the other buffer: an evaluating sweep never writes the buffer it reads).
    if (!reject) st->parity = (k - 1) & 1;
    if (is_init) {
        if (BOOK) {
            st->sweeps = sweeps0 + 1;
            if (l0) lm_trace(st, trace, err, -1.0, 2);
        }
        st->prev_err = err;
        if (nonfinite) { st->status = -4; st->done = 1; return; }
    } else if (is_eval) {
        if (BOOK) st->sweeps = sweeps0 + 1;
        if (reject) {
            // rejected: sweep the accepted point again with the larger lambda
            st->lambda_lg10 = lam_lg0 + 1;
            if (BOOK) {
                st->rejected = rejected0 + 1;
                if (l0) lm_trace(st, trace, err, -1.0, 0);
            }
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg0 + 1;
            st->lam_next = lam_up;
            return;
        }
        st->lambda_lg10 = lam_lg;
        st->iters = iters0 + 1;
        if (lane < 9) st->intr_prev[lane] = icl;
        st->prev_err = err;
        if (BOOK) {
            st->rel_step = rel;
            if (l0) lm_trace(st, trace, err, rel, 1);
        }
        if (nonfinite) { st->status = -4; st->done = 1; return; }
        if (iters0 + 1 >= max_iter || small) {
            if (!small) st->status = 1;             // CCAL_W_MAX_ITER
            st->done = 1;
            return;
        }
        if (lg_part != lam_lg) {                    // lambda saturated: the sums are for another lambda
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg;
            st->lam_next = lam;
            return;
        }
    } else {
        if (BOOK) st->reschur = reschur0 + 1;       // fresh system of the accepted point at the new lambda
    }
    if (!ok) {
        // damped system not positive definite (degenerate geometry): damp harder
        st->lambda_lg10 = lam_lg + 1;
        if (lam_lg + 1 <= 16) {
            st->mode = MODE_RESCHUR;
            st->lg_partials = lam_lg + 1;
            st->lam_next = POW10_SEL(lam_lg + 1);
        } else {
            st->status = -3;
            st->done = 1;
        }
        return;
    }
    if (lane < 9) {
        st->dint[lane] = dlt;
        st->intr_cand[lane] = cand;
    }
    st->lam_step = lam;
    const int lgn = max(lam_lg - 1, -16);
    st->lg_partials = lgn;
    st->lam_next = POW10_SEL(lgn);
    st->mode = MODE_EVAL;
#undef POW10_SEL
}

// wave-private LDS carve-up (doubles).  The Gram tile is dead once the MFMAs have issued, so
// everything after it re-uses the same 17 KB.
constexpr int W_G = 0;        // 16 x 16 Gram matrix
constexpr int W_HEAD = 256;   // the image's new 128-double block
constexpr int W_Y = 384;      // Y[j][k] = (L^-1 col_j)[k], 10 x 6
static_assert(W_Y + 60 <= TILE_D, "wave scratch exceeds the tile");

constexpr int STATE_WORDS = sizeof(LmState) / 8;
static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as 8-byte words");
static_assert(HOST_WORDS <= 64 && HOST_WORDS <= STATE_WORDS, "host mirror is written by one wave");
constexpr int ACC_PAIRS = ACC_ROW / 2;              // one replica is read as 170 16-byte pairs
static_assert(STATE_WORDS <= WAVES * 64 && ACC_PAIRS <= WAVES * 64, "one load per thread covers the state / a replica");
static_assert(ACC_ROW % 2 == 0, "replicas are read as 16-byte pairs");

// Reading the exact accumulators (ccal_acc.hpp) of one trial: thread t < 170 owns the 16-byte pair t of
// every replica; all loads are issued before anything waits on memory (one round trip), the replicas are
// added as split sums (no overflow for any grid size) and left in LDS for acc_row_decode().
struct AccRegs { longlong2 v[8]; };                // the first eight replicas' pair, requested before anything waits

// issue only: no arithmetic between these loads and the other loads of the prologue
__device__ __forceinline__ void acc_read_issue(const long long *__restrict__ acc, int R, int tid, AccRegs &q) {
    const longlong2 *src = reinterpret_cast<const longlong2 *>(acc) + (tid < ACC_PAIRS ? tid : 0);
#pragma unroll
    for (int r = 0; r < 8; r++) q.v[r] = (tid < ACC_PAIRS && r < R) ? src[(size_t)r * ACC_PAIRS] : make_longlong2(0, 0);
}

// Replica sums -> LDS.  Every replica's word is split on its own into the two exactly representable halves acc_decode()
// takes and the halves are added as doubles: one address receives at most ACC_MAX_ADDS adds of 52-bit payloads (no int64
// overflow per replica), while the total over all replicas of a grid of more than 2^11 workgroups would not fit an int64.
__device__ __forceinline__ void acc_read_stage(const long long *__restrict__ acc, int R, AccPart *s_acc, int tid, const AccRegs &q) {
    if (tid >= ACC_PAIRS) return;
    AccPart p0, p1;
    p0.hi = p0.lo = p1.hi = p1.lo = 0.0;
#pragma unroll
    for (int r = 0; r < 8; r++) { acc_part_add(p0, q.v[r].x); acc_part_add(p1, q.v[r].y); }
    const longlong2 *src = reinterpret_cast<const longlong2 *>(acc) + tid;
    for (int r0 = 8; r0 < R; r0 += 8) {             // more than eight replicas (very large grids only)
        longlong2 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            v[r] = (r0 + r >= R) ? make_longlong2(0, 0) : src[(size_t)(r0 + r) * ACC_PAIRS];
        }
#pragma unroll
        for (int r = 0; r < 8; r++) { acc_part_add(p0, v[r].x); acc_part_add(p1, v[r].y); }
    }
    acc_part_norm(p0);
    acc_part_norm(p1);
    s_acc[2 * tid] = p0;
    s_acc[2 * tid + 1] = p1;
}

// column `col` of the staged sums -> double (threads 0..67, after a barrier).  The pad column carries the
// flags: a NaN contribution anywhere makes the error NaN, an out-of-range one makes it +inf, so the
// state machine rejects (or aborts on) such a trial exactly as it does for a non-finite error.
__device__ __forceinline__ double acc_row_decode(const AccPart *s_acc, int col) {
    AccPart part[ACC_LIMBS];
#pragma unroll
    for (int j = 0; j < ACC_LIMBS; j++) part[j] = s_acc[j * NRED + col];
    double v = acc_decode(part);
    const AccPart n_nan = s_acc[0 * NRED + ACC_FLAG_COL], n_big = s_acc[1 * NRED + ACC_FLAG_COL];
    if (col == ACC_FLAG_COL) v = 0.0;
    if (col == RED_ERR) {
        if (n_big.hi != 0.0 || n_big.lo != 0.0) v = __longlong_as_double(0x7ff0000000000000ll);
        if (n_nan.hi != 0.0 || n_nan.lo != 0.0) v = __longlong_as_double(0x7ff8000000000000ll);
    }
    return v;
}

// this workgroup's 68 sums -> limbs -> integer atomics into one replica (threads 0..67)
__device__ __forceinline__ void acc_add(long long *__restrict__ dst, int col, double v) {
    int j, flag;
    int64_t lo, hi;
    acc_split(v, j, lo, hi, flag);
#ifdef CCAL_EXP_NO_ACC             // timing experiment only (no sums at all)
    if (col >= 0) return;
#endif
#ifdef CCAL_EXP_STORE_NOT_ATOMIC   // timing experiment only (wrong sums): what the atomics cost against plain stores
    dst[j * NRED + col] = lo;
    dst[(j + 1) * NRED + col] = hi;
    return;
#endif
    if (flag) {
        __hip_atomic_fetch_add(dst + (flag - 1) * NRED + ACC_FLAG_COL, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (lo != 0) __hip_atomic_fetch_add(dst + j * NRED + col, (long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (hi != 0) __hip_atomic_fetch_add(dst + (j + 1) * NRED + col, (long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// What the accumulators make of `n` workgroups' sums of one column, without the memory: every partial is split into its limbs,
// the limbs are added exactly and the total is decoded like a prologue decodes it (acc_read_stage + acc_decode) -- the one-launch
// solve of a few images (k_lm_trial<.., SOLO>) reproduces the bits of the launch-per-trial form with it.  flags: bit 0 = a NaN
// contribution, bit 1 = an out-of-range one.
template <int NP>
__device__ __forceinline__ double acc_sum_local(const double (&p)[NP], int &flags) {
    AccPart part[ACC_LIMBS];
#pragma unroll
    for (int j = 0; j < ACC_LIMBS; j++) part[j].hi = part[j].lo = 0.0;
    flags = 0;
#pragma unroll
    for (int i = 0; i < NP; i++) {
        int j, flag;
        int64_t lo, hi;
        acc_split(p[i], j, lo, hi, flag);
        flags |= flag;
#pragma unroll
        for (int jj = 0; jj < ACC_LIMBS; jj++) {
            if (flag == 0 && jj == j) acc_part_add(part[jj], lo);
            if (flag == 0 && jj == j + 1) acc_part_add(part[jj], hi);
        }
    }
#pragma unroll
    for (int j = 0; j < ACC_LIMBS; j++) acc_part_norm(part[j]);
    return acc_decode(part);
}

// clear `n_bufs` consecutive accumulator buffers (one workgroup, plain 16-byte stores)
__device__ __forceinline__ void acc_clear(long long *__restrict__ acc, int R, int n_bufs, int tid) {
    longlong2 *dst = reinterpret_cast<longlong2 *>(acc);
    const int n = n_bufs * R * ACC_PAIRS;
    for (int i = tid; i < n; i += WAVES * 64) dst[i] = make_longlong2(0, 0);
}

// Everything one launch needs, by value (a solve is a graph of these launches).
struct TrialArgs {
    // ---- first 64 bytes: everything the first burst of loads of a launch needs (one line of the kernel-argument segment;
    //      with -amdgpu-kernarg-preload-count=16 these sixteen dwords are in SGPRs when the wave starts) ----
    LmState *st;                  // [2], double-buffered by trial parity: S_k lives in st[k & 1]
    long long *acc;               // [3][R][ACC_ROW] exact accumulators: this trial adds into buffer k % 3,
                                  // its prologue reads buffer (k - 1) % 3, workgroup 0 clears buffer (k + 1) % 3
    double *blk;
    const float *img, *obj;
    const double *row_in;         // != NULL: the prologue takes the reduced system from this row of 68 doubles
                                  // (already summed across GPUs by RCCL / a host hook / k_lm_exchange) instead
    int N, P, k;
    int R;                        // replicas per buffer
    // ---- the rest ----
    const double *intr0, *ext0;   // start values (first trial only)
    int ext0_stride;
    int first;                    // k == 0: build S_0 instead of advancing
    int final_only;               // prologue only (detect termination, report to the host)
    int xchg_fused;               // image-sharded fit: the prologue exchanges the local row with the peer GPUs
    int max_trials;               // SOLO launches: the solve's trial budget (the launch ends there even if the solve has not)
    TraceRec *trace;
    LmState *host_result;
    unsigned long long *host_progress;
    double *gram_out;
    LmInit init;
    double *host_poses;           // != NULL: the launch that finds the solve done also writes the accepted poses there (pinned host
                                  // memory, N rows of 8 doubles: rvec, tvec, 2 pad) -- a small drop-in call then needs no read-back
    int *init_fail;               // first launch of a solve, workgroup 0: the initialisation's failure words (device) ...
    double *host_init_intr;       // ... and the start intrinsics go to pinned host memory (a pipelined fit needs no read-back kernel)
    int *host_init_fail;
    P2PDev p2p;                   // used by the exchanging instantiations only
    // MULTI launches (the workgroups of several independent fits in one grid: ccal_fit_many) -- the by-value argument carries only
    // these and k / first / final_only; everything else of a workgroup's fit comes from multi_table[multi_map[blockIdx.x]]
    const TrialArgs *multi_table;
    const int *multi_map;         // [grid]: which fit a workgroup belongs to
    const int *multi_first;       // [fits]: the fit's first workgroup in this grid
    const int *multi_nblk;        // [fits]: its number of workgroups
};
static_assert(offsetof(TrialArgs, R) + sizeof(int) == 64, "the launch's first loads take their addresses from the first 64 bytes");

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
// The wave-uniform operands of the rows (R, K, t, the intrinsics: 39 doubles) in scalar registers instead of 78 vector registers
#ifndef CCAL_SGPR_UNIFORMS
#define CCAL_SGPR_UNIFORMS 0
#endif
#ifndef CCAL_TRIAL_MIN_WAVES
#define CCAL_TRIAL_MIN_WAVES 2   /* two workgroups per CU (LDS-limited anyway): keep the register file share of two waves per SIMD */
#endif
// LOOP = the grid is smaller than the number of image groups (a workgroup sweeps several groups); the one-group form
// is compiled separately because the loop's carried state costs it 75 VGPRs and a few hundred cycles.
// KIND: 0 = first launch of a solve (builds S_0 instead of advancing), 1 = a later trial, 2 = the terminating
// prologue-only launch.  Separate instantiations: each kind drops the other kinds' paths (and their registers).
// WPI = waves per image (1, 2 or 4; non-looping grids only).  A board of more than 64 corners takes several 64-corner passes,
// and with one wave per image they are ONE wave's serial chain (rows -> tile -> Gram, three times over for 15 x 10 corners)
// while, for a few hundred images, most of the chip's SIMDs hold no wave at all.  With WPI > 1 the image's corners are split
// evenly over WPI waves of the workgroup -- each forms the rows and the 16 x 16 partial Gram of its share (the MFMA count
// follows the share: 38 corners are 20 instructions, not 32) --, the partial Grams meet in LDS and are added in wave order,
// and the image's first wave alone goes on to the 6 x 6 factor, the Schur contribution and the block.
// FLOW: where the LM state machine is advanced.
//   FLOW_PRO / FLOW_PRO_XCHG  in the PROLOGUE of the next launch, by every workgroup redundantly (from the accumulators,
//                  from an all-reduced row, or -- XCHG -- after the fused row exchange); a solve of T trials is T + 1 launches.
enum : int { FLOW_PRO = 0, FLOW_PRO_XCHG = 1 };
// SOLO: a fit of at most SOLO_WAVES images on one GPU runs EVERY trial in this one launch, one workgroup of SOLO_WAVES waves: the
//   sums of a trial stay in LDS (added up exactly like the accumulators would add the sums of the four-image workgroups of the
//   launch-per-trial form: same bits), wave 0 advances the state machine between two sweeps, a barrier is the hand-off -- no
//   kernel boundary, accumulator round trip, cold-cache prologue or terminating launch per trial, no graph.  KIND is 1.
//   Four waves, one per SIMD: measured (profiles/r05_one_launch_solve_probe.txt) a 5-sweep solve of 4 images takes 42.6 us
//   instead of 50.0; with eight waves (CCAL_SOLO_WAVES=8: up to 8 images) two sweeping waves share a SIMD, a trial grows by
//   1.6 us and the form returns nothing (6 images: 50.6 against 49.4 us), so beyond four images a trial stays a launch.
#ifndef CCAL_SOLO_WAVES
#define CCAL_SOLO_WAVES 4
#endif
constexpr int SOLO_WAVES = CCAL_SOLO_WAVES;
template <int FLOW, bool LOOP, int KIND, int WPI = 1, bool SOLO = false, bool MULTI = false>
#ifndef CCAL_SOLO_MINW
#define CCAL_SOLO_MINW 1
#endif
__global__ __launch_bounds__((SOLO ? SOLO_WAVES : WAVES) * 64, SOLO ? CCAL_SOLO_MINW : CCAL_TRIAL_MIN_WAVES) void k_lm_trial(const TrialArgs A_in) {
    constexpr bool XCHG = FLOW == FLOW_PRO_XCHG;
    static_assert(!MULTI || (FLOW == FLOW_PRO && !LOOP), "merged launches: single-GPU fits, one image group per workgroup");
    // MULTI: this workgroup's fit, its index among that fit's workgroups and their number (otherwise: the grid's)
    TrialArgs A_own;
    int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
    if constexpr (MULTI) {
        const int fit = A_in.multi_map[blockIdx.x];
        A_own = A_in.multi_table[fit];
        A_own.k = A_in.k; A_own.first = A_in.first; A_own.final_only = A_in.final_only;
        bid = (int)blockIdx.x - A_in.multi_first[fit];
        nblk = A_in.multi_nblk[fit];
    }
    const TrialArgs &A = MULTI ? A_own : A_in;
    (void)nblk;
    static_assert(!SOLO || (FLOW == FLOW_PRO && !LOOP && KIND == 1 && WPI == 1), "one-launch solves: single GPU, one image per wave");
    constexpr int NW = SOLO ? SOLO_WAVES : WAVES;   // waves of the workgroup
    static_assert(WPI == 1 || WPI == 2 || WPI == 4, "waves per image");
    static_assert(WPI == 1 || (!LOOP && KIND != 2), "split images: one group per workgroup");
    constexpr int IPW = NW / WPI;                // images per workgroup
    __shared__ __attribute__((aligned(16))) double s_wave[NW * TILE_D];
    __shared__ __attribute__((aligned(16))) double s_red[NW * NRED];
    __shared__ __attribute__((aligned(16))) double s_rs[NRED];
    __shared__ __attribute__((aligned(16))) double s_S[9 * SROW];
    __shared__ LmState ls;
    double *s_all = s_wave;       // XCHG: the peers' rows are staged in the (not yet used) tile space
    AccPart *s_acc = reinterpret_cast<AccPart *>(s_wave + P2P_MAX_RANKS * NRED);   // staged accumulator sums, likewise
    static_assert(P2P_MAX_RANKS * NRED + 2 * ACC_ROW <= NW * TILE_D, "prologue staging exceeds the tile space");

    const int tid = threadIdx.x;
    STAMP(0);
    STAMP_RT(30);
    WG_TIME(0);
#ifdef CCAL_STAMPS
    if (KIND == 1 && threadIdx.x == 0 && blockIdx.x < 1024) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_wg_times[blockIdx.x * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N, P = A.P;
#ifdef CCAL_STAMPS
    asm volatile("" ::"s"(N), "s"(A.st));        // the kernel arguments have arrived
    STAMP(13);
#ifdef CCAL_STAMPS_MEM
    // diagnostic of the diagnostic build: three dependent single-word round trips at kernel start -- a word of the
    // read-only corner stream, a word of the accumulators (last touched by memory-side atomics), a word of the
    // previous state (last written by workgroup 0's plain stores) -- to see which kind of line is slow after a boundary
    {
        const float w0 = __builtin_nontemporal_load(A.img + (size_t)blockIdx.x * 64 + (tid & 63));
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(w0));
        STAMP(14);
        const long long w1 = A.acc[(size_t)((A.k + 2) % 3) * A.R * ACC_ROW + (tid & 63)];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(w1));
        STAMP(15);
        const double w2 = reinterpret_cast<const double *>(A.st + ((A.k - 1) & 1))[tid & 63];
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(w2));
        STAMP(20);
    }
#endif
#endif
    // A workgroup sweeps the image groups g = blockIdx.x, blockIdx.x + gridDim.x, ... (four images each, one per
    // wave): the grid is capped at what the chip holds resident, so that the prologue above the loop -- state,
    // accumulators, LM advance -- runs once per resident workgroup, not once per four images.
    const int n_groups = (N + IPW - 1) / IPW;
    int g = bid;
    const int sub = WPI == 1 ? 0 : (wave & (WPI - 1));   // which share of the image's corners this wave takes
    const bool leader = sub == 0;               // the wave that factors the image and writes its block
    int img_i = g * IPW + (WPI == 1 ? wave : wave / WPI);
    bool active = img_i < N;                    // wave-uniform
    int ii = active ? img_i : N - 1;            // clamped: idle waves mirror the last image, store nothing
    // corners [cbeg, cend) of the image are this wave's (even shares: a Gram step consumes two corners)
    const int cpw = WPI == 1 ? P : ((((P + WPI - 1) / WPI) + 1) & ~1);
    const int cbeg = WPI == 1 ? 0 : min(sub * cpw, P);
    const int cend = WPI == 1 ? P : min(cbeg + cpw, P);
    const size_t acc_buf = (size_t)A.R * ACC_ROW;   // int64 words per accumulator buffer

    // kk = the trial this pass of the kernel runs: the launch's one trial -- or, SOLO, every trial of the solve in turn
    const int tid_launch = tid;
    // One trial.  `first` (build S_0 instead of advancing) is a compile-time constant of each copy: the launch's kind -- or, SOLO,
    // the peeled first trial and the loop over the later ones.  Returns false when the launch is over.
    auto trial = [&](auto first_c, const int kk) __attribute__((always_inline)) -> bool {
    constexpr bool first = decltype(first_c)::value;
    // SOLO: the thread index is made opaque once per trial, so that nothing derived from it (lane masks, per-lane addresses, the
    // packed-index decode) is hoisted out of the trial loop and kept alive across the sweep: that hoisting cost the one-launch form
    // 34 spilled vector registers at its two waves per SIMD
    int tid_trial = tid_launch;
    if constexpr (SOLO) asm volatile("" : "+v"(tid_trial));
    const int tid = tid_trial, lane = tid & 63;
    // ---- 0. loads that do not depend on the LM state are issued up front: the previous state, the
    //         accumulators of the previous trial, both parities of the block head (rvec,tvec | U | pd)
    //         -- or the start pose -- and the first 64 corners ----
    const LmState *prev = A.st + ((kk - 1) & 1);         // S_{k-1}
    double sw = 0.0, row_v = 0.0;
    AccRegs q;
    if (!first && !SOLO) {
        if (tid < STATE_WORDS) sw = reinterpret_cast<const double *>(prev)[tid];
        if (A.row_in != nullptr) { if (tid < NRED) row_v = A.row_in[tid]; }
        else acc_read_issue(A.acc + (size_t)((kk + 2) % 3) * acc_buf, A.R, tid, q);
    }
    // block words lane (ha) and 64 + lane (hb) of this image.  Trial k - 1 wrote buffer (k - 1) & 1; unless its point
    // was a rejected candidate (rare) that is the accepted point's block, so it is requested before the state is known
    double ha = 0.0, hb = 0.0;
    const float2 *img2 = reinterpret_cast<const float2 *>(A.img) + (size_t)ii * P;
    const int pc0 = min(cbeg + lane, max(cend - 1, 0));   // (an empty share reads corner 0 and uses nothing of it)
    float2 uv = make_float2(0.f, 0.f);
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if ((KIND != 2)) {
        if (first) {
            hb = (lane >= BLK_EXT - 64 && lane < BLK_EXT - 64 + 6) ? A.ext0[(size_t)ii * A.ext0_stride + lane - (BLK_EXT - 64)] : 0.0;
        } else {
            const double *bp = A.blk + ((size_t)((kk - 1) & 1) * N + ii) * BLK;
            ha = bp[lane];
            hb = bp[64 + lane];
        }
        uv = img2[pc0];                         // coalesced 8 B per lane
        m0 = A.obj[3 * pc0]; m1 = A.obj[3 * pc0 + 1]; m2 = A.obj[3 * pc0 + 2];
        // the buffer the NEXT trial adds into was last read one launch ago: clear it now (SOLO: no accumulators in use)
        if (!SOLO && bid == 0) acc_clear(A.acc + (size_t)((kk + 1) % 3) * acc_buf, A.R, 1, tid);
    }
    int oa, ob;                                 // (a, b) of the packed S0 entry this lane assembles
    tri_decode(lane < 45 ? lane : 0, oa, ob);
    // does the reduced-system entry this lane will assemble belong to free parameters only?  (lanes 0..44: S0(oa, ob);
    // 45..53: dA(a); 54..62: gs(a); 63: the error).  From the flag-word mask in the kernel arguments, worked out here
    // in the shadow of the first loads.
    bool keep;
    {
        unsigned int mbits = 0;
#pragma unroll
        for (int a = 0; a < NINTR; a++) mbits |= (A.init.mask[a] != 0 ? 1u : 0u) << a;
        const int ka = lane < RED_DA ? oa : (lane < RED_ERR ? (lane - RED_DA) % NINTR : 0);
        const int kb = lane < RED_DA ? ob : ka;
        keep = lane >= RED_ERR || (((mbits >> ka) & (mbits >> kb) & 1u) != 0);
    }

    // ---- 1. the state of this trial, S_k, in LDS (every workgroup computes the same bits) ----
    unsigned int seq = 0;
    if constexpr (XCHG) seq = __hip_atomic_load(A.p2p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first) {
        if (bid == 0 && A.host_init_intr != nullptr && tid < NINTR + 4) {
            // what the initialisation stages left for the host: nine start intrinsics, four failure words (ccal_fit reads them
            // after the solve; they used to travel by a kernel of their own between the pose stage and this launch).  Only when
            // a pose stage has run since the last report (its kernel raises word 2): stores to host memory hold the launch's
            // end back by a PCIe round trip, which a solve that merely restarts from the same initial values need not pay.
            const int fresh = A.init_fail[2];
            const int word = tid < NINTR ? 0 : (tid == NINTR + 2 ? 0 : A.init_fail[tid - NINTR]);
            if (fresh != 0) {
                if (tid < NINTR) __hip_atomic_store(A.host_init_intr + tid, A.intr0[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store(A.host_init_fail + (tid - NINTR), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (tid == NINTR + 2) A.init_fail[2] = 0;
            }
        }
        lm_boot(&ls, A.init, A.intr0, tid);
        // no other workgroup of this launch reads the sequence number; the launches after it see seq + 1
        if constexpr (XCHG)
            if (bid == 0 && tid == 0)
                __hip_atomic_store(A.p2p.epoch, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (SOLO) {
        // the sums of trial k - 1 are in s_rs / s_S (end of the previous pass), the state has stayed in LDS
        if (wave == 0) lm_advance_wave<true, false>(&ls, s_rs, s_S, A.trace, lane, kk);
        __syncthreads();
    } else {
        if (tid < STATE_WORDS) reinterpret_cast<double *>(&ls)[tid] = sw;
        if (A.row_in != nullptr) {
            if (tid < NRED) s_rs[tid] = row_v;
            if (tid < RED_DA) { s_S[oa * SROW + ob] = row_v; s_S[ob * SROW + oa] = row_v; }
        } else {
            acc_read_stage(A.acc + (size_t)((kk + 2) % 3) * acc_buf, A.R, s_acc, tid, q);
        }
        __syncthreads();
        STAMP(11);
        if (ls.done) {
            // solve already terminated: surplus launch.  The terminal state is carried along, so every
            // later launch of the batch sees it too (none of them sweeps or exchanges again); the
            // terminating launch leaves all accumulators cleared for the next solve.
            if (bid == 0) {
                LmState *cur = A.st + (kk & 1);
                if (tid < STATE_WORDS) reinterpret_cast<double *>(cur)[tid] = reinterpret_cast<const double *>(&ls)[tid];
                if ((KIND == 2)) acc_clear(A.acc, A.R, ACC_BUFS, tid);
            }
            return false;
        }
        if (A.row_in == nullptr) {
            if (tid < NRED) {
                const double v = acc_row_decode(s_acc, tid);
                s_rs[tid] = v;
                if (tid < RED_DA) { s_S[oa * SROW + ob] = v; s_S[ob * SROW + oa] = v; }   // (tid == lane of wave 0)
            }
            __syncthreads();
        }
        if constexpr (XCHG)
            if (A.xchg_fused) {
                p2p_exchange_rows(A.p2p, p2p_tag(seq, (unsigned)kk), p2p_slot(seq, (unsigned)kk), s_rs, s_all, NRED, tid,
                                  bid == 0);
                if (tid < RED_DA) {               // the symmetric copy again, now of the sum over the ranks (wave 0 only)
                    const double v = s_rs[tid];
                    s_S[oa * SROW + ob] = v;
                    s_S[ob * SROW + oa] = v;
                }
                if (wave == 0) wave_lds_sync();
            }
        STAMP(12);
        if (wave == 0) {
            if ((KIND == 2)) lm_advance_wave<true, true>(&ls, s_rs, s_S, A.trace, lane, kk);     // (one workgroup: it keeps the books)
            else if (bid == 0) lm_advance_wave<true, false>(&ls, s_rs, s_S, A.trace, lane, kk);
            else lm_advance_wave<false, false>(&ls, s_rs, s_S, nullptr, lane, kk);
        }
        __syncthreads();
    }
    STAMP(1);
    WG_TIME(1);
    const bool budget_out = SOLO && kk >= A.max_trials;   // the launch ends here even if the solve has not (the host reports it)
    if (bid == 0 && (!SOLO || ls.done || budget_out)) {
        LmState *cur = A.st + (kk & 1);
        if (tid < STATE_WORDS) reinterpret_cast<double *>(cur)[tid] = reinterpret_cast<const double *>(&ls)[tid];
        if (ls.done && A.host_poses != nullptr) {
            // the accepted poses, straight to pinned host memory ahead of the progress word (the blocks of every image were
            // complete one kernel boundary -- SOLO: one barrier -- ago)
            const double *bsrc = A.blk + (size_t)ls.parity * N * BLK + BLK_EXT;
            for (int i = tid; i < N * 6; i += NW * 64) {
                const int im = i / 6, w = i - 6 * im;
                __hip_atomic_store(A.host_poses + (size_t)im * 8 + w, bsrc[(size_t)im * BLK + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __threadfence_system();
            __syncthreads();
        }
        if (A.host_progress != nullptr && tid < 64) {
            // wave 0 alone talks to the host: the state header when the solve terminates, then the
            // progress word (trials << 32) | done that the host spins on, released at system scope
            if (ls.done != 0 && A.host_result != nullptr && tid < HOST_WORDS)
                reinterpret_cast<double *>(A.host_result)[tid] = reinterpret_cast<const double *>(&ls)[tid];
            if (tid == 0) {
                // the release (an L2 write-back) is only needed when the result header travels with the word
                const unsigned long long word = ((unsigned long long)(unsigned)ls.trials << 32) | (ls.done ? 1ull : 0ull);
                if (ls.done) __hip_atomic_store(A.host_progress, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store(A.host_progress, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        // the terminating launch of a finished solve leaves every accumulator cleared (its own input was
        // consumed before the barrier above); an unfinished one keeps its input for the trial that follows
        if ((KIND == 2) && ls.done) acc_clear(A.acc, A.R, ACC_BUFS, tid);
    }
    if (ls.done || (KIND == 2) || budget_out) return false;

    const int mode = ls.mode;
    const int par = ls.parity;                  // buffer holding the accepted point's blocks
    const int cand = kk & 1;                    // buffer this sweep writes (never the one an evaluating sweep reads)
    if (!first && par != ((kk - 1) & 1)) {
        // the previous trial's candidate was rejected: the accepted point's block is in the other buffer
        const double *bp = A.blk + ((size_t)par * N + ii) * BLK;
        ha = bp[lane];
        hb = bp[64 + lane];
    }

    double *W = s_wave + wave * TILE_D;
    double *Gs = W + W_G, *Hs = W + W_HEAD, *Ys = W + W_Y;
    double *red = s_red + wave * NRED;

    const double aspect = ls.aspect, lam_next = ls.lam_next;
    double in[NINTR];                             // the intrinsics this sweep evaluates: candidate, or (INIT / RESCHUR) accepted
    {
        const double *src = (mode == MODE_EVAL) ? ls.intr_cand : ls.intr_prev;
#pragma unroll
        for (int a = 0; a < NINTR; a++) in[a] = CCAL_SGPR_UNIFORMS ? to_sgpr(src[a]) : src[a];
    }
    auto bcast = [](double v, int src) -> double {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                                __builtin_amdgcn_readlane(__double2loint(v), src));
    };

    for (bool first_group = true;; first_group = false) {
    const int g_next = g + nblk;
    const bool last_group = !LOOP || g_next >= n_groups;   // workgroup-uniform
    // ---- 2. block head of the accepted point -> wave-uniform registers ----
    // lane j holds block word j (hsel) and word 64 + j (hselb); uniform values are pulled out with
    // v_readlane, no LDS staging of the block
    const double hsel = ha, hselb = hb;
    double e6[6];
#pragma unroll
    for (int k = 0; k < 6; k++) e6[k] = bcast(hselb, BLK_EXT - 64 + k);
    double dnum = 0.0, dden = 0.0;
#ifdef CCAL_EXP_SKIP_A             // timing experiment only (wrong results): what pose update + rotation cost a launch
    if (false) {
#else
    if (mode == MODE_EVAL) {
#endif
        // delta_i = C_l^-1 (h - B^T delta_int) = u_h - U_B delta_int, U kept by the last sweep one value per
        // lane: lane 16 k + j multiplies U[j][k] by (-delta_int[j] | +1 for j = 9 | 0), a DPP sum inside each row
        // of 16 lanes leaves component k in row k (k = 0..3: first block word; k = 4, 5: second one)
        const int j16 = lane & 15;
        const double mj = j16 < NINTR ? -ls.dint[j16 < NINTR ? j16 : 0] : (j16 == NINTR ? 1.0 : 0.0);
        const double sA = row_sum16(hsel * mj);
        const double sB = row_sum16(lane < 32 ? hselb * mj : 0.0);
        const bool ok = bcast(hselb, BLK_PD - 64) != 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double dl = k < 4 ? bcast(sA, 16 * k) : bcast(sB, 16 * (k - 4));
            const double dk = ok ? dl : 0.0;              // degenerate view: leave its pose unchanged
            dden += e6[k] * e6[k];
            dnum += dk * dk;
            e6[k] -= dk;
        }
    }
    wave_lds_sync();                              // head consumed: the tile may overwrite it

    // ---- 3. rows -> LDS -> f64 MFMA Gram ----
    STAMP(2);
    double R[9], K[9];
#ifdef CCAL_EXP_SKIP_A
#pragma unroll
    for (int q_ = 0; q_ < 9; q_++) { R[q_] = (q_ % 4 == 0) ? 1.0 : e6[q_ % 3] * 1e-3; K[q_] = R[q_]; }
#else
    rodrigues_RK(e6, R, K);
#endif
    double tv[3] = {e6[3], e6[4], e6[5]};
    if (CCAL_SGPR_UNIFORMS) {
#pragma unroll
        for (int q_ = 0; q_ < 9; q_++) { R[q_] = to_sgpr(R[q_]); K[q_] = to_sgpr(K[q_]); }
#pragma unroll
        for (int q_ = 0; q_ < 3; q_++) tv[q_] = to_sgpr(tv[q_]);
    }
    STAMP(3);
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    // (SOLO: a wave without an image forms no rows -- it would only take the SIMD it shares from a wave that has one)
    const int c_stop = (SOLO && !active) ? cbeg : cend;
    for (int c0 = cbeg; c0 < c_stop; c0 += 64) {
        const int p = c0 + lane;
        const bool valid = p < cend;
        if (c0 > cbeg) {
            const int pc = valid ? p : cend - 1;
            uv = img2[pc];
            m0 = A.obj[3 * pc]; m1 = A.obj[3 * pc + 1]; m2 = A.obj[3 * pc + 2];
        }
        const double M[3] = {(double)m0, (double)m1, (double)m2};
        double mu[NCOL], mv[NCOL];
        point_rows_K(M, (double)uv.x, (double)uv.y, R, K, tv, in, aspect, mu, mv, valid ? 1.0 : 0.0);
        STAMP(4);
#ifdef CCAL_HALF_TILE
        // half-size tile (occupancy): lanes 0..31 stage their rows, Gram of those corners, then lanes 32..63
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int n_h = min(32, cend - c0 - 32 * h);
            if (n_h > 0) {
                if ((lane >> 5) == h) tile_store_rows_masked(W, lane, mu, mv);
                wave_lds_sync();
                tile_gram(W, lane, n_h, acc0, acc1);
                wave_lds_sync();
            }
        }
#else
        tile_store_rows_masked(W, lane, mu, mv);
        wave_lds_sync();
        STAMP(5);
        tile_gram(W, lane, min(64, cend - c0), acc0, acc1);
        wave_lds_sync();                          // operands are in registers: the tile is free
#endif
    }
    gram_to_lds(Gs, lane, acc0, acc1);
    STAMP(6);
    if constexpr (WPI > 1) {
        // the partial Grams of the image's waves meet here; its first wave adds them in wave order (fixed: same bits every run)
        __syncthreads();
        if (leader) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double t = Gs[lane + 64 * r];
#pragma unroll
                for (int j = 1; j < WPI; j++) t += Gs[j * TILE_D + lane + 64 * r];
                Gs[lane + 64 * r] = t;
            }
        }
    }
    wave_lds_sync();


    const bool work = (WPI == 1 || leader) && (!SOLO || active);   // (wave-uniform) this wave carries the image beyond its Gram
    if (A.gram_out != nullptr && active && work) {
#pragma unroll
        for (int r = 0; r < 4; r++) A.gram_out[(size_t)ii * 256 + lane + 64 * r] = Gs[lane + 64 * r];
    }

    // ---- 4. damped factor, U = C_l^-1 [B^T | h] and Y = L^-1 [B^T | h] at lam_next ----
    STAMP(7);
    double C[21], inv[6];
    bool pd = false;
    // lane j < 10 owns column j of [B^T | h]
    const int yj = lane < 10 ? lane : 9;
    double y[6];
#ifdef CCAL_EXP_SKIP_C             // timing experiment only (wrong results): what the 6 x 6 factor and the Schur outputs cost a launch
    if (work && kk > 1000) {
#else
    if (work) {
#endif
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b <= a; b++) C[tri(a, b)] = Gs[a * 16 + b];   // LDS broadcast reads
    damp_diag<6>(C, lam_next);
    pd = chol_packed<6>(C, inv);
    {
        const int col = yj < 9 ? 6 + yj : COL_E;
#pragma unroll
        for (int k = 0; k < 6; k++) y[k] = Gs[k * 16 + col];
        fwd_subst<6>(C, inv, y);
        if (lane < 10) {
#pragma unroll
            for (int k = 0; k < 6; k++) Ys[yj * 6 + k] = y[k];
        }
    }
    wave_lds_sync();

    // ---- this image's contributions to the reduced system: S0 | dA | gs | err dnum dden bad.  They go
    //      first: their atomics are the only thing the next launch waits for, everything after them
    //      (U, the block store) is local and runs while they are in flight ----
    STAMP(8);
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
        STAMP(21);
        // this wave's sums over its images (wave-private LDS; fixed parameters leave zero rows / columns, dA = 0, g = 0)
        const double vv = (active && keep) ? v : 0.0;
        red[lane] = first_group ? vv : red[lane] + vv;
        if (lane < 4) {
            const double w = lane == 0 ? dnum : (lane == 1 ? dden : (lane == 2 ? ((pd && finite) ? 0.0 : 1.0) : 0.0));
            const double ww = active ? w : 0.0;
            red[64 + lane] = first_group ? ww : red[64 + lane] + ww;
        }
    }
    } else {                                      // a helper wave of a split image: its share is in the leader's Gram
        red[lane] = 0.0;
        if (lane < 4) red[64 + lane] = 0.0;
    }
    if (last_group) {
    STAMP(22);
    __syncthreads();
    STAMP(23);
    if constexpr (SOLO) {
        // the sums of the launch-per-trial form's four-image workgroups, each in wave order, added up exactly and decoded: what
        // the next prologue would have read from the accumulators, bit for bit -- straight into s_rs / s_S for the next pass
        double v = 0.0;
        int fl = 0;
        if (tid < NRED) {
            double pp[SOLO_WAVES / WAVES];
#pragma unroll
            for (int gq = 0; gq < SOLO_WAVES / WAVES; gq++) {
                double sg = s_red[gq * WAVES * NRED + tid];
#pragma unroll
                for (int w = 1; w < WAVES; w++) sg += s_red[(gq * WAVES + w) * NRED + tid];
                pp[gq] = sg;
            }
            v = acc_sum_local(pp, fl);
        }
        const int any_nan = __syncthreads_or(fl & 1), any_big = __syncthreads_or(fl & 2);
        if (tid < NRED) {
            if (tid == ACC_FLAG_COL) v = 0.0;
            if (tid == RED_ERR) {
                if (any_big) v = __longlong_as_double(0x7ff0000000000000ll);
                if (any_nan) v = __longlong_as_double(0x7ff8000000000000ll);
            }
            s_rs[tid] = v;
            if (tid < RED_DA) { s_S[oa * SROW + ob] = v; s_S[ob * SROW + oa] = v; }   // (tid == lane of wave 0)
        }
    } else if (tid < NRED) {
        double s = s_red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; w++) s += s_red[w * NRED + tid];   // fixed order inside the workgroup
        // ... and exact, order-independent across workgroups: limbs added with integer atomics
        // (replica = workgroup index mod R: with R = 8 and round-robin dispatch that is the workgroup's XCD, so one
        //  replica is only ever added to through one L2 -- choosing it from HW_REG_XCC_ID instead measured the same)
        acc_add(A.acc + ((size_t)(kk % 3) * A.R + (bid & (A.R - 1))) * ACC_ROW, tid, s);
    }
    }
    STAMP(9);

    // ---- U = L^-T Y and this image's block for the next sweep's back-substitution ----
#ifdef CCAL_EXP_SKIP_C
    if (work && kk > 1000) {
#else
    if (work) {
#endif
    bwd_subst<6>(C, inv, y);
    Hs[lane] = 0.0;                               // (the DS unit keeps one wave's writes in order)
    Hs[64 + lane] = 0.0;
    if (lane < 10) {
#pragma unroll
        for (int k = 0; k < 6; k++) Hs[(k < 4 ? 16 * k : 64 + 16 * (k - 4)) + yj] = y[k];
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; k++) Hs[BLK_EXT + k] = e6[k];
        Hs[BLK_PD] = pd ? 1.0 : 0.0;
    }
    wave_lds_sync();
    if (active) {
        double *bo = A.blk + ((size_t)cand * N + ii) * BLK;
#ifdef CCAL_EXP_NO_BLOCK           // timing experiment only (wrong results): what the block stores add to the launch boundary
        if (kk > 1000) { bo[lane] = Hs[lane]; bo[64 + lane] = Hs[64 + lane]; }
#else
        bo[lane] = Hs[lane];
        bo[64 + lane] = Hs[64 + lane];
#endif
    }
    }
    if (last_group) break;
    g = g_next;
    // next group of this workgroup: its images' blocks and first corners (requesting them earlier, behind the
    // factorisation, changed nothing: measured)
    img_i = g * WAVES + wave;
    active = img_i < N;
    ii = active ? img_i : N - 1;
    img2 = reinterpret_cast<const float2 *>(A.img) + (size_t)ii * P;
    if (first) {
        hb = (lane >= BLK_EXT - 64 && lane < BLK_EXT - 64 + 6) ? A.ext0[(size_t)ii * A.ext0_stride + lane - (BLK_EXT - 64)] : 0.0;
    } else {
        const double *bp = A.blk + ((size_t)par * N + ii) * BLK;
        ha = bp[lane];
        hb = bp[64 + lane];
    }
    uv = img2[pc0];
    m0 = A.obj[3 * pc0]; m1 = A.obj[3 * pc0 + 1]; m2 = A.obj[3 * pc0 + 2];
    wave_lds_sync();                              // the block has left LDS: the next group's tile may overwrite it
    }
    STAMP(10);
    if constexpr (SOLO) __syncthreads();          // s_rs / s_S complete; every wave has consumed this pass's state
    return true;
    };
    if constexpr (SOLO) {
        if (trial(std::true_type(), 0))
            for (int kk = 1; trial(std::false_type(), kk); kk++) {}
    } else {
        trial(std::integral_constant<bool, KIND == 0>(), A.k);
    }
    STAMP_RT(31);
    WG_TIME(2);
}

#ifdef CCAL_STAMPS
hipError_t debug_read_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32);
}
hipError_t debug_read_wg_times(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), sizeof(unsigned long long) * 4096);
}
#endif

// ---------------------------------------------------------------------------------------
// accumulators of trial k - 1 -> one row of 68 doubles (one workgroup); used when a collective
// library or a host hook sums the row across GPUs between two trials.  Does nothing once the
// solve has terminated.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_row(const long long *__restrict__ acc, int R, const LmState *__restrict__ st,
                                                        int k, double *__restrict__ out) {
    __shared__ AccPart s_acc[ACC_ROW];
    const int tid = threadIdx.x;
    if (__hip_atomic_load(&st[(k - 1) & 1].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    AccRegs q;
    acc_read_issue(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, tid, q);
    acc_read_stage(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, s_acc, tid, q);
    __syncthreads();
    if (tid < NRED) out[tid] = acc_row_decode(s_acc, tid);
}

// ---------------------------------------------------------------------------------------
// multi-GPU with many workgroups per rank: accumulators -> one row -> sum over the ranks of the node
// (xGMI exchange, ccal_p2p.hpp) -> out[0..68), one workgroup, so that thousands of workgroups do not
// each poll the inbox (few workgroups: the exchange is fused into k_lm_trial's prologue instead).
// Does nothing once the solve has terminated -- the peers stop exchanging at the same trial, since
// every rank advances the same state.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_exchange(const long long *__restrict__ acc, int R, const LmState *__restrict__ st,
                                                             double *__restrict__ out, P2PDev p, int k) {
    __shared__ AccPart s_acc[ACC_ROW];
    __shared__ double s_val[NRED];
    __shared__ double s_all[P2P_MAX_RANKS * NRED];
    const int tid = threadIdx.x;
    if (__hip_atomic_load(&st[(k - 1) & 1].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;   // a peer already timed out
    AccRegs q;
    acc_read_issue(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, tid, q);
    acc_read_stage(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, s_acc, tid, q);
    __syncthreads();
    if (tid < NRED) s_val[tid] = acc_row_decode(s_acc, tid);
    __syncthreads();
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p2p_exchange_rows(p, p2p_tag(seq, (unsigned)k), p2p_slot(seq, (unsigned)k), s_val, s_all, NRED, tid, true);
    if (tid < NRED) out[tid] = s_val[tid];
}

// ---------------------------------------------------------------------------------------
// The same exchange as two kernels, for shards of one fit that share a device (one in-order stream): every
// shard's k_lm_send is enqueued before any shard's k_lm_recv, so nothing that waits ever sits in front of what
// it waits for (ccal_p2p_dev.hpp).  k_lm_send: accumulators -> row -> every inbox, the own one included;
// k_lm_recv: all rows of exchange k -> sum in rank order -> out[0..68).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(WAVES * 64) void k_lm_send(const long long *__restrict__ acc, int R, const LmState *__restrict__ st,
                                                         P2PDev p, int k) {
    __shared__ AccPart s_acc[ACC_ROW];
    __shared__ double s_val[NRED];
    const int tid = threadIdx.x;
    if (__hip_atomic_load(&st[(k - 1) & 1].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    AccRegs q;
    acc_read_issue(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, tid, q);
    acc_read_stage(acc + (size_t)((k + 2) % 3) * R * ACC_ROW, R, s_acc, tid, q);
    __syncthreads();
    if (tid < NRED) s_val[tid] = acc_row_decode(s_acc, tid);
    __syncthreads();
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p2p_send_row(p, p2p_tag(seq, (unsigned)k), p2p_slot(seq, (unsigned)k), s_val, NRED, tid);
}

__global__ __launch_bounds__(WAVES * 64) void k_lm_recv(const LmState *__restrict__ st, double *__restrict__ out, P2PDev p, int k) {
    __shared__ double s_val[NRED];
    __shared__ double s_all[P2P_MAX_RANKS * NRED];
    const int tid = threadIdx.x;
    if (__hip_atomic_load(&st[(k - 1) & 1].done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;   // a peer already timed out
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p2p_recv_rows(p, p2p_tag(seq, (unsigned)k), p2p_slot(seq, (unsigned)k), s_val, s_all, NRED, tid);
    if (tid < NRED) out[tid] = s_val[tid];
}

// ---------------------------------------------------------------------------------------
// diagnostics: the exact accumulators on their own (ccal_debug_acc_sum).  "Workgroup" b adds its row of
// 68 doubles into replica b mod R exactly as a sweep does; one workgroup then reads and decodes the sums
// exactly as a trial's prologue does.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_acc_debug_add(const double *__restrict__ vals, long long *__restrict__ acc, int R) {
    const int tid = threadIdx.x;
    if (tid < NRED) acc_add(acc + (size_t)(blockIdx.x & (R - 1)) * ACC_ROW, tid, vals[(size_t)blockIdx.x * NRED + tid]);
}

__global__ __launch_bounds__(WAVES * 64) void k_acc_debug_read(const long long *__restrict__ acc, int R, double *__restrict__ out) {
    __shared__ AccPart s_acc[ACC_ROW];
    const int tid = threadIdx.x;
    AccRegs q;
    acc_read_issue(acc, R, tid, q);
    acc_read_stage(acc, R, s_acc, tid, q);
    __syncthreads();
    if (tid < NRED) out[tid] = acc_row_decode(s_acc, tid);
}

hipError_t launch_acc_debug(hipStream_t stream, const double *vals, int n_rows, long long *acc, int R, double *out) {
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(long long) * (size_t)R * ACC_ROW, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_acc_debug_add, dim3(n_rows), dim3(128), 0, stream, vals, acc, R);
    hipLaunchKernelGGL(k_acc_debug_read, dim3(1), dim3(WAVES * 64), 0, stream, acc, R, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------------------
int lm_sweep_blocks(int N) { return (N + WAVES - 1) / WAVES; }
int lm_solo_max_images() { return SOLO_WAVES; }
// which k_lm_trial<FLOW, LOOP, KIND, WPI, SOLO> a sweep of N images launches (ccal_stats.sweep_form): FLOW | LOOP << 4 | WPI << 8 | SOLO << 12
int lm_sweep_form(int N, int wpi, bool xchg, bool solo) {
    if (solo) return 1 << 8 | 1 << 12;
    const int groups = wpi == 1 ? lm_sweep_blocks(N) : (N * wpi + WAVES - 1) / WAVES;
    return (xchg ? 1 : 0) | (groups > lm_max_grid() ? 1 << 4 : 0) | wpi << 8;
}

// Waves per image of the sweep (k_lm_trial's WPI), decided ONCE per problem (ccal_set_problem keeps it in the context: the Gram
// of a split image is summed in another order, so the bits of a fit are a function of (device, N, P) -- and of nothing that could
// change between two launches of one solve).  Measured rules:
//  * boards of more than 64 corners (several 64-corner passes) are split over 4 or 2 waves as long as the split grid still is at
//    most ONE workgroup per compute unit: every wave repeats the image's pose and rotation, so beyond that the extra waves cost
//    the SIMDs they share more than the shorter chain returns (15 x 10 corners, launch duration in a solve, wpi 1 / 2 / 4 -- 100
//    images: 10.6 / 9.4 / 8.2 us; 500: 11.0 / 10.3 / 11.4; 1000: 12.9 / 12.9 / 12.9: profiles/r04_pass_parallel_probe.txt; with
//    three or four waves per SIMD resident, round 5, the 1250-image share gains 0.4 us: profiles/r05_role_split_probe.txt);
//  * boards of ONE pass (9 x 6 corners: 28 matrix instructions become 8) are split over 4 waves while the split grid is at most
//    one workgroup per TWO compute units: 100 images 7.2 -> 6.9 us per launch, 24.5 -> 23.7 us per solve; 250 images even, 500
//    and more slower (8.2 -> 10.4 us); two waves return nothing at any size (profiles/r05_wpi_small_board_probe.txt).
// CCAL_WPI = 1 / 2 / 4 forces a value, CCAL_WPI_CAP the grid limit, CCAL_WPI_MINP the smallest board that is split (experiments).
int lm_waves_per_image(int N, int P, bool fused) {
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cus[dev] <= 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        (void)hipGetLastError();
        cus[dev] = n;
    }
    (void)fused;                                  // (the fused exchange's 320-workgroup limit lies above both caps)
    int min_p = 16;
    if (const char *e = getenv("CCAL_WPI_MINP")) {
        const int v = atoi(e);
        if (v >= 8) min_p = v;
    }
    if (P < min_p) return 1;
    const bool one_pass = P <= 64;
    int cap = std::min(one_pass ? cus[dev] / 2 : cus[dev], lm_max_grid());
    int want = 4;
    if (const char *e = getenv("CCAL_WPI")) {
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) want = v;
    }
    if (const char *e = getenv("CCAL_WPI_CAP")) {
        const int v = atoi(e);
        if (v > 0) cap = std::min(v, lm_max_grid());
    }
    for (int w = want; w > 1; w /= 2) {
        if (one_pass && w == 2 && want != 2) break;   // (a one-pass board gains nothing from two waves)
        if (((long long)N * w + WAVES - 1) / WAVES <= cap) return w;
    }
    return 1;
}

// Workgroups of k_lm_trial the current device keeps resident (occupancy query x compute units; CCAL_MAX_GRID
// overrides).  Both instantiations use the same resources up to a few SGPRs.
constexpr int LM_GRID_LIMIT = 8192;   // (ACC_MAX_ADDS adds per address with up to ACC_MAX_REPLICAS replicas would allow far more)
int lm_max_grid() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (cached[dev] > 0) return cached[dev];
    int per_cu = 0, cus = 0;
    if (const char *e = getenv("CCAL_MAX_GRID")) {
        const int v = atoi(e);
        if (v > 0) return cached[dev] = std::min(v, LM_GRID_LIMIT);
    }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_lm_trial<FLOW_PRO_XCHG, true, 1>, WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    return cached[dev] = std::min(per_cu * cus, LM_GRID_LIMIT);
}

MergeRecorder *&merge_recorder() {
    static thread_local MergeRecorder *r = nullptr;
    return r;
}
static_assert(sizeof(TrialArgs) <= sizeof(MergeRecord::args), "a recorded launch keeps its arguments by value");

static void merge_add(MergeRecorder *r, int kernel, int key, int grid, int block, size_t dyn, const TrialArgs &a) {
    MergeRecord m;
    m.kernel = kernel; m.key = key; m.grid = grid; m.block = block; m.dyn_lds = dyn;
    m.k = a.k; m.first = a.first; m.final_only = a.final_only;
    memcpy(m.args, &a, sizeof a);
    m.args_bytes = sizeof a;
    r->recs.push_back(m);
}

// key = KIND | WPI << 4 | SOLO << 8
hipError_t launch_lm_trial_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, const int *nblk, int grid,
                                 int k, int is_first, int final_only) {
    TrialArgs a;
    memset(&a, 0, sizeof a);
    a.multi_table = static_cast<const TrialArgs *>(table); a.multi_map = map; a.multi_first = first; a.multi_nblk = nblk;
    a.k = k; a.first = is_first; a.final_only = final_only;
    const int kind = key & 15, wpi = (key >> 4) & 15, solo = (key >> 8) & 1;
    auto go = [&](auto kernel, int block) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, a); };
    if (solo) go(k_lm_trial<FLOW_PRO, false, 1, 1, true, true>, SOLO_WAVES * 64);
    else if (kind == 2) go(k_lm_trial<FLOW_PRO, false, 2, 1, false, true>, WAVES * 64);
    else if (kind == 0 && wpi == 4) go(k_lm_trial<FLOW_PRO, false, 0, 4, false, true>, WAVES * 64);
    else if (kind == 1 && wpi == 4) go(k_lm_trial<FLOW_PRO, false, 1, 4, false, true>, WAVES * 64);
    else if (kind == 0 && wpi == 2) go(k_lm_trial<FLOW_PRO, false, 0, 2, false, true>, WAVES * 64);
    else if (kind == 1 && wpi == 2) go(k_lm_trial<FLOW_PRO, false, 1, 2, false, true>, WAVES * 64);
    else if (kind == 0 && wpi == 1) go(k_lm_trial<FLOW_PRO, false, 0, 1, false, true>, WAVES * 64);
    else if (kind == 1 && wpi == 1) go(k_lm_trial<FLOW_PRO, false, 1, 1, false, true>, WAVES * 64);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_lm_trial(hipStream_t stream, const TrialLaunch &t) {
    if (t.acc == nullptr || t.R < 1 || t.R > ACC_MAX_REPLICAS || (t.R & (t.R - 1)) != 0) return hipErrorInvalidValue;
    TrialArgs a;
    a.obj = t.obj; a.img = t.img; a.blk = t.blk; a.st = t.st; a.acc = t.acc; a.R = t.R;
    a.row_in = t.row_in; a.trace = t.trace; a.host_result = t.host_result;
    a.host_progress = t.host_progress; a.intr0 = t.intr0; a.ext0 = t.ext0; a.ext0_stride = t.ext0_stride;
    a.init = t.init; a.gram_out = t.gram_out; a.N = t.N; a.P = t.P; a.k = t.k; a.first = t.first;
    a.final_only = t.final_only;
    a.xchg_fused = 0;
    a.max_trials = t.max_trials;
    a.host_poses = t.host_poses;
    a.init_fail = t.init_fail; a.host_init_intr = t.host_init_intr; a.host_init_fail = t.host_init_fail;
    memset(&a.p2p, 0, sizeof a.p2p);
    a.multi_table = nullptr; a.multi_map = nullptr; a.multi_first = nullptr; a.multi_nblk = nullptr;
    MergeRecorder *rec = merge_recorder();
    if (rec != nullptr && (t.p2p != nullptr || t.row_in != nullptr || t.ev_start != nullptr || t.gram_out != nullptr)) {
        rec->unsupported = true;
        return hipErrorInvalidValue;
    }
    if (t.solo) {
        // every trial of the solve in ONE launch of one workgroup (a few images on one GPU: no accumulators, no row, no exchange)
        if (t.N > SOLO_WAVES || t.final_only || t.row_in != nullptr || t.p2p != nullptr || t.k != 0 || t.max_trials < 1) return hipErrorInvalidValue;
        if (rec != nullptr) { merge_add(rec, MERGE_LM, 1 | 1 << 4 | 1 << 8, 1, SOLO_WAVES * 64, 0, a); return hipSuccess; }
        if (t.ev_start != nullptr && t.ev_stop != nullptr)
            hipExtLaunchKernelGGL((k_lm_trial<FLOW_PRO, false, 1, 1, true>), dim3(1), dim3(SOLO_WAVES * 64), 0, stream, t.ev_start, t.ev_stop, 0, a);
        else
            hipLaunchKernelGGL((k_lm_trial<FLOW_PRO, false, 1, 1, true>), dim3(1), dim3(SOLO_WAVES * 64), 0, stream, a);
        return hipGetLastError();
    }
    // sweep grid: one workgroup per four images up to what the chip keeps resident; beyond that the workgroups loop
    const int wpi = t.final_only ? 1 : (t.wpi == 1 || t.wpi == 2 || t.wpi == 4 ? t.wpi : lm_waves_per_image(t.N, t.P, t.p2p != nullptr && t.xchg_fused));
    const int groups = wpi == 1 ? lm_sweep_blocks(t.N) : (t.N * wpi + WAVES - 1) / WAVES;
    const int grid = t.final_only ? 1 : std::min(groups, lm_max_grid());
    const bool loop = !t.final_only && grid < groups;
    if (rec != nullptr) {
        if (loop) { rec->unsupported = true; return hipErrorInvalidValue; }
        merge_add(rec, MERGE_LM, (t.final_only ? 2 : (t.first ? 0 : 1)) | wpi << 4, grid, WAVES * 64, 0, a);
        return hipSuccess;
    }
    auto launch = [&](auto kernel) {
        if (t.ev_start != nullptr && t.ev_stop != nullptr)   // events bound to this dispatch: its own begin / end
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), 0, stream, t.ev_start, t.ev_stop, 0, a);
        else
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(WAVES * 64), 0, stream, a);
    };
    auto pick = [&](auto flow) {
        constexpr int X = decltype(flow)::value;
        if (t.final_only) launch(k_lm_trial<X, false, 2>);
        else if (wpi == 4) { if (t.first) launch(k_lm_trial<X, false, 0, 4>); else launch(k_lm_trial<X, false, 1, 4>); }
        else if (wpi == 2) { if (t.first) launch(k_lm_trial<X, false, 0, 2>); else launch(k_lm_trial<X, false, 1, 2>); }
        else if (t.first) { if (loop) launch(k_lm_trial<X, true, 0>); else launch(k_lm_trial<X, false, 0>); }
        else { if (loop) launch(k_lm_trial<X, true, 1>); else launch(k_lm_trial<X, false, 1>); }
    };
    if (t.p2p != nullptr) {
        a.p2p = *t.p2p;
        a.xchg_fused = t.xchg_fused;
        pick(std::integral_constant<int, FLOW_PRO_XCHG>());
    } else {
        pick(std::integral_constant<int, FLOW_PRO>());
    }
    return hipGetLastError();
}

hipError_t launch_lm_row(hipStream_t stream, const long long *acc, int R, const LmState *st, int k, double *out) {
    hipLaunchKernelGGL(k_lm_row, dim3(1), dim3(WAVES * 64), 0, stream, acc, R, st, k, out);
    return hipGetLastError();
}

hipError_t launch_lm_exchange(hipStream_t stream, const long long *acc, int R, const LmState *st, double *out, const P2PDev &p,
                              int k) {
    hipLaunchKernelGGL(k_lm_exchange, dim3(1), dim3(WAVES * 64), 0, stream, acc, R, st, out, p, k);
    return hipGetLastError();
}

hipError_t launch_lm_send(hipStream_t stream, const long long *acc, int R, const LmState *st, const P2PDev &p, int k) {
    hipLaunchKernelGGL(k_lm_send, dim3(1), dim3(WAVES * 64), 0, stream, acc, R, st, p, k);
    return hipGetLastError();
}

hipError_t launch_lm_recv(hipStream_t stream, const LmState *st, double *out, const P2PDev &p, int k) {
    hipLaunchKernelGGL(k_lm_recv, dim3(1), dim3(WAVES * 64), 0, stream, st, out, p, k);
    return hipGetLastError();
}

// replicas per accumulator buffer for a grid of `nblocks` workgroups: 8 unless that would put more than
// ACC_MAX_ADDS adds on one address per trial
int lm_acc_replicas(int nblocks) {
    int R = 8;
    while (R < ACC_MAX_REPLICAS && (nblocks + R - 1) / R > ACC_MAX_ADDS) R *= 2;
    return R;
}

}  // namespace ccal
