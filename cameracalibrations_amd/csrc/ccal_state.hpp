// ccal_state.hpp -- device-resident layouts shared by the kernels and the host driver.
//
// HBM layout of one fit (N = images on this GPU, P = corners per image):
//   obj      [3,P]      f32   board template (shared by every image)
//   img      [2,P,N]    f32   observed corners; the only O(N*P) stream: 8 B per corner-residual
//   ext0     [N][8]     f64   initial rvec,tvec (+2 pad) per image (a solve can be re-run from it)
//   blk      [2][N][128] f64   per-image state, double-buffered: trial k writes buffer k & 1; `parity` names the
//                             buffer of the last ACCEPTED point (an evaluating sweep reads it and writes the
//                             other one, so a rejected candidate leaves it intact).  Per image (1 KB):
//                             rvec,tvec | U = C_l^-1 [B^T | h] for the lambda the next step uses
//                             (back-substitution is then one 6x9 mat-vec) | pd flag
//   acc      [3][R][5][68] i64  the reduced intrinsic system of one trial, summed EXACTLY: every workgroup adds
//                             its 68 partial sums as fixed-point limbs with integer atomics (order-independent,
//                             hence bit-deterministic); R replicas spread the adds, three buffers rotate
//                             (trial k adds into k % 3, reads (k - 1) % 3, clears (k + 1) % 3)
//   red      [68]       f64   the reduced system as doubles (what is all-reduced across GPUs)
//   LmState             the OpenCV CvLevMarq state machine, advanced on device
#pragma once

#include <stdint.h>

namespace ccal {

constexpr int NRED = 68;       // doubles in the reduced system (the all-reduced message: 544 B)
constexpr int RED_S = 0;       // S0 = sum (Ji^T Ji - B C_l^-1 B^T)     packed lower 9x9 (45)
constexpr int RED_DA = 45;     // diag(sum Ji^T Ji): Marquardt damping adds lambda * dA to diag(S0)   (9)
constexpr int RED_GS = 54;     // sum (Ji^T e - B C_l^-1 h)                                           (9)
constexpr int RED_ERR = 63;    // sum ||e||^2
constexpr int RED_DNUM = 64;   // sum ||ext_cand - ext_prev||^2
constexpr int RED_DDEN = 65;   // sum ||ext_prev||^2
constexpr int RED_BAD = 66;    // images with a non-PD block or non-finite error (67: pad)

// ---- exact accumulation of the reduced system across workgroups ---------------------------------
// A double is split into (at most two) signed 52-bit limbs of a 260-bit fixed-point number
//     value = sum_j limb_j * 2^(52 j - 150),   |value| < 2^109,  resolution 2^-150
// and added with 64-bit integer atomics: integer addition commutes, so the sum does not depend on
// the order in which workgroups finish -- every rerun, every workgroup and every grid shape sees the
// same bits, and the sum is exact (one rounding, when the limbs are turned back into a double).
// Headroom: a limb holds 52 payload bits, so 2^11 adds per address cannot overflow an int64.
constexpr int ACC_LIMBS = 5;
constexpr int ACC_LIMB_BITS = 52;
constexpr int ACC_LSB_EXP = -150;
constexpr int ACC_ROW = ACC_LIMBS * NRED;   // int64 words per replica, layout [limb][column]
constexpr int ACC_FLAG_COL = 67;            // pad column: limb 0 counts NaN contributions, limb 1 out-of-range ones
constexpr int ACC_BUFS = 3;
constexpr int ACC_MAX_REPLICAS = 64;
constexpr int ACC_MAX_ADDS = 1500;          // adds per address and trial the host allows (2^11 would still be safe)

// per-image block (doubles): two coalesced 512-byte reads per parity fetch everything the next sweep's
// back-substitution needs, already laid out one value per lane so that the 6x10 mat-vec
//     delta_i = u_h - U_B delta_int          U = C_l^-1 [B^T | h],  U[j][k]: j = 0..8 columns of B^T, j = 9: h
// is one multiply and a 4-step DPP sum inside rows of 16 lanes (no LDS transposition):
//   words   0..63   lane 16 k + j: U[j][k] for k = 0..3  (j = 0..9, lanes with j > 9 are zero)
//   words  64..95   lane 16 (k - 4) + j: U[j][k] for k = 4, 5
//   words  96..101  rvec, tvec;   word 102: 1.0 if the damped 6x6 block was positive definite;   rest zero
constexpr int BLK = 128;       // per-image block stride
constexpr int BLK_EXT = 96;    // rvec, tvec (6)
constexpr int BLK_PD = 102;    // pd flag
// A rejected step is re-damped by sweeping the accepted point again with the larger lambda
// (MODE_RESCHUR): rejections are rare, and not keeping the raw 6x16 Gram rows per image
// removes 768 B of writes per image from every sweep.
constexpr int EXT = 8;         // per-image extrinsic stride (doubles)

enum : int { MODE_INIT = 0, MODE_EVAL = 1, MODE_RESCHUR = 2 };   // RESCHUR: re-sweep the accepted point, new lambda
constexpr int MAX_KINDS = 64;

struct LmState {
    // ---- header: the first HOST_WORDS 8-byte words are mirrored into pinned host memory by the
    //      k_lm_trial launch that terminates the solve ----
    int32_t mode;          // what the NEXT sweep does
    int32_t done;
    int32_t parity;        // blk buffer holding the accepted point's blocks
    int32_t lambda_lg10;   // OpenCV lambdaLg10
    int32_t lg_partials;   // log10(lambda) the sweep must use for its Schur sums
    int32_t iters;
    int32_t max_iter;
    int32_t sweeps, reschur, rejected, trials;
    int32_t n_trace, max_trace;
    int32_t n_bad;
    int32_t status;        // 0 ok, CCAL_E_NONFINITE ...
    int32_t P, N_local, nblocks;
    int64_t N_global;
    double eps;
    double aspect;         // > 0 <=> CALIB_FIX_ASPECT_RATIO
    double lam_step;       // lambda of the step being evaluated (back-substitution)
    double lam_next;       // lambda for the Schur sums of this sweep
    double prev_err, rel_step;
    double intr_prev[9];
    int32_t kinds[MAX_KINDS];   // per trial: 1 + mode of the sweep it ran (0 = no-op after `done`)
    // ---- device only ----
    double intr_cand[9], dint[9];
    int32_t mask[9];
    int32_t pad_;
    double pow10[33];           // lambda table 10^(lg), lg = -16..16, filled by the first launch of a solve (lm_boot)
};
constexpr int HOST_WORDS = 64;  // covers everything up to and including kinds[]

// Everything the first launch of a solve needs to (re)build LmState on the device; passed by value as a kernel
// argument so that a whole solve (begin + trials + read-back) is one replayable hipGraph.
struct LmInit {
    int32_t max_iter, max_trace, P, N_local, nblocks;
    int32_t zero_tangent;      // CALIB_ZERO_TANGENT_DIST: p1 = p2 = 0
    int32_t aspect_from_guess; // FIX_ASPECT_RATIO with USE_INTRINSIC_GUESS: aspect = fx0 / fy0
    int32_t pad_;
    int64_t N_global;
    double eps, aspect;
    int32_t mask[9];
    int32_t pad2_;
};

struct TraceRec {          // mirrors ccal_trace in include/ccal.h
    double err, rel_step;
    int32_t accepted, lambda_lg10, iters, pad;
};

}  // namespace ccal
