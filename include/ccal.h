/*
 * ccal.h -- C ABI of the MI355X-native camera-model fit (libccal_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of yakir12/CameraCalibrations:
 * the Julia function `fit_model` (reference src/detect_fit.jl:27-61), whose body is a
 * single CxxWrap call into OpenCV 4.10 `calibrateCamera` (src/detect_fit.jl:47), plus the
 * RowCol <-> XYZ point maps and error metrics that consume its result
 * (src/meta.jl:27-103, src/buildcalibrations.jl:1-67).  Each entry point below names the
 * reference interface it replaces.  Julia binds these with `ccall`, the tests with
 * Python `ctypes` (INTEGRATION.md shows both stubs).
 *
 * Conventions
 *   - plain C types only; every array is caller-owned, contiguous, Julia column-major:
 *       obj  [3,P]   Float32   board template, first n_corners dim fastest, z = 0
 *                              (src/detect_fit.jl:69, marshalled at :30)
 *       img  [2,P,N] Float32   observed corners, RowCol = (x_cv, y_cv) (src/detect_fit.jl:31)
 *       rvecs/tvecs [3,N] Float64 (src/detect_fit.jl:51-52)
 *   - every function returns an int status: 0 ok, > 0 warning, < 0 error; no exception
 *     ever crosses the boundary (the reference surfaces cv::Exception through CxxWrap).
 *     `ccal_last_error()` gives the thread-local message of the last failure.
 *   - a context is used by one host thread at a time; different contexts are independent
 *     (the reference's FFI wrapper is exercised under Threads.@threads, test/runtests.jl:45-51).
 *   - all arithmetic is FP64 on the device; there is NO CPU fallback: without a usable
 *     gfx950 device every compute entry point fails with CCAL_E_DEVICE.
 */
#ifndef CCAL_H
#define CCAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCAL_VERSION 105 /* 0.1.5: ccal_fit_many / ccal_calibrate_many return the UNION of the warning bits (was: the largest), merged launches join the participants' streams; 0.1.4: ccal_fit_many, ccal_calibrate_many, ccal_detect_corners_many; 0.1.3: ccal_stats.sweep_form (was padding); 0.1.2: ccal_p2p_export2 / ccal_p2p_connect2, ccal_p2p_set_exchange mode 2 */

/* status codes */
#define CCAL_OK            0
#define CCAL_W_MAX_ITER    1   /* stopped on max_iter, not on eps */
#define CCAL_W_RANK_DEFICIENT 2 /* (bit, may come with CCAL_W_MAX_ITER as 3) the fit returned, but in its last sweep the damped 6 x 6
                                 * block of at least one view could not be factored (a zero or non-finite diagonal: ccal_stats.n_bad
                                 * counts the views); such a view's pose is left where it was.  Marquardt's damping keeps every system
                                 * with a positive diagonal positive definite, and a parameter whose column is exactly zero gets a zero
                                 * step -- the minimum-norm solution OpenCV's SVD returns for it -- so on finite, observable data the
                                 * Cholesky / Gauss-Jordan steps here and OpenCV's DECOMP_SVD step solve the same nonsingular system */
#define CCAL_E_ARG        -1   /* bad argument / unsupported flag (rational, thin-prism, tilted models) */
#define CCAL_E_NONPLANAR  -2   /* template not planar with z == 0 and no intrinsic guess (OpenCV throws likewise) */
#define CCAL_E_DEGENERATE -3   /* homography / pose initialisation failed */
#define CCAL_E_NONFINITE  -4   /* non-finite residual or parameter on the device */
#define CCAL_E_DEVICE     -5   /* HIP error, no gfx950 device, out of memory */
#define CCAL_E_COMM       -6   /* the all-reduce hook failed */
#define CCAL_E_STATE      -7   /* call order violated (e.g. LM before initialisation) */

/* OpenCV calib3d flag word, same bit values as the reference builds at src/detect_fit.jl:40 */
#define CCAL_USE_INTRINSIC_GUESS 0x00001
#define CCAL_FIX_ASPECT_RATIO    0x00002
#define CCAL_FIX_PRINCIPAL_POINT 0x00004
#define CCAL_ZERO_TANGENT_DIST   0x00008
#define CCAL_FIX_FOCAL_LENGTH    0x00010
#define CCAL_FIX_K1              0x00020
#define CCAL_FIX_K2              0x00040
#define CCAL_FIX_K3              0x00080
#define CCAL_FIX_TANGENT_DIST    0x200000

#define CCAL_NINTR 9   /* fx fy cx cy k1 k2 p1 p2 k3 */

typedef struct ccal_ctx ccal_ctx;

/* Per-fit statistics (the reference returns only the parameters; OpenCV's rms is discarded
 * at src/detect_fit.jl:47).  Times are host wall-clock around device work, seconds. */
typedef struct ccal_stats {
    double rms;            /* sqrt(sum ||e||^2 / (N_global * P)) at the returned point */
    double final_err;      /* sum ||e||^2 */
    double rel_step;       /* last ||p - prev|| / ||prev|| */
    double t_init_intr_s;  /* A4: homographies + focal least squares */
    double t_init_extr_s;  /* A5: per-view pose initialisation */
    double t_lm_s;         /* A6-A9: global LM loop */
    int32_t iters;         /* accepted LM iterations (OpenCV `iters`) */
    int32_t sweeps;        /* residual+Jacobian sweeps over all corners at new candidate points (+ the initial one) */
    int32_t reschur;       /* further full sweeps that re-damp the accepted point after a rejected step */
    int32_t rejected;      /* rejected trial steps */
    int32_t lambda_lg10;   /* final log10(lambda) */
    int32_t n_bad;         /* images whose 6x6 block was not positive definite in the last sweep */
    int32_t timed_sweeps;  /* sweeps covered by sweep_us (0 unless ccal_set_sweep_timing is on) */
    int32_t sweep_form;    /* which instantiation k_lm_trial<FLOW, LOOP, KIND, WPI, SOLO, MULTI> swept: FLOW | LOOP << 4 | WPI << 8 | SOLO << 12 |
                            * MULTI << 16 (FLOW 1 = row exchange fused in, LOOP = looping grid, WPI = waves per image, SOLO = every trial in
                            * one launch, MULTI = launches shared with other contexts' fits: ccal_fit_many) */
    double sweep_us;       /* mean device time of the sweep kernel, HIP events on the launch stream */
    double solve_kernels_us; /* device time of ALL launches of the solve (every trial + the terminating one), summed;
                              * 0 unless ccal_set_sweep_timing is on */
} ccal_stats;

/* One LM decision, for trajectory parity tests against the oracle. */
typedef struct ccal_trace {
    double err;            /* sum ||e||^2 at the evaluated candidate */
    double rel_step;       /* when accepted, else -1 */
    int32_t accepted;      /* 1 accepted, 0 rejected, 2 initial evaluation */
    int32_t lambda_lg10;   /* after the decision */
    int32_t iters;         /* after the decision */
    int32_t pad;
} ccal_trace;

int         ccal_version(void);
const char *ccal_last_error(void);
/* number of usable gfx950 devices (0 if none); never fails */
int         ccal_device_count(void);

/* ---- context ------------------------------------------------------------------------ */
int  ccal_create(int device, ccal_ctx **out);
void ccal_destroy(ccal_ctx *ctx);
/* run all device work on the caller's hipStream_t (e.g. PyTorch's current stream);
 * NULL restores the context's own stream. */
int  ccal_set_stream(ccal_ctx *ctx, void *hip_stream);

/* Multi-GPU: images are sharded one-extrinsic-block-per-GPU.  The hook must
 * sum `n` doubles in place across ranks on `hip_stream` (RCCL all-reduce).  world_size == 1
 * or fn == NULL disables it.  n_images_global = total images over all ranks. */
typedef int (*ccal_allreduce_fn)(void *user, void *device_buf, int32_t n, void *hip_stream);
int  ccal_set_allreduce(ccal_ctx *ctx, ccal_allreduce_fn fn, void *user, int32_t world_size,
                        int32_t rank, int64_t n_images_global);

/* Native variant: the library calls RCCL itself (ncclAllReduce on the launch stream; RCCL is
 * resolved at run time).  Rank 0 obtains an id with ccal_comm_unique_id and ships the 128 bytes to
 * the other ranks by any means (the Python mirror broadcasts it through torch.distributed);
 * every rank then calls ccal_comm_init_rank.  Replaces any hook set before. */
int  ccal_comm_unique_id(char id_out[128]);
int  ccal_comm_init_rank(ccal_ctx *ctx, int32_t world_size, int32_t rank, const char id[128],
                         int64_t n_images_global);

/* Direct variant for the GPUs of ONE node (the default of bench.py): no collective library at
 * all.  The per-trial message is one 544-byte row, so its cost is pure latency; each rank owns an
 * inbox in fine-grained device memory, peers map it through HIP IPC and store their row straight
 * into it over xGMI ({epoch | 32 payload bits} words), the receiver sums the rows in rank order
 * (bit-identical on every rank).  The exchange is a kernel on the launch stream, so a sharded
 * solve is one hipGraph like a single-GPU one.  Every rank calls ccal_p2p_export (64-byte
 * handle out), the handles are all-gathered by any means (rank order, world x 64 bytes), then
 * every rank calls ccal_p2p_connect.  At most 16 ranks.  A peer that does not show up within
 * CCAL_P2P_TIMEOUT_MS (default 5000) makes the solve return CCAL_E_COMM instead of hanging.
 * Replaces any hook / RCCL communicator set before. */
int  ccal_p2p_export(ccal_ctx *ctx, int32_t world_size, int32_t rank, char handle_out[64]);
int  ccal_p2p_connect(ccal_ctx *ctx, const char *handles, int64_t n_images_global);
/* The same pair with the rank's device id inside the blob (64-byte inbox handle + 16-byte device uuid; all-gather world x 80
 * bytes): the library then sees for itself when two ranks share a GPU and takes the one-workgroup exchange (see
 * ccal_p2p_set_exchange) without the caller's help.  ccal_p2p_shared_device: 1 if ccal_p2p_connect2 found such a pair. */
#define CCAL_P2P_BLOB_BYTES 80
int  ccal_p2p_export2(ccal_ctx *ctx, int32_t world_size, int32_t rank, char blob_out[CCAL_P2P_BLOB_BYTES]);
int  ccal_p2p_connect2(ccal_ctx *ctx, const char *blobs, int64_t n_images_global);
int  ccal_p2p_shared_device(ccal_ctx *ctx);
/* `rounds` exchanges of known rows, checked on the host (every rank must call it with the same
 * count): CCAL_OK, or CCAL_E_COMM when a peer stayed silent or a sum came back wrong. */
int  ccal_p2p_selftest(ccal_ctx *ctx, int32_t rounds);
/* How a rank takes part in the exchange.  mode 0 (default): automatic -- fused into the prologue of every
 * trial's kernel (all its workgroups poll the inbox) while a rank runs few workgroups, one exchange workgroup
 * per trial otherwise.  mode 1: always the one-workgroup exchange; REQUIRED on every rank when ranks of several
 * processes share one GPU (rehearsals, time-shared devices): a rank that waits then holds one compute unit
 * instead of a whole grid that keeps the sender off the chip.  Same wire format either way; call it with the same
 * value on every rank, before the first solve.  With ccal_p2p_export2 / ccal_p2p_connect2 mode 0 already includes the
 * detection (two equal device ids in the blobs = a shared device); mode 2 keeps the fused form even then (tests of the
 * fused kernels on a one-GPU box).  ccal_device_uuid serves callers of the 64-byte pair. */
int  ccal_p2p_set_exchange(ccal_ctx *ctx, int32_t mode);
int  ccal_device_uuid(int32_t device, char uuid_out[16]);

/* ---- problem ------------------------------------------------------------------------ */
/* Replaces the marshalling of fit_model (src/detect_fit.jl:30-45): template, image size
 * (width, height) as OpenCV sees it == Julia size(img) (src/detect_fit.jl:32), aspect
 * (cameraMatrix(0,0)/cameraMatrix(1,1), src/detect_fit.jl:34-36), flag word (:40) and
 * TermCriteria (src/CameraCalibrations.jl:16).  `obj` is a host pointer. */
int  ccal_set_problem(ccal_ctx *ctx, const float *obj, int32_t P, int32_t N, int32_t width,
                      int32_t height, double aspect, uint32_t flags, int32_t max_iter, double eps);
/* observed corners [2,P,N]: host copy (drop-in path) or a device pointer that already
 * lives in HBM (bench / torch tensors); the device pointer must stay valid until replaced. */
int  ccal_set_image_points(ccal_ctx *ctx, const float *img_host);
int  ccal_set_image_points_device(ccal_ctx *ctx, const float *img_device);

/* ---- stages of calibrateCamera (src/detect_fit.jl:47) -------------------------------- */
/* A4 initial intrinsics from per-image homographies; or supply them (USE_INTRINSIC_GUESS). */
int  ccal_init_intrinsics(ccal_ctx *ctx);
int  ccal_set_intrinsics(ccal_ctx *ctx, const double intr[CCAL_NINTR]);
/* A5 per-view pose initialisation; or supply [6,N] = rvec,tvec per image. */
int  ccal_init_extrinsics(ccal_ctx *ctx);
int  ccal_set_extrinsics(ccal_ctx *ctx, const double *rt_host);
/* A6-A9 global Levenberg-Marquardt with OpenCV's schedule, on the device. */
int  ccal_lm_run(ccal_ctx *ctx);
/* A solve is replayed as one hipGraph (default; CCAL_NO_GRAPH=1 in the environment turns that off at ccal_create) or
 * launched kernel by kernel: same kernels, same results, different host cost. */
int  ccal_set_graph(ccal_ctx *ctx, int32_t enable);
/* How ccal_init_extrinsics obtains the board -> normalised-image homography OpenCV's cvFindExtrinsicCameraParams2 computes before
 * its pose polish, when the intrinsics stage has run on the same corners (otherwise always mode 0):
 *   0  OpenCV's own sequence: DLT + LMSolver polish on the normalised points
 *   1  (default) the same sequence with the DLT's result taken from the intrinsics stage's pixel-space DLT (the two DLTs solve the
 *      same per-axis-normalised system): OpenCV's trajectory to ~1e-10
 *   2  the intrinsics stage's POLISHED pixel-space homography decomposed directly, no second polish: 15 % less time in the pose
 *      stage; views whose 6-parameter polish converges end at the same pose (1e-11), a view that exhausts CvLevMarq's 20
 *      iterations ends where they take it from the slightly different start (seen: 2e-4 rad on one view of a model-mismatched
 *      set, 8e-7 px in the RMS of the fit that follows) -- not OpenCV's trajectory for such views */
int  ccal_set_pose_init(ccal_ctx *ctx, int32_t mode);
/* A4 + A5 + A6-A9 as ONE pipeline on the problem and image points already set (e.g. a device pointer): the stages are
 * enqueued back to back and the host waits once.  Same results as the three calls above; their per-stage times in
 * ccal_stats then only cover the enqueueing. */
int  ccal_fit(ccal_ctx *ctx);
/* ccal_fit for n INDEPENDENT problems, one context each (a caller that calibrates many cameras: the loop around
 * src/buildcalibrations.jl:12-26): every context's pipeline is enqueued on its own stream before any is waited for, so the
 * GPU overlaps them -- a fit of a few images occupies a few compute units and spends most of its time in launch and
 * hand-off latency.  Contexts may live on different devices; shards of a joint fit are refused (CCAL_E_STATE).  Each
 * context ends exactly as after its own ccal_fit (same bits; results through ccal_get_result).  Four or more contexts of one
 * device whose fits are small (at most 128 workgroups per sweep: up to 128 images of a one-pass board) and in their steady state
 * (a fit before on the same context) do not even get launches of their own: their kernels are merged into one launch per
 * pipeline stage whose workgroups look their own fit up in a table (ccal_stats.sweep_form bit 16; CCAL_NO_MERGE=1 disables).  status_out[i] (may be
 * NULL) = what ccal_fit would have returned for context i; the return value is the first failure (ccal_last_error() then names
 * that context), else the UNION of the contexts' warning bits (CCAL_W_*). */
int  ccal_fit_many(ccal_ctx *const *ctxs, int32_t n, int32_t *status_out);
/* ccal_calibrate (= fit_model, src/detect_fit.jl:27-61) for n calibrations of one caller: template obj [P][3], image size, aspect,
 * flags and criteria shared; img[i] -> [N[i]][P][2] float32 corners of calibration i; results packed per calibration into
 * out[i]: K4 (4) | dist5 (5) | rms | iters | rvecs [N[i]][3] | tvecs [N[i]][3]   (11 + 6 N[i] doubles).
 * Host arrays in and out; contexts come from a pool of the calling thread (released by ccal_thread_cleanup / at thread exit);
 * the fits run through ccal_fit_many.  Same values as n calls of ccal_calibrate.  status_out[i] (may be NULL) per calibration;
 * returns the first failure (ccal_last_error() = that calibration's own message), else the union of the warning bits.
 * CCAL_USE_INTRINSIC_GUESS is not supported here (CCAL_E_ARG). */
int  ccal_calibrate_many(const float *obj, int32_t P, int32_t n, const float *const *img, const int32_t *N, int32_t width,
                         int32_t height, double aspect, uint32_t flags, int32_t max_iter, double eps, int32_t device,
                         double *const *out, int32_t *status_out);
/* Bracket every sweep launch of the next ccal_lm_run calls with a HIP event pair on the launch
 * stream and report the mean in ccal_stats.sweep_us (roofline figure of bench.py). */
int  ccal_set_sweep_timing(ccal_ctx *ctx, int32_t enable);
/* results (any pointer may be NULL) */
int  ccal_get_result(ccal_ctx *ctx, double intr[CCAL_NINTR], double *rvecs, double *tvecs,
                     ccal_stats *stats);
int  ccal_get_trace(ccal_ctx *ctx, ccal_trace *out, int32_t max_entries, int32_t *n_entries);

/* ---- one-shot drop-in for fit_model / OpenCV.calibrateCamera ------------------------- */
/* Replaces src/detect_fit.jl:47 including its un-marshalling (:49-60): K4 = fx fy cx cy
 * (frow fcol crow ccol), dist5 = k1 k2 p1 p2 k3 (`k = first(dist)`), rvecs/tvecs [3,N]. */
int  ccal_calibrate(const float *obj, const float *img, int32_t P, int32_t N, int32_t width,
                    int32_t height, double aspect, uint32_t flags, int32_t max_iter, double eps,
                    int32_t device, double K4[4], double dist5[5], double *rvecs, double *tvecs,
                    double *rms, int32_t *iters, ccal_stats *stats);

/* The same call over several GPUs of one node from ONE host thread (the form a Julia caller wants:
 * SURVEY.md 8b `n_dev, dev_ids`): images are sharded contiguously over dev_ids[0..n_dev), every device
 * runs the same kernels as in the one-process-per-GPU form and the reduced intrinsic system travels by
 * the direct xGMI row exchange (the inboxes are plain peer-mapped pointers here, no IPC).  Every stage
 * is enqueued on all devices before the host waits for any.  img / rvecs / tvecs are the full [.,.,N] /
 * [3,N] host arrays; results are identical on every device and taken from dev_ids[0].  1..16 devices,
 * N >= n_dev; a device id may appear more than once (several shards on one GPU -- how a one-GPU box rehearses
 * 2/4/8-way sharding): all shards of a device then run on ONE stream and the exchange becomes two kernels,
 * every shard's send enqueued before any shard's receive, so nothing ever waits for work queued behind it. */
int  ccal_calibrate_multi(const float *obj, const float *img, int32_t P, int32_t N, int32_t width,
                          int32_t height, double aspect, uint32_t flags, int32_t max_iter, double eps,
                          int32_t n_dev, const int32_t *dev_ids, double K4[4], double dist5[5],
                          double *rvecs, double *tvecs, double *rms, int32_t *iters, ccal_stats *stats);

/* ccal_calibrate keeps one context per (host thread, device) alive between calls (creating a
 * stream, the device buffers and the launch graph costs more than a small fit); this frees the
 * calling thread's cached contexts. */
void ccal_thread_cleanup(void);

/* ---- point maps and error metrics (src/meta.jl, src/buildcalibrations.jl) ------------ */
/* real2image[i] (src/meta.jl:27-33,39-44,88): xyz [3,M] -> rowcol [2,M], host arrays. */
int  ccal_project(ccal_ctx *ctx, const double *xyz, int32_t M, const double f[2], const double c[2],
                  double k, const double rvec[3], const double tvec[3], double checker_size,
                  double *rowcol);
/* image2real[i] (src/meta.jl:50-76,82): rowcol [2,M] -> xyz [3,M] on the board plane. */
int  ccal_unproject(ccal_ctx *ctx, const double *rowcol, int32_t M, const double f[2],
                    const double c[2], double k, const double rvec[3], const double tvec[3],
                    double checker_size, double *xyz);
/* The same maps for N images in ONE call (what `c.(objpoints, i)` over all i does in calculate_errors and the
 * plots, src/buildcalibrations.jl:29,46): image n goes through pose n of rvecs/tvecs [3,N].
 * ccal_project_many: xyz is [3,M,N] (per_image != 0) or one [3,M] template mapped through every pose
 * (per_image == 0); rowcol [2,M,N].  ccal_unproject_many: rowcol [2,M,N] -> xyz [3,M,N].
 * None of the map / metric / warp entry points allocates device memory per call (the context keeps
 * grow-only scratch); calls of up to a few thousand points do not copy at all (pinned staging). */
int  ccal_project_many(ccal_ctx *ctx, const double *xyz, int32_t M, int32_t N, int32_t per_image,
                       const double f[2], const double c[2], double k, const double *rvecs,
                       const double *tvecs, double checker_size, double *rowcol);
int  ccal_unproject_many(ccal_ctx *ctx, const double *rowcol, int32_t M, int32_t N, const double f[2],
                         const double c[2], double k, const double *rvecs, const double *tvecs,
                         double checker_size, double *xyz);
/* calculate_errors (src/buildcalibrations.jl:37-67): out = reprojection, projection,
 * distance, inverse.  obj_scaled [3,P] Float64 (template * checker_size), img [2,P,N] Float64,
 * inv_pts [2,S,N] caller-supplied round-trip samples (the reference uses the global RNG). */
int  ccal_errors(ccal_ctx *ctx, int32_t N, int32_t n1, int32_t n2, const double *obj_scaled,
                 const double *img, const double f[2], const double c[2], double k,
                 const double *rvecs, const double *tvecs, double checker_size, int32_t S,
                 const double *inv_pts, double out[4]);

/* ---- rectified image -------------------------------------------- */
/* `warp(img, tform, axs)` of src/plot_calibration.jl:15-22,40 with tform = real2image[i] o push(.,0) o
 * inv(ratio I): out[a - a0, b - b0, :] = img at real2image((a, b, 0) / ratio) - index_offset, bilinear,
 * 0 outside.  img: 8-bit, element (row, col, ch) at img[row*stride_row + col*stride_col + ch*stride_ch];
 * out: [out_rows, out_cols, channels] C order.  index_offset = 1 reproduces the reference, which uses
 * the 0-based RowCol of the fit as 1-based Julia indices. */
int  ccal_rectify_image(ccal_ctx *ctx, const uint8_t *img, int32_t rows, int32_t cols, int32_t channels,
                        int64_t stride_row, int64_t stride_col, int64_t stride_ch, const double f[2],
                        const double c[2], double k, const double rvec[3], const double tvec[3],
                        double checker_size, double ratio, int32_t a0, int32_t b0, int32_t out_rows,
                        int32_t out_cols, double index_offset, uint8_t *out);

/* ---- corner detection (next row after the fit: SURVEY.md 8f) -------------------------- */
/* Replaces `_detect_corners` (src/detect_fit.jl:5-21: OpenCV findChessboardCorners + cornerSubPix)
 * for one 8-bit gray image.  Element (row, col) of the Julia image is gray[row * stride_row +
 * col * stride_col] -- Julia hands OpenCV the column-major matrix (stride_row = 1, stride_col =
 * rows; src/detect_fit.jl:7), numpy uses (cols, 1).  corners [2, n1*n2] Float64 = RowCol (row, col),
 * 0-based pixel centres, first n_corners dimension fastest, in OpenCV's corner order; *found = 0
 * is the reference's `missing`.  stage_out: NULL, or 2*rows*cols doubles (blurred image, response)
 * for stage-wise tests. */
int  ccal_detect_corners(ccal_ctx *ctx, const uint8_t *gray, int32_t rows, int32_t cols, int64_t stride_row,
                         int64_t stride_col, int32_t n1, int32_t n2, double *corners, int32_t *found,
                         double *stage_out);

/* The same for the n images of one calibration (one size, one layout): three images are kept in flight, so that the host's share
 * of an image (staging, peak sort, lattice search) overlaps the device's share of the next ones.  corners[i] [2, n1*n2],
 * found[i] as above; the same corners as n calls of ccal_detect_corners.  The images must stay valid until the call returns. */
int  ccal_detect_corners_many(ccal_ctx *ctx, const uint8_t *const *gray, int32_t n, int32_t rows, int32_t cols, int64_t stride_row,
                              int64_t stride_col, int32_t n1, int32_t n2, double *const *corners, int32_t *found);

/* ---- diagnostics (stage-wise parity tests; not part of the reference surface) -------- */
/* per-image 16x16 Gram matrices [Je|Ji|e]^T[Je|Ji|e] at the given point -> G [256,N] host */
int  ccal_debug_image_grams(ccal_ctx *ctx, const double intr[CCAL_NINTR], const double *rt_host,
                            double *G_host);
/* the exact accumulators on their own: `n_rows` rows of 68 doubles are added as n_rows workgroups of a sweep would
 * add them (integer atomics into `replicas` replicas), then read and decoded as a trial's prologue does.
 * out[0..67] = the column sums (the last column carries no value; the error column, 63, turns NaN / +inf when a
 * NaN / out-of-range value was added). */
int  ccal_debug_acc_sum(ccal_ctx *ctx, const double *rows, int32_t n_rows, int32_t replicas, double out[68]);
/* the per-image homographies the intrinsics stage (ccal_init_intrinsics) left on the device: H[N][18] row-major, the polished
 * board -> pixel homography (9, H[8] = 1) followed by the normalised DLT's result before its polish (9).  CCAL_E_STATE when no
 * intrinsics stage has run on the current corners.  (tools/fuzz_case.py: where an initial focal length comes from) */
int  ccal_debug_get_homographies(ccal_ctx *ctx, double *H_host);
/* the xGMI row exchange on its own, at any world size on one GPU: `world` (1..16) workgroups of ONE launch act as the ranks of a
 * node (rank r = workgroup r, its own inbox in this device's memory) and run `rounds` consecutive exchanges numbered the way a
 * series of solves numbers them (three trials per solve, then the next sequence number, starting at seq0 -- 0xFFFD.. crosses the
 * 16-bit wrap of the sequence field in the tag).  rows / sums: [rounds][world][68] doubles on the host; sums[rd][r] = what rank
 * r holds after exchange rd = the rank-ordered sum of rows[rd][0..world). */
int  ccal_debug_p2p_world(ccal_ctx *ctx, int32_t world, int32_t rounds, uint32_t seq0, const double *rows, double *sums);
/* overwrite the solve sequence number of every shard of this thread's cached ccal_calibrate_multi device set (same n_dev /
 * dev_ids as the call that created it): lets a test run sharded solves across the tag's 16-bit wrap */
int  ccal_debug_multi_set_epoch(int32_t n_dev, const int32_t *dev_ids, uint32_t epoch);
/* current initial values held by the context: intr[9], rt [6,N] */
int  ccal_get_init(ccal_ctx *ctx, double intr[CCAL_NINTR], double *rt_host);
/* average device time (microseconds, HIP events on the context's stream) of `reps` launches
 * of the residual+Jacobian sweep kernel at the current point; used by bench.py for the
 * roofline figure.  Does not change the LM state. */
int  ccal_time_sweep(ccal_ctx *ctx, int32_t reps, double *avg_us);

#ifdef __cplusplus
}
#endif
#endif /* CCAL_H */
