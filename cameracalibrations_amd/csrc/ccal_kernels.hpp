// ccal_kernels.hpp -- host-callable launchers of the gfx950 kernels (one per .hip file).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "ccal_p2p.hpp"
#include "ccal_state.hpp"

namespace ccal {

// ccal_lm.hip
int lm_sweep_blocks(int N);
int lm_waves_per_image(int N, int P, bool fused);   // 1, 2 or 4: multi-pass boards are split over the waves of a workgroup while that fits the chip
// ---- merged launches (ccal_fit_many): the workgroups of several independent fits in one grid ------------------------------------
// While a recorder is installed on the calling thread, the three launch functions of a steady-state pipeline
// (launch_init_homography, launch_init_extrinsics, launch_lm_trial) append what they would have launched instead of launching it.
// The caller then merges the records of several contexts position by position: one launch_*_multi per kernel instantiation,
// whose workgroups find their own fit's arguments in a device table (args as recorded, one entry per fit).
enum { MERGE_HOMOGRAPHY = 0, MERGE_EXTRINSICS = 1, MERGE_LM = 2 };
struct MergeRecord {
    int kernel = 0;               // MERGE_*
    int key = 0;                  // which instantiation (the launch function's own encoding)
    int grid = 0, block = 0;
    size_t dyn_lds = 0;
    int k = 0, first = 0, final_only = 0;     // MERGE_LM: the per-launch words (everything else is the same for every trial of a solve)
    unsigned char args[512];
    size_t args_bytes = 0;
};
struct MergeRecorder {
    std::vector<MergeRecord> recs;
    bool unsupported = false;     // a launch that has no merged form was attempted: the caller discards the recording
};
MergeRecorder *&merge_recorder();          // this thread's recorder (nullptr: launches launch)
// table: device array of the recorded args (entry = fit), map [grid] fit of each workgroup, first / nblk [fits] (device)
hipError_t launch_lm_trial_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, const int *nblk, int grid,
                                 int k, int is_first, int final_only);
hipError_t launch_init_homography_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, int grid);
hipError_t launch_init_extrinsics_multi(hipStream_t stream, int key, const void *table, const int *map, const int *first, int grid, size_t dyn_lds);

int lm_solo_max_images();        // fits of at most this many images on one GPU run all their trials in ONE launch (TrialLaunch::solo)
int lm_sweep_form(int N, int wpi, bool xchg, bool solo);   // ccal_stats.sweep_form of such a sweep
int lm_max_grid();               // workgroups of k_lm_trial resident on the current device (the sweep grid is capped there)
struct TrialLaunch {              // one launch of k_lm_trial (see ccal_lm.hip)
    const float *obj = nullptr, *img = nullptr;
    double *blk = nullptr;
    LmState *st = nullptr;        // [2]
    long long *acc = nullptr;     // [3][R][ACC_ROW] exact accumulators of the reduced system (ccal_acc.hpp)
    int R = 8;                    // replicas per buffer (lm_acc_replicas)
    const double *row_in = nullptr;   // != NULL: the prologue reads this all-reduced row instead of the accumulators
    TraceRec *trace = nullptr;
    LmState *host_result = nullptr;
    unsigned long long *host_progress = nullptr;
    const double *intr0 = nullptr, *ext0 = nullptr;
    int ext0_stride = EXT;
    LmInit init{};
    double *gram_out = nullptr;
    int N = 0, P = 0, k = 0, first = 0, final_only = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // optional: timestamps of this dispatch
    const P2PDev *p2p = nullptr;  // image-sharded fit: launch the exchanging instantiation
    int xchg_fused = 0;           // ... whose prologue runs the row exchange itself
    int wpi = 0;                  // waves per image as ccal_set_problem resolved it (lm_waves_per_image); 0: resolve at the launch
    int solo = 0;                 // one launch, one workgroup, every trial of the solve (N <= lm_solo_max_images(), single GPU; k = 0):
    int max_trials = 0;           // ... up to this many trials
    double *host_poses = nullptr;       // pinned host memory [N][8]: the launch that finds the solve done writes the accepted poses there
    int *init_fail = nullptr;           // first launch: the init stages' four failure words (device; word 2 = "a pose stage has run since
                                        // the last report", cleared here) and the start intrinsics are
    double *host_init_intr = nullptr;   // copied to pinned host memory by workgroup 0 (all three set, or none)
    int *host_init_fail = nullptr;
};
hipError_t launch_lm_trial(hipStream_t stream, const TrialLaunch &t);
int lm_acc_replicas(int nblocks);
// diagnostics: n_rows rows of 68 doubles added like n_rows workgroups would, read back like a prologue does -> out[68]
hipError_t launch_acc_debug(hipStream_t stream, const double *vals, int n_rows, long long *acc, int R, double *out);
// accumulators of trial k - 1 -> out[0..68) (what RCCL / a host hook then sums across GPUs); no-op after `done`
hipError_t launch_lm_row(hipStream_t stream, const long long *acc, int R, const LmState *st, int k, double *out);
// multi-GPU, many workgroups per rank: accumulators -> one row -> sum over the node's ranks by direct xGMI
// exchange -> out[0..68), as exchange k of the current solve (few workgroups: fused into k_lm_trial)
hipError_t launch_lm_exchange(hipStream_t stream, const long long *acc, int R, const LmState *st, double *out, const P2PDev &p,
                              int k);
// the same exchange as two stream-ordered kernels (shards sharing one in-order stream: all sends before any receive)
hipError_t launch_lm_send(hipStream_t stream, const long long *acc, int R, const LmState *st, const P2PDev &p, int k);
hipError_t launch_lm_recv(hipStream_t stream, const LmState *st, double *out, const P2PDev &p, int k);

// ccal_init.hip -- closed-form initialisation
constexpr int NFOC = 8;   // doubles all-reduced for the focal least squares: AtA(3) Atb(2) fail(1) pad(2)
int init_simd_count();          // SIMDs of the current device: up to one image per SIMD the init kernels run their speculative form
constexpr int LM_FOC_ACC_WORDS = 8 * 5 * 8;   // int64 words of one buffer of exact focal sums: [8 replicas][5 limbs][NFOC]
constexpr int LM_FOC_MAX_IMAGES = 8 * 1500;   // images per GPU whose focal sums the exact accumulators take (adds per address within the limbs' headroom)
// foc_acc != NULL (this GPU holds every image): the kernel also adds each image's focal sums, exactly, into that buffer (zero on
// entry); fail0 != NULL: the kernel zeroes the pose stage's failure counter (no focal kernel will)
hipError_t launch_init_homography(hipStream_t stream, const float *obj, const float *img, double *H_out,
                                  double *foc_partials, long long *foc_acc, int *fail0, int N, int P, int width, int height, bool spec);
// the exact focal sums of `acc` -> 2 x 2 solve -> intr0, fail_count[1] (and fail_count[0] = 0); clears `acc_clear`
hipError_t launch_init_focal_acc(hipStream_t stream, const long long *acc, long long *acc_clear, double *intr0, int *fail_count,
                                 int width, int height, double aspect);
// image-sharded fits: this rank's exact sums as NFOC doubles (what is summed across the ranks); clears `acc_clear`
hipError_t launch_init_focal_acc_row(hipStream_t stream, const long long *acc, long long *acc_clear, double *red_out);
struct InitFocal {                            // pipeline form of the pose stage: the focal solve in its prologue
    const long long *acc;
    long long *acc_clear;
    double *intr0_out;
    int width, height;
    double aspect;
};
hipError_t launch_init_focal_reduce(hipStream_t stream, const double *foc_partials, double *foc_red, int N);
hipError_t launch_init_focal_reduce_solve(hipStream_t stream, const double *foc_partials, double *foc_red, int N, double *intr0,
                                          int *fail_count, int width, int height, double aspect);
hipError_t launch_init_focal_solve(hipStream_t stream, const double *foc_red, double *intr0, int *fail_count,
                                   int width, int height, double aspect);
// H_pix: k_init_homography's output for the same image points (NULL: none); hmode 0 / 1 / 2 see k_init_extrinsics
hipError_t launch_init_extrinsics(hipStream_t stream, const float *obj, const float *img, const double *intr0, const double *H_pix,
                                  int hmode, bool spec, double *ext, int *fail_count, int N, int P, const InitFocal *focal = nullptr);

// ccal_maps.hip -- RowCol <-> XYZ maps and the four RMS metrics
struct MapIntr { double f[2], c[2], k, cs; };   // intrinsic map parameters + checker_size, a kernel argument by value
// [3, M, N] -> [2, M, N] (xyz_stride = 3 M) or one [3, M] template through N poses (xyz_stride = 0); rt [6, N]
// flag != nullptr (one-workgroup launches only, map_is_one_workgroup): the kernel writes flag_value there (system scope) after its results
bool map_is_one_workgroup(int M, int N);
hipError_t launch_map_forward(hipStream_t stream, const double *xyz, long long xyz_stride, double *rc, int M, int N,
                              const MapIntr &in, const double *rt, unsigned long long *flag = nullptr, unsigned long long flag_value = 0);
hipError_t launch_map_inverse(hipStream_t stream, const double *rc, double *xyz, int M, int N, const MapIntr &in,
                              const double *rt, unsigned long long *flag = nullptr, unsigned long long flag_value = 0);
hipError_t launch_signal(hipStream_t stream, unsigned long long *flag, unsigned long long value);   // pinned completion word
// n 32-bit words src -> dst by a kernel on `stream` (src: pinned, device-mapped host memory a small call staged its inputs in)
hipError_t launch_copy_words(hipStream_t stream, const void *src, void *dst, size_t n_words);
struct CopySeg { const uint32_t *src1; uint32_t *dst1; size_t n1; const uint32_t *src2; uint32_t *dst2; size_t n2; };
hipError_t launch_copy_multi(hipStream_t stream, const CopySeg *segs, int n);   // segs: device-visible (pinned, mapped) memory
hipError_t launch_copy_words2(hipStream_t stream, const void *src1, void *dst1, size_t n1, const void *src2, void *dst2, size_t n2);
// intr[9], fail[4] (device) -> pinned host memory, by a kernel on `stream`
hipError_t launch_init_readback(hipStream_t stream, const double *intr, const int *fail, double *host_intr, int *host_fail);
struct RectPar { double v[16]; };   // f0 f1 c0 c1 k rvec(3) tvec(3) checker_size ratio index_offset, by value
hipError_t launch_rectify(hipStream_t stream, const uint8_t *img, int rows, int cols, int ch, long long sr, long long sc,
                          long long sk, int a0, int b0, int out_rows, int out_cols, const RectPar &par, uint8_t *out);
hipError_t launch_errors(hipStream_t stream, const double *obj, const double *img, const MapIntr &in,
                         const double *rt, const double *inv_pts, double *partials, int N, int n1, int n2, int S);

}  // namespace ccal
