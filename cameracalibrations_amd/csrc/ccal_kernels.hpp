// ccal_kernels.hpp -- host-callable launchers of the gfx950 kernels (one per .hip file).
#pragma once

#include <hip/hip_runtime.h>

#include "ccal_state.hpp"

namespace ccal {

// ccal_lm.hip
int lm_sweep_blocks(int N);
hipError_t launch_lm_begin(hipStream_t stream, LmState *st, const LmInit &init, const double *intr0, const double *ext0,
                           int ext0_stride, double *blk);
hipError_t launch_lm_sweep(hipStream_t stream, const float *obj, const float *img, double *blk, const LmState *st,
                           double *partials, double *gram_out, int N, int P);
hipError_t launch_lm_solve(hipStream_t stream, LmState *st, const double *partials, double *red, TraceRec *trace,
                           LmState *host_result, unsigned long long *host_progress, int nblocks, int do_reduce,
                           int do_solve);

// ccal_init.hip -- closed-form initialisation
constexpr int NFOC = 8;   // doubles all-reduced for the focal least squares: AtA(3) Atb(2) fail(1) pad(2)
hipError_t launch_init_homography(hipStream_t stream, const float *obj, const float *img, double *H_out,
                                  double *foc_partials, int N, int P, int width, int height);
hipError_t launch_init_focal_reduce(hipStream_t stream, const double *foc_partials, double *foc_red, int N);
hipError_t launch_init_focal_solve(hipStream_t stream, const double *foc_red, double *intr0, int *fail_count,
                                   int width, int height, double aspect);
hipError_t launch_init_extrinsics(hipStream_t stream, const float *obj, const float *img, const double *intr0,
                                  double *ext, int *fail_count, int N, int P);

// ccal_maps.hip -- RowCol <-> XYZ maps and the four RMS metrics
hipError_t launch_map_forward(hipStream_t stream, const double *xyz, double *rc, int M, const double *par);
hipError_t launch_map_inverse(hipStream_t stream, const double *rc, double *xyz, int M, const double *par);
hipError_t launch_errors(hipStream_t stream, const double *obj, const double *img, const double *par,
                         const double *rt, const double *inv_pts, double *partials, int N, int n1, int n2, int S);

}  // namespace ccal
