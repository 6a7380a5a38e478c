// ccal_maps.hip -- RowCol <-> XYZ point maps and the four RMS error metrics on gfx950.
//
// Batched forms of the reference's per-point closures:
//   real2image[i]  src/meta.jl:27-33,39-44,88   + obj2img  src/buildcalibrations.jl:1-6
//   image2real[i]  src/meta.jl:50-76,82         (closed-form cubic instead of Polynomials.roots)
//   calculate_errors  src/buildcalibrations.jl:28-31,37-67
// All FP64; one lane per point, coalesced [3,M] / [2,M] streams.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdint>

#include "ccal_kernels.hpp"
#include "ccal_math.hpp"

namespace ccal {

// Intrinsic part of a map, passed BY VALUE as a kernel argument (no upload of its own): f, c, k, checker_size.
// The extrinsic part is one (rvec, tvec) per image: rt [6, N] in device (or device-visible pinned) memory.
// Point streams are [3, M, N] / [2, M, N] (image index = blockIdx.y); xyz_stride = 0 maps ONE [3, M] template
// through every image's pose (what calculate_errors and the plots do, src/buildcalibrations.jl:29).
// (the image index strides over gridDim.y: a grid's y extent is capped at 65535, the image count is not)
// flag (one-workgroup launches on pinned buffers only): the completion word the host spins on, written by the map kernel itself once
// its results have left the device -- a point-wise call is then ONE launch (no k_signal behind it).
__device__ __forceinline__ void map_done(unsigned long long *flag, unsigned long long value) {
    if (flag == nullptr) return;                  // (uniform)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_map_forward(const double *__restrict__ xyz, long long xyz_stride, double *__restrict__ rc,
                                                      int M, int N, const MapIntr in, const double *__restrict__ rt,
                                                      unsigned long long *flag, unsigned long long flag_value) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    for (int img = blockIdx.y; i < M && img < N; img += gridDim.y) {
        const double *e = rt + (size_t)6 * img;
        const double r[3] = {e[0], e[1], e[2]}, t[3] = {e[3], e[4], e[5]};
        double R[9];
        rodrigues_R(r, R);
        const double *src = xyz + (size_t)img * xyz_stride;
        const double p[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
        double o[2];
        map_forward(p, in.f, in.c, in.k, R, t, 1.0 / in.cs, o);
        double *dst = rc + (size_t)2 * M * img;
        dst[2 * i] = o[0];
        dst[2 * i + 1] = o[1];
    }
    map_done(flag, flag_value);
}

__global__ __launch_bounds__(256) void k_map_inverse(const double *__restrict__ rc, double *__restrict__ xyz, int M, int N,
                                                      const MapIntr in, const double *__restrict__ rt,
                                                      unsigned long long *flag, unsigned long long flag_value) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    for (int img = blockIdx.y; i < M && img < N; img += gridDim.y) {
        const double *e = rt + (size_t)6 * img;
        const double r[3] = {e[0], e[1], e[2]}, t[3] = {e[3], e[4], e[5]};
        double R[9], ti[3];
        rodrigues_R(r, R);
#pragma unroll
        for (int a = 0; a < 3; a++) ti[a] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
        const double *src = rc + (size_t)2 * M * img;
        const double q[2] = {src[2 * i], src[2 * i + 1]};
        double o[3];
        map_inverse(q, in.f, in.c, in.k, R, ti, in.cs, o);
        double *dst = xyz + (size_t)3 * M * img;
        dst[3 * i] = o[0];
        dst[3 * i + 1] = o[1];
        dst[3 * i + 2] = o[2];
    }
    map_done(flag, flag_value);
}

// One workgroup per image: the four sums of calculate_errors for that image.
// rt [6,N];  partials [4,N]
__global__ __launch_bounds__(256) void k_errors(const double *__restrict__ obj, const double *__restrict__ img,
                                                 const MapIntr in, const double *__restrict__ rt,
                                                 const double *__restrict__ inv_pts, double *__restrict__ partials,
                                                 int n1, int n2, int S) {
    extern __shared__ __attribute__((aligned(16))) double s_xyz[];   // [3,P] projected corners
    __shared__ double s_acc[4][256];
    const int i = blockIdx.x, tid = threadIdx.x, P = n1 * n2;
    const double f[2] = {in.f[0], in.f[1]}, c[2] = {in.c[0], in.c[1]}, k = in.k, cs = in.cs;
    const double r[3] = {rt[6 * i], rt[6 * i + 1], rt[6 * i + 2]};
    const double t[3] = {rt[6 * i + 3], rt[6 * i + 4], rt[6 * i + 5]};
    double R[9], ti[3];
    rodrigues_R(r, R);
#pragma unroll
    for (int a = 0; a < 3; a++) ti[a] = -(R[a] * t[0] + R[3 + a] * t[1] + R[6 + a] * t[2]);
    double reproj = 0.0, proj = 0.0, dist = 0.0, inv = 0.0;
    const double *mi = img + (size_t)2 * P * i;
    for (int p = tid; p < P; p += 256) {
        const double o[3] = {obj[3 * p], obj[3 * p + 1], obj[3 * p + 2]};
        const double m[2] = {mi[2 * p], mi[2 * p + 1]};
        double rc[2], x[3];
        map_forward(o, f, c, k, R, t, 1.0 / cs, rc);                       // _reprojection, :28-31
        reproj += (rc[0] - m[0]) * (rc[0] - m[0]) + (rc[1] - m[1]) * (rc[1] - m[1]);
        map_inverse(m, f, c, k, R, ti, cs, x);                             // projected = c.(imgpoints, i), :46
        proj += (x[0] - o[0]) * (x[0] - o[0]) + (x[1] - o[1]) * (x[1] - o[1]) + (x[2] - o[2]) * (x[2] - o[2]);
        s_xyz[3 * p] = x[0]; s_xyz[3 * p + 1] = x[1]; s_xyz[3 * p + 2] = x[2];
    }
    __syncthreads();
    for (int p = tid; p < P; p += 256) {                                   // diff(projected; dims), :49-51
        const int a = p % n1, b = p / n1;
        const double *u = s_xyz + 3 * p;
        if (a + 1 < n1) {
            const double *w = u + 3;
            const double d = sqrt((w[0] - u[0]) * (w[0] - u[0]) + (w[1] - u[1]) * (w[1] - u[1]) +
                                  (w[2] - u[2]) * (w[2] - u[2])) - cs;
            dist += d * d;
        }
        if (b + 1 < n2) {
            const double *w = u + 3 * n1;
            const double d = sqrt((w[0] - u[0]) * (w[0] - u[0]) + (w[1] - u[1]) * (w[1] - u[1]) +
                                  (w[2] - u[2]) * (w[2] - u[2])) - cs;
            dist += d * d;
        }
    }
    if (inv_pts != nullptr) {
        const double *pts = inv_pts + (size_t)2 * S * i;
        for (int q = tid; q < S; q += 256) {                               // round trip, :53-58
            const double m[2] = {pts[2 * q], pts[2 * q + 1]};
            double x[3], rc[2];
            map_inverse(m, f, c, k, R, ti, cs, x);
            map_forward(x, f, c, k, R, t, 1.0 / cs, rc);
            inv += (m[0] - rc[0]) * (m[0] - rc[0]) + (m[1] - rc[1]) * (m[1] - rc[1]);
        }
    }
    s_acc[0][tid] = reproj; s_acc[1][tid] = proj; s_acc[2][tid] = dist; s_acc[3][tid] = inv;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {                                    // fixed-shape tree: deterministic
        if (tid < w) {
#pragma unroll
            for (int j = 0; j < 4; j++) s_acc[j][tid] += s_acc[j][tid + w];
        }
        __syncthreads();
    }
    if (tid < 4) partials[(size_t)4 * i + tid] = s_acc[tid][0];
}

// fixed-order sum of the per-image partials [4,N] -> out[4]: thread (metric m, slice s) adds the images i = s, s + 64,
// s + 128, ... in order, the 64 slice sums of a metric are then added in slice order (one thread per metric) -- the same
// bits on every run for a given N, coalesced 2-KB reads, N / 64 dependent adds instead of N
__global__ __launch_bounds__(256) void k_errors_reduce(const double *__restrict__ partials, double *__restrict__ out,
                                                       int N) {
    __shared__ double s_slice[64][4];
    const int tid = threadIdx.x, m = tid & 3, sl = tid >> 2;
    double s = 0.0;
    for (int i = sl; i < N; i += 64) s += partials[(size_t)4 * i + m];
    s_slice[sl][m] = s;
    __syncthreads();
    if (tid < 4) {
        double t = 0.0;
        for (int k = 0; k < 64; k++) t += s_slice[k][tid];
        out[tid] = t;
    }
}

// Rectified view of one image (reference src/plot_calibration.jl:15-22,40: `warp(img, tform, axs)` with
// tform = real2image[i] o push(.,0) o inv(ratio I)): output pixel (a, b) of the axes a0.., b0.. shows
// the image at real2image((a, b, 0) / ratio), bilinear, `fill` outside.  par = f0 f1 c0 c1 k r t(6)
// checker_size ratio index_offset (by value).  One lane per output pixel and channel triple.
__global__ __launch_bounds__(256) void k_rectify(const uint8_t *__restrict__ img, int rows, int cols, int ch,
                                                  long long sr, long long sc, long long sk, int a0, int b0, int out_rows,
                                                  int out_cols, const RectPar rp, uint8_t *__restrict__ out) {
    const double *par = rp.v;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= out_cols) return;
    const double f[2] = {par[0], par[1]}, c[2] = {par[2], par[3]};
    const double r[3] = {par[5], par[6], par[7]}, t[3] = {par[8], par[9], par[10]};
    const double ratio = par[12], off = par[13];
    double R[9];
    rodrigues_R(r, R);
    const double w[3] = {(double)(a0 + i) / ratio, (double)(b0 + j) / ratio, 0.0};
    double rc[2];
    map_forward(w, f, c, par[4], R, t, 1.0 / par[11], rc);
    const double y = rc[0] - off, x = rc[1] - off;            // 0-based (row, col) into the array
    const bool inside = y >= 0.0 && y <= (double)(rows - 1) && x >= 0.0 && x <= (double)(cols - 1);
    const int y0 = inside ? (int)floor(y) : 0, x0 = inside ? (int)floor(x) : 0;
    const int y1 = min(y0 + 1, rows - 1), x1 = min(x0 + 1, cols - 1);
    const double fy = y - y0, fx = x - x0;
    for (int k = 0; k < ch; k++) {
        double v = 0.0;
        if (inside) {
            const double p00 = img[y0 * sr + x0 * sc + k * sk], p01 = img[y0 * sr + x1 * sc + k * sk];
            const double p10 = img[y1 * sr + x0 * sc + k * sk], p11 = img[y1 * sr + x1 * sc + k * sk];
            v = (1 - fy) * ((1 - fx) * p00 + fx * p01) + fy * ((1 - fx) * p10 + fx * p11);
        }
        out[((size_t)i * out_cols + j) * ch + k] = (uint8_t)fmin(fmax(floor(v + 0.5), 0.0), 255.0);
    }
}

hipError_t launch_rectify(hipStream_t stream, const uint8_t *img, int rows, int cols, int ch, long long sr, long long sc,
                          long long sk, int a0, int b0, int out_rows, int out_cols, const RectPar &par, uint8_t *out) {
    hipLaunchKernelGGL(k_rectify, dim3((out_cols + 255) / 256, out_rows), dim3(256), 0, stream, img, rows, cols, ch, sr, sc,
                       sk, a0, b0, out_rows, out_cols, par, out);
    return hipGetLastError();
}

// completion marker for the zero-copy path: stream order puts it behind the map kernel, whose stores to pinned
// host memory have left the device when it ends; the host spins on this word instead of paying the wake-up
// latency of hipStreamSynchronize
__global__ void k_signal(unsigned long long *flag, unsigned long long value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// read-back of the initialisation's nine intrinsic values and four failure words straight into pinned host memory by a
// kernel (one launch instead of two copy commands); the completion word that follows it (k_signal) releases them
__global__ void k_init_readback(const double *__restrict__ intr, const int *__restrict__ fail, double *host_intr, int *host_fail) {
    const int t = threadIdx.x;
    if (t < 9) __hip_atomic_store(host_intr + t, intr[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (t < 13) __hip_atomic_store(host_fail + (t - 9), fail[t - 9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_init_readback(hipStream_t stream, const double *intr, const int *fail, double *host_intr, int *host_fail) {
    hipLaunchKernelGGL(k_init_readback, dim3(1), dim3(64), 0, stream, intr, fail, host_intr, host_fail);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_copy_words(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// two segments in one launch (a small drop-in call's template and corners, both staged in the pinned block)
__global__ __launch_bounds__(256) void k_copy_words2(const uint32_t *__restrict__ src1, uint32_t *__restrict__ dst1, size_t n1,
                                                      const uint32_t *__restrict__ src2, uint32_t *__restrict__ dst2, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1 + n2; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n1) dst1[i] = src1[i];
        else dst2[i - n1] = src2[i - n1];
    }
}

hipError_t launch_copy_words2(hipStream_t stream, const void *src1, void *dst1, size_t n1, const void *src2, void *dst2, size_t n2) {
    if (n1 + n2 == 0) return hipSuccess;
    const int grid = (int)std::min<size_t>((n1 + n2 + 255) / 256, 64);
    hipLaunchKernelGGL(k_copy_words2, dim3(grid), dim3(256), 0, stream, static_cast<const uint32_t *>(src1), static_cast<uint32_t *>(dst1), n1,
                       static_cast<const uint32_t *>(src2), static_cast<uint32_t *>(dst2), n2);
    return hipGetLastError();
}

// the staged inputs of many small calls in one launch: workgroup b copies segment pair b (table in device-visible pinned memory)
__global__ __launch_bounds__(256) void k_copy_multi(const CopySeg *__restrict__ segs) {
    const CopySeg s = segs[blockIdx.x];
    for (size_t i = threadIdx.x; i < s.n1 + s.n2; i += 256) {
        if (i < s.n1) s.dst1[i] = s.src1[i];
        else s.dst2[i - s.n1] = s.src2[i - s.n1];
    }
}
hipError_t launch_copy_multi(hipStream_t stream, const CopySeg *segs, int n) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_multi, dim3(n), dim3(256), 0, stream, segs);
    return hipGetLastError();
}

hipError_t launch_copy_words(hipStream_t stream, const void *src, void *dst, size_t n_words) {
    if (n_words == 0) return hipSuccess;
    const int grid = (int)std::min<size_t>((n_words + 255) / 256, 64);
    hipLaunchKernelGGL(k_copy_words, dim3(grid), dim3(256), 0, stream, static_cast<const uint32_t *>(src), static_cast<uint32_t *>(dst), n_words);
    return hipGetLastError();
}

hipError_t launch_signal(hipStream_t stream, unsigned long long *flag, unsigned long long value) {
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, stream, flag, value);
    return hipGetLastError();
}

bool map_is_one_workgroup(int M, int N) { return M <= 256 && N == 1; }
hipError_t launch_map_forward(hipStream_t stream, const double *xyz, long long xyz_stride, double *rc, int M, int N,
                              const MapIntr &in, const double *rt, unsigned long long *flag, unsigned long long flag_value) {
    if (flag != nullptr && !map_is_one_workgroup(M, N)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_map_forward, dim3((M + 255) / 256, N < 65535 ? N : 65535), dim3(256), 0, stream, xyz, xyz_stride, rc, M, N, in, rt,
                       flag, flag_value);
    return hipGetLastError();
}
hipError_t launch_map_inverse(hipStream_t stream, const double *rc, double *xyz, int M, int N, const MapIntr &in,
                              const double *rt, unsigned long long *flag, unsigned long long flag_value) {
    if (flag != nullptr && !map_is_one_workgroup(M, N)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_map_inverse, dim3((M + 255) / 256, N < 65535 ? N : 65535), dim3(256), 0, stream, rc, xyz, M, N, in, rt, flag, flag_value);
    return hipGetLastError();
}
hipError_t launch_errors(hipStream_t stream, const double *obj, const double *img, const MapIntr &in,
                         const double *rt, const double *inv_pts, double *partials, int N, int n1, int n2, int S) {
    const size_t lds = sizeof(double) * 3 * (size_t)n1 * n2;
    hipLaunchKernelGGL(k_errors, dim3(N), dim3(256), lds, stream, obj, img, in, rt, inv_pts, partials, n1, n2, S);
    hipLaunchKernelGGL(k_errors_reduce, dim3(1), dim3(256), 0, stream, partials, partials + (size_t)4 * N, N);
    return hipGetLastError();
}

}  // namespace ccal
