// Feasibility probe of the sweep in lane-group form (tools/lg_probe_math.hpp): rows + Gram sums of N images x P corners
// with L lanes per image and the sums in registers (pass A: pose blocks, reduced over the group's lanes and rotated per image; pass B:
// intrinsic block, one total per wave), checked against point_rows_K on the host and timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-sched-strategy=max-ilp -I cameracalibrations_amd/csrc -I tools -o /tmp/lg_probe tools/lg_probe.hip
//   /tmp/lg_probe [N] [P1] [P2]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "lg_probe_math.hpp"
#include "ccal_wave.hpp"

using namespace ccal;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// sum over the L lanes of a group (consecutive lanes, L | 16), every lane gets the total
template <int L>
__device__ __forceinline__ double group_sum(double v) {
    if constexpr (L >= 2) v += dpp_mov<0xB1>(v);
    if constexpr (L >= 4) v += dpp_mov<0x4E>(v);
    if constexpr (L >= 8) v += dpp_mov<0x141>(v);
    if constexpr (L >= 16) v += dpp_mov<0x140>(v);
    return v;
}

// one walk of a lane over every L-th corner of its image: the next corner's observation is requested before the current one's
// arithmetic (one wave per SIMD or two: nobody else hides the load), the template comes from LDS
template <int L, class F>
__device__ __forceinline__ void walk(const float2 *__restrict__ ip, const double *__restrict__ s_obj, int P, int sub, F &&f) {
    int p = sub;
    float2 uv_n = ip[p < P ? p : 0];
    for (; p < P; p += L) {
        const float2 uv = uv_n;
        const int pn = p + L;
        uv_n = ip[pn < P ? pn : p];
        const double M[3] = {s_obj[3 * p], s_obj[3 * p + 1], s_obj[3 * p + 2]};
        f(M, (double)uv.x, (double)uv.y);
    }
}

#ifndef LG_MINW
#define LG_MINW 2
#endif
#ifndef LG_SPLIT_A
#define LG_SPLIT_A 1
#endif

// MODE 0: timing (pass A per batch of 64 / L images with the group reduce and the rotation, pass B over all batches, one row of
// totals per wave);  MODE 1: check (every image's 16 x 16 Gram matrix to memory, free-aspect form folded like the product would)
template <int L, int MODE>
__global__ __launch_bounds__(256, LG_MINW) void k_lg(const float2 *__restrict__ img, const float *__restrict__ obj,
                                                      const double *__restrict__ ext, const double *__restrict__ intr, double aspect,
                                                      int N, int P, double *__restrict__ gram_out, double *__restrict__ wave_out) {
    constexpr int G = 64 / L;
    // per wave: K and the reduced B~ of the batch's images are parked in LDS while the next walk over the corners runs (kept in
    // registers they would stay live across it: 126 vector registers on top of the walk's own sums)
    __shared__ double s_park[4 * G * (LG_NB + 9)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *park = s_park + (wave * G + (lane / L)) * (LG_NB + 9);
    __shared__ double s_obj[3 * 256];                // the template as doubles (P <= 256 in this probe)
    for (int k = threadIdx.x; k < 3 * P; k += 256) s_obj[k] = (double)obj[k];
    __syncthreads();
    const int sub = lane % L, grp = lane / L;
    double in[NINTR];
#pragma unroll
    for (int a = 0; a < NINTR; a++) in[a] = to_sgpr(intr[a]);
    const int n_waves = gridDim.x * 4, wid = blockIdx.x * 4 + wave;
    const int n_batches = (N + G - 1) / G;
    double chk = 0.0;
    // ---- pass A ----
    for (int b = wid; b < n_batches; b += n_waves) {
        const int i = b * G + grp;
        const bool act = i < N;
        const int ii = act ? i : N - 1;
        double e6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) e6[k] = ext[(size_t)ii * 8 + k];
        double R[9];
        {
            double K[9];
            rodrigues_RK(e6, R, K);
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) park[LG_NB + k] = K[k];
            }
        }
        const double tv[3] = {e6[3], e6[4], e6[5]};
        double At[LG_NA], Bt[LG_NB], ht[LG_NH];
#pragma unroll
        for (int k = 0; k < LG_NA; k++) At[k] = 0.0;
#pragma unroll
        for (int k = 0; k < LG_NB; k++) Bt[k] = 0.0;
#pragma unroll
        for (int k = 0; k < LG_NH; k++) ht[k] = 0.0;
        const float2 *ip = img + (size_t)ii * P;
#if LG_SPLIT_A
        // 54 + 27 sums in two walks over the corners (the rows are formed twice)
        walk<L>(ip, s_obj, P, sub, [&](const double (&M)[3], double uo, double vo) {
            LgPoint q;
            lg_project(M, uo, vo, R, tv, in, q);
            double cu[6], cv[6];
            lg_pose_columns(q, in, cu, cv);
            lg_accum_A1(q, cu, cv, Bt);
        });
#pragma unroll
        for (int k = 0; k < LG_NB; k++) {
            const double v = group_sum<L>(Bt[k]);
            if (sub == 0) park[k] = v;
        }
        walk<L>(ip, s_obj, P, sub, [&](const double (&M)[3], double uo, double vo) {
            LgPoint q;
            lg_project(M, uo, vo, R, tv, in, q);
            double cu[6], cv[6];
            lg_pose_columns(q, in, cu, cv);
            lg_accum_A2(q, cu, cv, At, ht);
        });
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < LG_NB; k++) Bt[k] = park[k];
#else
        walk<L>(ip, s_obj, P, sub, [&](const double (&M)[3], double uo, double vo) {
            LgPoint q;
            lg_project(M, uo, vo, R, tv, in, q);
            double cu[6], cv[6];
            lg_pose_columns(q, in, cu, cv);
            lg_accum_A(q, cu, cv, At, Bt, ht);
        });
#pragma unroll
        for (int k = 0; k < LG_NB; k++) Bt[k] = group_sum<L>(Bt[k]);
#endif
#pragma unroll
        for (int k = 0; k < LG_NA; k++) At[k] = group_sum<L>(At[k]);
#pragma unroll
        for (int k = 0; k < LG_NH; k++) ht[k] = group_sum<L>(ht[k]);
        double K[9];
#if !LG_SPLIT_A
        wave_lds_sync();
#endif
#pragma unroll
        for (int k = 0; k < 9; k++) K[k] = park[LG_NB + k];
        wave_lds_sync();
        lg_apply_T(K, At, Bt, ht);
        if (aspect > 0.0) lg_fold_aspect_B(aspect, Bt);
        if (MODE == 1) {
            if (act && sub == 0) {
                double *Gm = gram_out + (size_t)i * 256;
                for (int a = 0; a < 6; a++) {
                    for (int c = 0; c <= a; c++) { Gm[a * 16 + c] = At[tri(a, c)]; Gm[c * 16 + a] = At[tri(a, c)]; }
                    for (int s = 0; s < 9; s++) { Gm[a * 16 + 6 + s] = Bt[a * 9 + s]; Gm[(6 + s) * 16 + a] = Bt[a * 9 + s]; }
                    Gm[a * 16 + 15] = ht[a]; Gm[15 * 16 + a] = ht[a];
                }
            }
        } else {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < LG_NA; k++) s += At[k];
#pragma unroll
            for (int k = 0; k < LG_NB; k++) s += Bt[k];
#pragma unroll
            for (int k = 0; k < LG_NH; k++) s += ht[k];
            chk += (act && sub == 0) ? s : 0.0;
        }
    }
    // ---- pass B ----
    double Ct[LG_NC], gt[LG_NG], ee = 0.0;
#pragma unroll
    for (int k = 0; k < LG_NC; k++) Ct[k] = 0.0;
#pragma unroll
    for (int k = 0; k < LG_NG; k++) gt[k] = 0.0;
    for (int b = wid; b < n_batches; b += n_waves) {
        const int i = b * G + grp;
        const bool act = i < N;
        const int ii = act ? i : N - 1;
        double e6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) e6[k] = ext[(size_t)ii * 8 + k];
        double R[9], K[9];
        rodrigues_RK(e6, R, K);
        const double tv[3] = {e6[3], e6[4], e6[5]};
        const float2 *ip = img + (size_t)ii * P;
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < LG_NC; k++) Ct[k] = 0.0;
#pragma unroll
            for (int k = 0; k < LG_NG; k++) gt[k] = 0.0;
            ee = 0.0;
        }
        if (act) {
            walk<L>(ip, s_obj, P, sub, [&](const double (&M)[3], double uo, double vo) {
                LgPoint q;
                lg_project(M, uo, vo, R, tv, in, q);
                lg_accum_B(q, Ct, gt, ee);
            });
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < LG_NC; k++) Ct[k] = group_sum<L>(Ct[k]);
#pragma unroll
            for (int k = 0; k < LG_NG; k++) gt[k] = group_sum<L>(gt[k]);
            ee = group_sum<L>(ee);
            if (aspect > 0.0) lg_fold_aspect_C(aspect, Ct, gt);
            if (act && sub == 0) {
                double *Gm = gram_out + (size_t)i * 256;
                for (int a = 0; a < 9; a++) {
                    for (int c = 0; c <= a; c++) { Gm[(6 + a) * 16 + 6 + c] = Ct[tri(a, c)]; Gm[(6 + c) * 16 + 6 + a] = Ct[tri(a, c)]; }
                    Gm[(6 + a) * 16 + 15] = gt[a]; Gm[15 * 16 + 6 + a] = gt[a];
                }
                Gm[15 * 16 + 15] = ee;
            }
        }
    }
    if (MODE == 0) {
        if (aspect > 0.0) lg_fold_aspect_C(aspect, Ct, gt);
        double s = ee;
#pragma unroll
        for (int k = 0; k < LG_NC; k++) s += Ct[k];
#pragma unroll
        for (int k = 0; k < LG_NG; k++) s += gt[k];
        s = wave_sum(s + chk);
        if (lane == 0) wave_out[wid] = s;
    }
}

// What the vector unit sustains chip-wide on FP64: NCH independent v_fma_f64 chains per lane, `iters` rounds, every wave; shader clock
// from s_memtime against the 100 MHz s_memrealtime over the same interval.
template <int NCH>
__global__ __launch_bounds__(1024) void k_fma_rate(double *out, unsigned long long *clk, int iters, double y) {
    double a[NCH];
#pragma unroll
    for (int k = 0; k < NCH; k++) a[k] = 1.0 + 1e-3 * (threadIdx.x + k);
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    // (128 instructions between two branches: a taken branch costs a wave ~30 cycles, which a loop of 16 would report as issue time)
    constexpr int UNR = 128 / NCH;
    for (int i = 0; i < iters; i += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; u++)
#pragma unroll
            for (int k = 0; k < NCH; k++) a[k] = fma(a[k], y, 1e-9);
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NCH; k++) s += a[k];
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) : "v"(s) : "memory");
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x < 1024) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NCH>
static void fma_rate_t(int grid, int block, int iters);
static void fma_rate(int grid, int iters) { fma_rate_t<16>(grid, 256, iters); }
template <int NCH>
static void fma_rate_t(int grid, int block, int iters) {
    double *d_out; unsigned long long *d_clk;
    CK(hipMalloc(&d_out, 8)); CK(hipMalloc(&d_clk, 16 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_fma_rate<NCH>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, iters, 0.999999);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_fma_rate<NCH>), dim3(grid), dim3(block), 0, 0, d_out, d_clk, iters, 0.999999);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2048];
    CK(hipMemcpy(h, d_clk, sizeof h, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    const int nb = grid < 1024 ? grid : 1024;
    for (int b = 0; b < nb; b++) { cyc += (double)h[2 * b]; real += (double)h[2 * b + 1]; }
    const double flop = 2.0 * NCH * (double)iters * (double)block * grid;
    printf("  v_fma_f64 x %2d independent chains, grid %4d x %3d threads, %d rounds: %.1f us, %.1f TFLOP/s; shader clock %.2f GHz (s_memtime / "
           "s_memrealtime); a wave issues one every %.2f cycles\n", NCH, grid, block, iters, 1e3 * ms, flop / (1e-3 * ms) * 1e-12, cyc / real * 0.1,
           (cyc / nb) / ((double)NCH * iters));
    CK(hipFree(d_out)); CK(hipFree(d_clk));
}

struct Problem {
    int N, P;
    std::vector<float> img, obj;
    std::vector<double> ext, intr;
};

static Problem make(int N, int n1, int n2, unsigned seed) {
    Problem pr;
    pr.N = N; pr.P = n1 * n2;
    pr.intr = {1610.0, 1600.0, 756.8, 995.4, -0.12, 0.03, 1e-3, -5e-4, -4e-3};
    for (int j = 0; j < n2; j++)
        for (int i = 0; i < n1; i++) { pr.obj.push_back((float)i); pr.obj.push_back((float)j); pr.obj.push_back(0.f); }
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> ur(-0.5, 0.5), ud(14.0, 28.0);
    std::normal_distribution<double> nz(0.0, 0.1);
    pr.ext.resize((size_t)N * 8, 0.0);
    pr.img.resize((size_t)N * pr.P * 2);
    for (int n = 0; n < N; n++) {
        double e6[6] = {ur(rng), ur(rng), ur(rng), 0, 0, 0};
        double R[9], K[9];
        rodrigues_RK(e6, R, K);
        const double cxb = 0.5 * (n1 - 1), cyb = 0.5 * (n2 - 1), depth = ud(rng);
        // board centre at (ox, oy, depth)
        const double ox = ur(rng) * 2.0, oy = ur(rng) * 2.0;
        e6[3] = ox - (R[0] * cxb + R[1] * cyb); e6[4] = oy - (R[3] * cxb + R[4] * cyb); e6[5] = depth - (R[6] * cxb + R[7] * cyb);
        for (int k = 0; k < 6; k++) pr.ext[(size_t)n * 8 + k] = e6[k] + (k < 3 ? 1e-3 : 1e-2) * ur(rng);    // (the sweep evaluates near, not at, the truth)
        for (int p = 0; p < pr.P; p++) {
            const double M[3] = {pr.obj[3 * p], pr.obj[3 * p + 1], pr.obj[3 * p + 2]};
            double eu, ev;
            point_residual(M, 0.0, 0.0, R, e6 + 3, pr.intr.data(), eu, ev);
            pr.img[((size_t)n * pr.P + p) * 2] = (float)(eu + nz(rng));
            pr.img[((size_t)n * pr.P + p) * 2 + 1] = (float)(ev + nz(rng));
        }
    }
    return pr;
}

template <int L>
static void run(const Problem &pr, double aspect, const float2 *d_img, const float *d_obj, const double *d_ext, const double *d_intr,
                double *d_gram, double *d_wave, int grid, int reps) {
    // check
    CK(hipMemset(d_gram, 0, sizeof(double) * 256 * (size_t)pr.N));
    hipLaunchKernelGGL((k_lg<L, 1>), dim3(grid), dim3(256), 0, 0, d_img, d_obj, d_ext, d_intr, aspect, pr.N, pr.P, d_gram, d_wave);
    CK(hipDeviceSynchronize());
    const int n_chk = std::min(pr.N, 48);
    std::vector<double> Gd((size_t)256 * pr.N);
    CK(hipMemcpy(Gd.data(), d_gram, sizeof(double) * 256 * (size_t)pr.N, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int t = 0; t < n_chk; t++) {
        const int n = (int)((long long)t * (pr.N - 1) / std::max(1, n_chk - 1));
        double R[9], K[9];
        rodrigues_RK(&pr.ext[(size_t)n * 8], R, K);
        std::vector<double> Gh(256, 0.0);
        for (int p = 0; p < pr.P; p++) {
            const double M[3] = {pr.obj[3 * p], pr.obj[3 * p + 1], pr.obj[3 * p + 2]};
            double mu[16], mv[16];
            point_rows_K(M, pr.img[((size_t)n * pr.P + p) * 2], pr.img[((size_t)n * pr.P + p) * 2 + 1], R, K, &pr.ext[(size_t)n * 8 + 3],
                         pr.intr.data(), aspect, mu, mv);
            for (int a = 0; a < 16; a++)
                for (int b = 0; b < 16; b++) Gh[a * 16 + b] += mu[a] * mu[b] + mv[a] * mv[b];
        }
        for (int a = 0; a < 16; a++)
            for (int b = 0; b < 16; b++) {
                const double sc = sqrt(Gh[a * 16 + a] * Gh[b * 16 + b]);
                const double d = fabs(Gd[(size_t)n * 256 + a * 16 + b] - Gh[a * 16 + b]);
                if (sc > 0.0) worst = std::max(worst, d / sc);
                else if (d != 0.0) worst = std::max(worst, 1.0);
            }
    }
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k_lg<L, 0>), dim3(grid), dim3(256), 0, 0, d_img, d_obj, d_ext, d_intr, aspect, pr.N, pr.P, d_gram, d_wave);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL((k_lg<L, 0>), dim3(grid), dim3(256), 0, 0, d_img, d_obj, d_ext, d_intr, aspect, pr.N, pr.P, d_gram, d_wave);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  L = %2d  grid %4d  aspect %.0f: %8.2f us per launch (back to back), Gram vs point_rows_K on the host: %.2e (scaled, %d images)\n", L, grid,
           aspect, 1e3 * ms / reps, worst, n_chk);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000, n1 = argc > 2 ? atoi(argv[2]) : 15, n2 = argc > 3 ? atoi(argv[3]) : 10;
    Problem pr = make(N, n1, n2, 20241224u);
    float2 *d_img; float *d_obj; double *d_ext, *d_intr, *d_gram, *d_wave;
    CK(hipMalloc(&d_img, sizeof(float) * pr.img.size())); CK(hipMalloc(&d_obj, sizeof(float) * pr.obj.size()));
    CK(hipMalloc(&d_ext, sizeof(double) * pr.ext.size())); CK(hipMalloc(&d_intr, sizeof(double) * 9));
    CK(hipMalloc(&d_gram, sizeof(double) * 256 * (size_t)N)); CK(hipMalloc(&d_wave, sizeof(double) * 65536));
    CK(hipMemcpy(d_img, pr.img.data(), sizeof(float) * pr.img.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_obj, pr.obj.data(), sizeof(float) * pr.obj.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ext, pr.ext.data(), sizeof(double) * pr.ext.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_intr, pr.intr.data(), sizeof(double) * 9, hipMemcpyHostToDevice));
    printf("FP64 vector rate of the chip (78.6 TFLOP/s is the 2.4 GHz figure):\n");
    for (int grid : {256, 512, 768, 1024, 2048}) fma_rate(grid, 4096);
    printf("who shares what: one wave alone on the chip | one wave per compute unit | two | four (one per SIMD) | 1 of 4 compute units with four\n");
    fma_rate_t<16>(1, 64, 4096);
    fma_rate_t<16>(256, 64, 4096);
    fma_rate_t<16>(256, 128, 4096);
    fma_rate_t<16>(256, 256, 4096);
    fma_rate_t<16>(64, 256, 4096);
    printf("one dependent chain per wave (what a latency-bound wave sees): alone | one wave per SIMD everywhere\n");
    fma_rate_t<1>(1, 64, 65536);
    fma_rate_t<1>(256, 256, 65536);
    fma_rate_t<2>(256, 256, 32768);
    fma_rate_t<4>(256, 256, 16384);
    printf("lane-group rows + Gram sums, %d images x %d corners (K = 9), min waves per SIMD hint %d, pose blocks in %d walk(s)\n", N, pr.P, LG_MINW, LG_SPLIT_A ? 2 : 1);
    for (int grid : {256, 512, 1024}) {
        run<4>(pr, 0.0, d_img, d_obj, d_ext, d_intr, d_gram, d_wave, grid, 20);
        run<8>(pr, 0.0, d_img, d_obj, d_ext, d_intr, d_gram, d_wave, grid, 20);
        run<16>(pr, 0.0, d_img, d_obj, d_ext, d_intr, d_gram, d_wave, grid, 20);
    }
    run<8>(pr, 1.0, d_img, d_obj, d_ext, d_intr, d_gram, d_wave, 512, 20);
    return 0;
}
