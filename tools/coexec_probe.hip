// Do v_mfma_f64_16x16x4_f64 and v_fma_f64 of two co-resident waves of one SIMD overlap, or do they share the FP64 datapath?
// One workgroup of 512 threads = 2 waves per SIMD: waves 0-3 run the "A" loop, waves 4-7 the "B" loop; three runs:
// MFMA alone (B idle), FMA alone (A idle), both.  If the pipes were separate, "both" would take max(A, B); if they share, A + B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(double *out, unsigned long long *cyc, int do_mfma, int do_fma, int n) {
    const int wave = threadIdx.x >> 6;
    double x = 1.0 + threadIdx.x * 1e-6, y = 0.999999;
    d4 acc = {0, 0, 0, 0};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (do_mfma)
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
            }
    } else {
        if (do_fma) {
            double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int u = 0; u < 16; u++) {       // 128 independent-ish FMAs per iteration = 8 MFMA-equivalents of pipe time?
                    a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9);
                    a4 = fma(a4, y, 1e-9); a5 = fma(a5, y, 1e-9); a6 = fma(a6, y, 1e-9); a7 = fma(a7, y, 1e-9);
                }
            }
            x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    unsigned long long t2 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
    if (threadIdx.x == 0) cyc[8] = t2 - t0;
    out[threadIdx.x] = x + acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    double *out;
    unsigned long long *cyc, h[9];
    (void)hipMalloc(&out, 512 * 8);
    (void)hipMalloc(&cyc, 9 * 8);
    const int n = 2000;
    const char *names[] = {"MFMA f64 alone (waves 0-3)", "FMA f64 alone (waves 4-7)", "both"};
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; c++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, cfg[c][0], cfg[c][1], n);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        printf("%-28s wave0 %8.1f cycles/iter   wave4 %8.1f cycles/iter   workgroup %8.1f\n", names[c], (double)h[0] / n, (double)h[4] / n,
               (double)h[8] / n);
    }
    printf("(per iteration: 8 MFMA 16x16x4 f64 = 8 x 2048 flop per wave; 128 v_fma_f64 = 128 x 128 flop per wave)\n");
    return 0;
}
