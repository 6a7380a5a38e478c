// Dependent-issue latency of the instructions the LM advance chains (one wave, s_memtime around N dependent ops).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/lat_probe tools/lat_probe.hip && /tmp/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 256
// the stamp takes the chained value as an in/out operand so that the chain cannot move across it
__device__ __forceinline__ unsigned long long now(double &x) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(x)::"memory");
    return t;
}

__global__ void k(double *out, unsigned long long *cyc, double x0, double y0) {
    __shared__ double lds[64];
    const int lane = threadIdx.x;
    double x = x0 + lane * 1e-3, y = y0;
    lds[lane] = x;
    unsigned long long t0, t1;
    int c = 0;
#define RUN(body)                                   \
    t0 = now(x);                                    \
    _Pragma("unroll") for (int i = 0; i < N; i++) { body; } \
    t1 = now(x);                                    \
    if (lane == 0) cyc[c] = t1 - t0;                \
    c++;
    RUN(x = fma(x, y, 1e-9))                                                         // 0 fma chain
    RUN(x = x * y)                                                                   // 1 mul chain
    RUN(x = x + y)                                                                   // 2 add chain
    RUN(x = __builtin_amdgcn_update_dpp(0.0, x, 0x150, 0xF, 0xF, true))              // 3 mov_b64_dpp chain
    RUN(x = __builtin_amdgcn_update_dpp(0.0, x, 0x150, 0xF, 0xF, true); x = x * y)   // 4 dpp + mul
    RUN(x = __builtin_amdgcn_rcp(x))                                                 // 5 rcp chain
    RUN(x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 3), __builtin_amdgcn_readlane(__double2loint(x), 3)) * y)  // 6 readlane + mul
    RUN(x = ldexp(x, 1))                                                             // 7 ldexp
    RUN(x = (lane == (i & 63)) ? y : x; x = x * y)                                   // 8 cndmask + mul
    RUN(lds[lane] = x; x = lds[(lane + 1) & 63])                                     // 9 LDS write + read round trip
    RUN(x = __builtin_amdgcn_rsq(x))                                                 // 10 rsq
    {   // 11: 8 independent fma chains (issue rate)
        double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
        t0 = now(a0);
#pragma unroll
        for (int i = 0; i < N / 8; i++) {
            a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9);
            a4 = fma(a4, y, 1e-9); a5 = fma(a5, y, 1e-9); a6 = fma(a6, y, 1e-9); a7 = fma(a7, y, 1e-9);
        }
        x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        t1 = now(x);
        if (lane == 0) cyc[c] = t1 - t0;
        c++;
    }
    {   // 12: float fma chain for comparison
        float f = (float)x;
        double fd = f;
        t0 = now(fd);
        f = (float)fd;
#pragma unroll
        for (int i = 0; i < N; i++) f = fmaf(f, 1.0001f, 1e-9f);
        fd = f;
        t1 = now(fd);
        f = (float)fd;
        if (lane == 0) cyc[c] = t1 - t0;
        c++;
        x += f;
    }
    out[lane] = x;
}

int main() {
    double *out;
    unsigned long long *cyc, h[16];
    (void)hipMalloc(&out, 64 * 8);
    (void)hipMalloc(&cyc, 16 * 8);
    const char *names[] = {"v_fma_f64 chain", "v_mul_f64 chain", "v_add_f64 chain", "v_mov_b64_dpp chain", "dpp + v_mul_f64", "v_rcp_f64 chain",
                           "2 x v_readlane + v_mul_f64", "v_ldexp_f64 chain", "v_cndmask x2 + v_mul_f64", "ds_write_b64 + ds_read_b64 round trip",
                           "v_rsq_f64 chain", "8 independent v_fma_f64 (per instruction)", "v_fma_f32 chain"};
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, 1.000001, 0.999999);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 13; i++) printf("%-45s %6.1f cycles per step\n", names[i], (double)h[i] / N);
    return 0;
}
