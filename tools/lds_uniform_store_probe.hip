// Do LDS stores of a wave-uniform value by ALL 64 lanes to ONE address cost more than the same store by one lane?
// (lm_advance_wave writes the state machine's scalars that way.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/p tools/lds_uniform_store_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int MODE>
__global__ void k(unsigned long long *cyc, double *out, double v0) {
    __shared__ double s[256];
    const int lane = threadIdx.x;
    double v = v0 + lane * 0.0;
    s[lane] = 0.0; s[64 + lane] = 0.0;
    __syncthreads();
    const unsigned long long t0 = now();
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if (MODE == 0) s[i] = v + i;                          // all lanes, one address per store
        else if (MODE == 1) { if (lane == 0) s[i] = v + i; }  // one lane
        else if (MODE == 2) s[64 + ((i * 64 + lane) & 127)] = v + i;   // all lanes, consecutive addresses (conflict-free)
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = now();
    if (lane == 0) cyc[MODE] = t1 - t0;
    out[lane] = s[lane] + s[64 + lane];
}
int main() {
    unsigned long long *c; double *o;
    hipMalloc(&c, 64); hipMalloc(&o, 8 * 256);
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, c, o, 1.0);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, c, o, 1.0);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, c, o, 1.0);
        unsigned long long h[3];
        hipMemcpy(h, c, 24, hipMemcpyDeviceToHost);
        printf("32 x ds_write_b64: all lanes one address %llu cycles | lane 0 only %llu | all lanes, consecutive addresses %llu\n", h[0], h[1], h[2]);
    }
    return 0;
}
