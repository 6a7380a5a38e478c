// mfma4x4_probe.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: which lanes' A and B values meet in which output lane, and what
// one instruction costs next to v_mfma_f64_16x16x4_f64.  (The guide documents the 16x16x4 f64 maps only.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4x4_probe tools/mfma4x4_probe.hip && /tmp/mfma4x4_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// wave w = (la, lb): A = 1 in lane la, B = 1 in lane lb, zero elsewhere; every lane writes its D
__global__ void k_map(double *out) {
    const int lane = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + lane] = d;
}

template <int KIND>
__global__ void k_time(double *out, unsigned long long *cyc, int n) {
    const int lane = threadIdx.x & 63;
    double a = 1.0 + lane * 1e-3, b = 2.0 - lane * 1e-3;
    double c0 = 0.0, c1 = 0.0;
    d4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, c1, 0, 0, 0);
        } else {
            e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, e1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + e0[0] + e1[1] + e0[2] + e1[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double *d;
    unsigned long long *c;
    hipMalloc(&d, sizeof(double) * 4096 * 64);
    hipMalloc(&c, 8);
    hipLaunchKernelGGL(k_map, dim3(4096), dim3(64), 0, 0, d);
    std::vector<double> h(4096 * 64);
    hipMemcpy(h.data(), d, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
    // for every output lane: the (la, lb) pairs that reach it
    printf("output lane <- list of (A lane, B lane) pairs whose product it receives\n");
    for (int lo = 0; lo < 64; lo++) {
        printf("D lane %2d:", lo);
        for (int la = 0; la < 64; la++)
            for (int lb = 0; lb < 64; lb++)
                if (h[(size_t)(la * 64 + lb) * 64 + lo] != 0.0) printf(" (%d,%d)", la, lb);
        printf("\n");
    }
    for (int kind = 0; kind < 2; kind++) {
        const int n = 2000;
        for (int rep = 0; rep < 2; rep++) {
            if (kind == 0) hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(64), 0, 0, d, c, n);
            else hipLaunchKernelGGL(k_time<1>, dim3(1), dim3(64), 0, 0, d, c, n);
        }
        unsigned long long cy;
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per instruction (one wave, two independent accumulators, back to back)\n",
               kind == 0 ? "v_mfma_f64_4x4x4_4b_f64" : "v_mfma_f64_16x16x4_f64", (double)cy / (2.0 * n));
    }
    return 0;
}
