// Calibration of FETCH_SIZE / WRITE_SIZE for THIS code's access width (8 B per lane, coalesced):
// one pass reads `n` doubles (far larger than L2 + Infinity Cache) and writes n/64 doubles.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read8(const double *__restrict__ in, double *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 123.456) out[0] = s;   // never true: keeps the loads alive
}
__global__ void k_write8(double *__restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (double)i;
}
int main() {
    const size_t n = (size_t)1 << 28;   // 2 GiB of doubles
    double *a, *b;
    hipMalloc(&a, n * 8);
    hipMalloc(&b, 4096);
    hipMemset(a, 0, n * 8);
    hipDeviceSynchronize();
    k_read8<<<2048, 256>>>(a, b, n);
    k_write8<<<2048, 256>>>(a, n);
    hipDeviceSynchronize();
    printf("calib bytes per kernel: %zu\n", n * 8);
    return 0;
}
