// Probe (diagnostic, not shipped): can a fine-grained device allocation be shared between two
// processes through hipIpc*MemHandle, and does a kernel in one process see LL-style
// (value|flag) 8-byte stores made by a kernel of the other while it spins?
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ipc_probe tools/ipc_probe.hip ; run: /tmp/ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_send(unsigned long long *peer, unsigned int epoch, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        const unsigned long long w = ((unsigned long long)epoch << 32) | (unsigned int)(i * 7 + epoch);
        __hip_atomic_store(peer + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_recv(unsigned long long *inbox, unsigned int epoch, int n, unsigned long long *out, long long budget) {
    const int i = threadIdx.x;
    const long long t0 = wall_clock64();
    unsigned long long w = 0;
    bool ok = false;
    if (i < n) {
        while (true) {
            w = __hip_atomic_load(inbox + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)(w >> 32) == epoch) { ok = true; break; }
            if (wall_clock64() - t0 > budget) break;
        }
        out[i] = ok ? w : ~0ull;
    }
    if (i == 0) out[n] = (unsigned long long)(wall_clock64() - t0);
}

int main() {
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    const int n = 136;
    pid_t pid = fork();
    if (pid == 0) {   // child: opens the parent's inbox and sends after a delay
        hipIpcMemHandle_t h;
        if (read(p2c[0], &h, sizeof h) != sizeof h) return 3;
        CHK(hipSetDevice(0));
        void *peer = nullptr;
        CHK(hipIpcOpenMemHandle(&peer, h, hipIpcMemLazyEnablePeerAccess));
        char c = 1;
        if (write(c2p[1], &c, 1) != 1) return 3;
        for (unsigned int epoch = 1; epoch <= 3; epoch++) {
            usleep(20000);   // the parent is already spinning
            k_send<<<1, 256>>>((unsigned long long *)peer, epoch, n);
            CHK(hipDeviceSynchronize());
            if (read(p2c[0], &c, 1) != 1) return 3;
        }
        CHK(hipIpcCloseMemHandle(peer));
        return 0;
    }
    CHK(hipSetDevice(0));
    unsigned long long *inbox = nullptr, *out = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&inbox, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    CHK(hipMemset(inbox, 0, 4096));
    CHK(hipMalloc((void **)&out, (n + 1) * 8));
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, inbox);
    printf("hipIpcGetMemHandle: %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    if (write(p2c[1], &h, sizeof h) != sizeof h) return 3;
    char c;
    if (read(c2p[0], &c, 1) != 1) { printf("child failed to open the handle\n"); return 2; }
    int bad = 0;
    for (unsigned int epoch = 1; epoch <= 3; epoch++) {
        k_recv<<<1, 256>>>(inbox, epoch, n, out, 200000000LL /* 2 s at 100 MHz */);
        CHK(hipDeviceSynchronize());
        unsigned long long host[137];
        CHK(hipMemcpy(host, out, sizeof host, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++) if (host[i] != (((unsigned long long)epoch << 32) | (unsigned int)(i * 7 + epoch))) bad++;
        printf("epoch %u: waited %.3f ms, mismatches so far %d\n", epoch, host[n] / 1.0e5, bad);
        c = 1;
        if (write(p2c[1], &c, 1) != 1) return 3;
    }
    int st = 0;
    waitpid(pid, &st, 0);
    printf("child exit %d; %s\n", WEXITSTATUS(st), bad == 0 && st == 0 ? "PROBE OK" : "PROBE FAILED");
    return bad == 0 && st == 0 ? 0 : 1;
}
