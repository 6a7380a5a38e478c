// persistent_probe.hip -- what does the hand-off between two LM trials cost as a kernel boundary (the shipped design:
// one k_lm_trial launch per trial) and as a grid-wide hand-off inside ONE persistent launch?
//
// Both variants do exactly the hand-off work of a trial and nothing else (no sweep, no LM advance -- those are the
// same instructions either way): every one of G workgroups (256 threads, one per CU) adds its 68 partial sums as two
// int64 limbs each into replica (workgroup % 8) of a 3-buffer rotation of exact accumulators (ccal_acc.hpp layout:
// [R = 8][5 limbs][68]), then EVERY workgroup needs the grand totals: it reads all 8 replicas (21.8 KB) and the
// 900-byte state, sums them through LDS and folds the result into a value that feeds its next adds (so nothing can be
// hoisted).  Workgroup 0 clears the buffer of the trial after next.
//   A  launch per trial: the totals are read at the top of the next launch (stream order = the barrier)
//   B  persistent, flat barrier: s_waitcnt vmcnt(0) -> one agent-scope arrival counter -> sc1 poll -> sc1 loads
//   C  persistent, XCD-hierarchical barrier: per-XCD counter (HW_REG_XCC_ID), last arriver of an XCD adds to the top
//      counter, pollers watch a generation word
// Every spin is bounded (a stuck barrier ends the kernel with an error flag instead of hanging the queue).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/persistent_probe tools/persistent_probe.hip && /tmp/persistent_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NRED = 68, LIMBS = 5, R = 8, ROW = LIMBS * NRED, BUF = R * ROW;   // int64 words
constexpr int STATE_WORDS = 112;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ long long ld_sc1(const long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the work of one hand-off on the consumer side: totals of buffer `src` -> one double per thread (a fold of everything)
__device__ __forceinline__ double read_totals(const long long *acc, int src, const double *state, double *s_tot, int tid, bool sc1) {
    const long long *b = acc + (size_t)src * BUF;
    long long t = 0;
    if (tid < ROW / 2 * 2) {   // thread t < 340 would own one column; with 256 threads: two strided passes
    }
    double f = 0.0;
    for (int c = tid; c < ROW; c += 256) {
        long long s = 0;
#pragma unroll
        for (int r = 0; r < R; r++) s += sc1 ? ld_sc1(b + r * ROW + c) : b[r * ROW + c];
        t += s;
        s_tot[c] = (double)s;
    }
    const double st = tid < STATE_WORDS ? (sc1 ? __longlong_as_double(ld_sc1(reinterpret_cast<const long long *>(state) + tid)) : state[tid]) : 0.0;
    __syncthreads();
    if (tid < NRED) {
#pragma unroll
        for (int j = 0; j < LIMBS; j++) f += s_tot[j * NRED + tid];
    }
    __syncthreads();
    return f * 1e-300 + st * 1e-300 + (double)(t & 1);
}

__device__ __forceinline__ void add_partials(long long *acc, int dst, int tid, double seed) {
    if (tid < NRED) {
        long long *row = acc + (size_t)dst * BUF + (blockIdx.x & (R - 1)) * ROW;
        const long long v = (long long)(tid + 1) + (long long)seed;
        __hip_atomic_fetch_add(row + 1 * NRED + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(row + 2 * NRED + tid, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (inside one launch the zeros must be written through: a dirty line in workgroup 0's L2 would be written back over
//  the sums the other XCDs' memory-side atomics have formed meanwhile; across a kernel boundary plain stores do)
__device__ __forceinline__ void clear_buf(long long *acc, int which, int tid, bool through = false) {
    if (through) {
        long long *d = acc + (size_t)which * BUF;
        for (int i = tid; i < BUF; i += 256) __hip_atomic_store(d + i, 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    longlong2 *d = reinterpret_cast<longlong2 *>(acc + (size_t)which * BUF);
    for (int i = tid; i < BUF / 2; i += 256) d[i] = make_longlong2(0, 0);
}

// ---- A: one launch per trial --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_trial(long long *acc, double *state, double *sink, int k) {
    __shared__ double s_tot[ROW];
    const int tid = threadIdx.x;
    const double f = read_totals(acc, (k + 2) % 3, state, s_tot, tid, false);
    if (blockIdx.x == 0) {
        clear_buf(acc, (k + 1) % 3, tid);
        if (tid < STATE_WORDS) state[tid] = f + k;
    }
    add_partials(acc, k % 3, tid, f);
    if (tid == 0) sink[blockIdx.x] = f;
}

// ---- B / C: persistent ---------------------------------------------------------------------------------------------
struct Sync { unsigned int top; unsigned int pad0[31]; unsigned int gen; unsigned int pad1[31]; unsigned int xcd[8 * 32]; unsigned int err; };

template <bool HIER>
__global__ __launch_bounds__(256) void k_persistent(long long *acc, double *state, double *sink, Sync *sy, int trials,
                                                     unsigned long long *t_per_trial) {
    __shared__ double s_tot[ROW];
    __shared__ int s_ok;
    const int tid = threadIdx.x, G = gridDim.x;
    unsigned int xcc = 0, n_xcd = 0;
    if (HIER) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        // census of the XCDs' workgroups (once): every workgroup counts itself, then everybody waits for G
        if (tid == 0) {
            __hip_atomic_fetch_add(&sy->xcd[(8 + xcc) * 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&sy->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long t0 = wall_clock64();
            while (__hip_atomic_load(&sy->top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                if (wall_clock64() - t0 > 200000000ll) { sy->err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            n_xcd = __hip_atomic_load(&sy->xcd[(8 + xcc) * 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    double f = 0.0;
    const unsigned long long t_begin = wall_clock64();
    for (int k = 0; k < trials; k++) {
        if (blockIdx.x == 0) clear_buf(acc, (k + 1) % 3, tid, true);
        add_partials(acc, k % 3, tid, f);
        if (blockIdx.x == 0 && tid < STATE_WORDS)
            __hip_atomic_store(reinterpret_cast<long long *>(state) + tid, __double_as_longlong(f + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // every adding wave drains its atomics / stores, then one lane arrives for the workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            bool ok = true;
            const long long t0 = wall_clock64();
            if (!HIER) {
                const unsigned int want = (unsigned)(k + 1) * (unsigned)G;
                __hip_atomic_fetch_add(&sy->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&sy->top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (wall_clock64() - t0 > 100000000ll) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            } else {
                // per-XCD arrival; the last arriver of an XCD reports to the top; the last XCD publishes the generation
                const unsigned int a = __hip_atomic_fetch_add(&sy->xcd[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                if (a == (unsigned)(k + 1) * n_xcd) {
                    const unsigned int t = __hip_atomic_fetch_add(&sy->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
                    // (top starts at G after the census; 8 XCD reports per trial -- fewer if an XCD got no workgroup)
                    unsigned int live = 0;
                    for (int x = 0; x < 8; x++) live += __hip_atomic_load(&sy->xcd[8 + x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                    if (t == (unsigned)G + (unsigned)(k + 1) * live)
                        __hip_atomic_store(&sy->gen, (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                while (__hip_atomic_load(&sy->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(k + 1)) {
                    if (wall_clock64() - t0 > 100000000ll) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            s_ok = ok ? 1 : 0;
            if (!ok) sy->err = 2;
        }
        __syncthreads();
        if (!s_ok) break;                               // (workgroup-uniform: a stuck barrier ends the kernel)
        f = read_totals(acc, k % 3, state, s_tot, tid, true);
    }
    if (tid == 0) {
        sink[blockIdx.x] = f;
        if (blockIdx.x == 0) *t_per_trial = (wall_clock64() - t_begin);
    }
}

int main() {
    const int G = 250, TRIALS = 2000;
    long long *acc; double *state, *sink; Sync *sy; unsigned long long *tp;
    CHK(hipMalloc(&acc, sizeof(long long) * 3 * BUF));
    CHK(hipMalloc(&state, 8 * STATE_WORDS));
    CHK(hipMalloc(&sink, 8 * 1024));
    CHK(hipMalloc(&sy, sizeof(Sync)));
    CHK(hipMalloc(&tp, 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipStream_t st;
    CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
        // A
        CHK(hipMemsetAsync(acc, 0, sizeof(long long) * 3 * BUF, st));
        CHK(hipMemsetAsync(state, 0, 8 * STATE_WORDS, st));
        for (int k = 0; k < 50; k++) hipLaunchKernelGGL(k_trial, dim3(G), dim3(256), 0, st, acc, state, sink, k);
        CHK(hipEventRecord(e0, st));
        for (int k = 0; k < TRIALS; k++) hipLaunchKernelGGL(k_trial, dim3(G), dim3(256), 0, st, acc, state, sink, k);
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("A  one launch per trial (250 workgroups, hand-off work only): %.2f us per trial\n", 1e3 * ms / TRIALS);
        // B, C
        for (int hier = 0; hier < 2; hier++) {
            CHK(hipMemsetAsync(acc, 0, sizeof(long long) * 3 * BUF, st));
            CHK(hipMemsetAsync(sy, 0, sizeof(Sync), st));
            CHK(hipEventRecord(e0, st));
            if (hier) hipLaunchKernelGGL(k_persistent<true>, dim3(G), dim3(256), 0, st, acc, state, sink, sy, TRIALS, tp);
            else hipLaunchKernelGGL(k_persistent<false>, dim3(G), dim3(256), 0, st, acc, state, sink, sy, TRIALS, tp);
            CHK(hipEventRecord(e1, st));
            CHK(hipEventSynchronize(e1));
            CHK(hipEventElapsedTime(&ms, e0, e1));
            Sync h;
            unsigned long long ticks;
            CHK(hipMemcpy(&h, sy, sizeof h, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(&ticks, tp, 8, hipMemcpyDeviceToHost));
            printf("%s persistent, %s barrier: %.2f us per trial (in-kernel clock: %.2f us), barrier error flag %u\n", hier ? "C " : "B ",
                   hier ? "XCD-hierarchical" : "flat counter", 1e3 * ms / TRIALS, ticks / 100.0 / TRIALS, h.err);
        }
    }
    return 0;
}
