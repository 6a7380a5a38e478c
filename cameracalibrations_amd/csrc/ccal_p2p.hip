// ccal_p2p.hip -- inbox allocation / IPC mapping and the generic small all-reduce kernel of the
// xGMI row exchange (protocol: ccal_p2p.hpp, device side: ccal_p2p_dev.hpp).
#include "ccal_p2p.hpp"

#include <cstdlib>
#include <cstring>

#include "ccal_p2p_dev.hpp"

namespace ccal {

static_assert(sizeof(hipIpcMemHandle_t) == P2P_HANDLE_BYTES, "IPC handle size");

#define P2PCHK(x)                                                                              \
    do {                                                                                       \
        const hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                                \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                              \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)

P2P::~P2P() {
    if (!created_) return;
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < dev_.world; r++)
        if (connected_ && ipc_ && r != dev_.rank && dev_.peer[r]) (void)hipIpcCloseMemHandle(dev_.peer[r]);
    (void)hipFree(dev_.inbox);
    (void)hipFree(dev_.epoch);
    (void)hipHostFree(dev_.err);
}

int P2P::create(int world, int rank, char handle_out[P2P_HANDLE_BYTES], std::string &err) {
    if (created_) { err = "P2P::create called twice"; return -1; }
    if (world < 1 || world > P2P_MAX_RANKS || rank < 0 || rank >= world) { err = "P2P: bad world/rank (at most 16 ranks)"; return -1; }
    P2PCHK(hipGetDevice(&device_));
    memset(&dev_, 0, sizeof dev_);
    dev_.world = world;
    dev_.rank = rank;
    // fine-grained: remote stores land coherently and local polls are not served from a stale L2 line
    P2PCHK(hipExtMallocWithFlags(reinterpret_cast<void **>(&dev_.inbox), P2P_INBOX_WORDS * 8, hipDeviceMallocFinegrained));
    P2PCHK(hipMemset(dev_.inbox, 0, P2P_INBOX_WORDS * 8));
    unsigned int *words = nullptr;
    P2PCHK(hipMalloc(reinterpret_cast<void **>(&words), 64));
    P2PCHK(hipMemset(words, 0, 64));
    dev_.epoch = words;
    // the time-out flag lives in pinned host memory: the host reads it without a copy
    P2PCHK(hipHostMalloc(reinterpret_cast<void **>(&dev_.err), 64, hipHostMallocMapped));
    *dev_.err = 0;
    P2PCHK(hipDeviceSynchronize());
    double ms = 5000.0;
    if (const char *s = getenv("CCAL_P2P_TIMEOUT_MS")) { const double v = atof(s); if (v > 0) ms = v; }
    dev_.budget = (long long)(ms * 1.0e5);   // wall_clock64 counts at 100 MHz
    hipIpcMemHandle_t h;
    P2PCHK(hipIpcGetMemHandle(&h, dev_.inbox));
    memcpy(handle_out, &h, P2P_HANDLE_BYTES);
    dev_.peer[rank] = dev_.inbox;
    created_ = true;
    return 0;
}

int P2P::connect(const char *handles, std::string &err) {
    if (!created_) { err = "P2P::connect before create"; return -1; }
    if (connected_) { err = "P2P::connect called twice"; return -1; }
    P2PCHK(hipSetDevice(device_));
    for (int r = 0; r < dev_.world; r++) {
        if (r == dev_.rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)r * P2P_HANDLE_BYTES, P2P_HANDLE_BYTES);
        void *p = nullptr;
        P2PCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        dev_.peer[r] = static_cast<unsigned long long *>(p);
    }
    connected_ = true;
    ipc_ = true;
    return 0;
}

int P2P::connect_local(P2P *const *peers, std::string &err) {
    if (!created_) { err = "P2P::connect_local before create"; return -1; }
    if (connected_) { err = "P2P::connect_local called twice"; return -1; }
    P2PCHK(hipSetDevice(device_));
    for (int r = 0; r < dev_.world; r++) {
        if (r == dev_.rank) continue;
        if (!peers[r] || !peers[r]->created_) { err = "P2P::connect_local: peer inbox missing"; return -1; }
        if (peers[r]->device_ != device_) {
            int can = 0;
            P2PCHK(hipDeviceCanAccessPeer(&can, device_, peers[r]->device_));
            if (!can) { err = "P2P::connect_local: no peer access between the devices"; return -1; }
            const hipError_t e = hipDeviceEnablePeerAccess(peers[r]->device_, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return -1; }
            (void)hipGetLastError();
        }
        dev_.peer[r] = peers[r]->dev_.inbox;
    }
    connected_ = true;
    ipc_ = false;
    return 0;
}

int P2P::check_and_clear(std::string &err) {
    volatile int *flag = dev_.err;
    if (*flag == 0) return 0;
    *flag = 0;
    err = "xGMI row exchange timed out waiting for a peer rank";
    return 1;
}

__global__ __launch_bounds__(256) void k_p2p_allreduce(P2PDev p, double *buf, int n) {
    __shared__ double s_val[P2P_MAX_DOUBLES];
    __shared__ double s_all[P2P_MAX_RANKS * P2P_MAX_DOUBLES];
    const int tid = threadIdx.x;
    if (tid < n) s_val[tid] = buf[tid];
    __syncthreads();
    p2p_allreduce_wg(p, s_val, s_all, n, tid);
    if (tid < n) buf[tid] = s_val[tid];
}

// the stand-alone all-reduce in two stream-ordered halves (shards sharing one in-order queue, ccal_p2p_dev.hpp):
// the send takes the next sequence number without publishing it, the receive publishes it
__global__ __launch_bounds__(256) void k_p2p_send(P2PDev p, const double *buf, int n) {
    __shared__ double s_val[P2P_MAX_DOUBLES];
    const int tid = threadIdx.x;
    if (tid < n) s_val[tid] = buf[tid];
    __syncthreads();
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    p2p_send_row(p, p2p_tag(seq, 0), p2p_slot(seq, 0), s_val, n, tid);
}

__global__ __launch_bounds__(256) void k_p2p_recv(P2PDev p, double *buf, int n) {
    __shared__ double s_val[P2P_MAX_DOUBLES];
    __shared__ double s_all[P2P_MAX_RANKS * P2P_MAX_DOUBLES];
    const int tid = threadIdx.x;
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    p2p_recv_rows(p, p2p_tag(seq, 0), p2p_slot(seq, 0), s_val, s_all, n, tid);
    if (tid < n) buf[tid] = s_val[tid];
    if (tid == 0) __hip_atomic_store(p.epoch, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// diagnostics: the `world` workgroups of ONE launch act as the ranks of a node -- rank r = workgroup r with its own inbox in this
// device's memory -- and run `rounds` consecutive exchanges of p2p_exchange_rows the way a series of solves numbers them (three
// trials per solve, k = 1..3, then the next sequence number, starting at seq0).  All ranks are co-resident by construction
// (world <= 16 workgroups), so this exercises the protocol itself at any world size on one GPU: tags, the four-slot rotation, the
// rank-ordered sum, the 16-bit wrap of the sequence field in the tag.
__global__ __launch_bounds__(256) void k_p2p_world_debug(const P2PDev *devs, int rounds, unsigned int seq0, const double *rows,
                                                         double *sums) {
    __shared__ double s_val[P2P_MAX_DOUBLES];
    __shared__ double s_all[P2P_MAX_RANKS * P2P_MAX_DOUBLES];
    const int tid = threadIdx.x, rank = blockIdx.x, world = gridDim.x;
    const P2PDev p = devs[rank];
    for (int rd = 0; rd < rounds; rd++) {
        const unsigned int seq = seq0 + (unsigned)(rd / 3), k = 1u + (unsigned)(rd % 3);
        if (tid < P2P_MAX_DOUBLES) s_val[tid] = rows[((size_t)rd * world + rank) * P2P_MAX_DOUBLES + tid];
        __syncthreads();
        const bool ok = p2p_exchange_rows(p, p2p_tag(seq, k), p2p_slot(seq, k), s_val, s_all, P2P_MAX_DOUBLES, tid, true);
        if (tid < P2P_MAX_DOUBLES) sums[((size_t)rd * world + rank) * P2P_MAX_DOUBLES + tid] = s_val[tid];
        __syncthreads();
        if (!ok) return;                      // (workgroup-uniform; the peers time out in turn)
    }
}

hipError_t p2p_world_debug(hipStream_t stream, int world, int rounds, unsigned int seq0, const double *rows_host, double *sums_host,
                           int *timed_out) {
    if (world < 1 || world > P2P_MAX_RANKS || rounds < 1) return hipErrorInvalidValue;
    unsigned long long *boxes = nullptr;
    P2PDev *d_devs = nullptr;
    double *d_rows = nullptr, *d_sums = nullptr;
    int *h_err = nullptr;
    const size_t n = (size_t)rounds * world * P2P_MAX_DOUBLES;
    hipError_t e;
    auto done = [&](hipError_t r) {
        (void)hipFree(boxes); (void)hipFree(d_devs); (void)hipFree(d_rows); (void)hipFree(d_sums); (void)hipHostFree(h_err);
        return r;
    };
    if ((e = hipExtMallocWithFlags(reinterpret_cast<void **>(&boxes), (size_t)world * P2P_INBOX_WORDS * 8, hipDeviceMallocFinegrained)) != hipSuccess) return done(e);
    if ((e = hipMemset(boxes, 0, (size_t)world * P2P_INBOX_WORDS * 8)) != hipSuccess) return done(e);
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&h_err), 64, hipHostMallocMapped)) != hipSuccess) return done(e);
    *h_err = 0;
    P2PDev devs[P2P_MAX_RANKS];
    memset(devs, 0, sizeof devs);
    for (int r = 0; r < world; r++) {
        devs[r].inbox = boxes + (size_t)r * P2P_INBOX_WORDS;
        for (int q = 0; q < world; q++) devs[r].peer[q] = boxes + (size_t)q * P2P_INBOX_WORDS;
        devs[r].epoch = nullptr;
        devs[r].err = h_err;
        devs[r].budget = 200000000ll;         // 2 s
        devs[r].world = world;
        devs[r].rank = r;
    }
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_devs), sizeof devs)) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_rows), n * 8)) != hipSuccess) return done(e);
    if ((e = hipMalloc(reinterpret_cast<void **>(&d_sums), n * 8)) != hipSuccess) return done(e);
    if ((e = hipMemcpy(d_devs, devs, sizeof devs, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
    if ((e = hipMemcpy(d_rows, rows_host, n * 8, hipMemcpyHostToDevice)) != hipSuccess) return done(e);
    if ((e = hipMemset(d_sums, 0, n * 8)) != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_p2p_world_debug, dim3(world), dim3(256), 0, stream, d_devs, rounds, seq0, d_rows, d_sums);
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    if ((e = hipMemcpy(sums_host, d_sums, n * 8, hipMemcpyDeviceToHost)) != hipSuccess) return done(e);
    *timed_out = *h_err;
    return done(hipSuccess);
}

hipError_t launch_p2p_send(hipStream_t stream, const P2PDev &p, const double *buf, int n) {
    if (n < 1 || n > P2P_MAX_DOUBLES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_p2p_send, dim3(1), dim3(256), 0, stream, p, buf, n);
    return hipGetLastError();
}

hipError_t launch_p2p_recv(hipStream_t stream, const P2PDev &p, double *buf, int n) {
    if (n < 1 || n > P2P_MAX_DOUBLES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_p2p_recv, dim3(1), dim3(256), 0, stream, p, buf, n);
    return hipGetLastError();
}

hipError_t launch_p2p_allreduce(hipStream_t stream, const P2PDev &p, double *buf, int n) {
    if (n < 1 || n > P2P_MAX_DOUBLES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_p2p_allreduce, dim3(1), dim3(256), 0, stream, p, buf, n);
    return hipGetLastError();
}

}  // namespace ccal
