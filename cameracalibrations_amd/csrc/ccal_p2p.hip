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
