// ccal_api.hip -- C ABI (include/ccal.h) and host driver of the device-side fit.
//
// Host work is limited to marshalling, launching and polling the device-side `done` flag;
// every arithmetic stage of calibrateCamera (reference src/detect_fit.jl:47) runs in the
// gfx950 kernels.  There is no CPU fallback: without a HIP device every entry fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ccal.h"
#include "ccal_comm.hpp"
#include "ccal_detect.hpp"
#include "ccal_kernels.hpp"
#include "ccal_state.hpp"

using namespace ccal;

#ifdef CCAL_STAMPS
namespace ccal { hipError_t debug_read_stamps(unsigned long long *out); hipError_t debug_read_wg_times(unsigned long long *out); hipError_t debug_read_init_stamps(unsigned long long *out); hipError_t debug_read_init_blocks(unsigned int *out); }
#endif

// direct xGMI exchange: fused into the prologue of k_lm_trial while a rank has at most this many
// workgroups (every one of them polls the inbox); beyond it one k_lm_exchange workgroup per trial does it.
// Decided from the GLOBAL problem size so that every rank of a fit takes the same path.
static constexpr int FUSED_XCHG_MAX_BLOCKS = 320;
// the pinned staging block of a context (64 KB) as a small ccal_calibrate call divides it
static constexpr size_t PIN_OBJ_OFF = 0, PIN_OBJ_BYTES = 8 * 1024, PIN_IMG_OFF = 8 * 1024, PIN_IMG_BYTES = 32 * 1024,
                        PIN_POSE_OFF = 40 * 1024, PIN_POSE_BYTES = 23 * 1024;

static constexpr size_t ACC_BYTES = sizeof(long long) * ACC_BUFS * ACC_MAX_REPLICAS * ACC_ROW;

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(CCAL_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct ccal_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // problem
    int P = 0, N = 0, width = 0, height = 0, max_iter = 30;
    double aspect = 1.0, eps = 1e-3;
    uint32_t flags = 0;
    bool planar_z0 = false;
    // device buffers
    float *d_obj = nullptr, *d_img_own = nullptr;
    const float *d_img = nullptr;
    size_t img_cap = 0, n_cap = 0;
    int p_cap = 0;
    double *d_ext0 = nullptr, *d_blk = nullptr, *d_red = nullptr;
    long long *d_acc = nullptr, *d_acc_scratch = nullptr;   // [3][R][ACC_ROW] exact accumulators (+ one set for diagnostics launches)
    int acc_R = 8;
    bool acc_dirty = true;             // the accumulators are not known to be all zero (cleared before the next solve)
    double *d_H = nullptr, *d_foc_partials = nullptr, *d_foc_red = nullptr, *d_gram = nullptr;
    int *d_fail = nullptr;
    double *d_intr0 = nullptr, *d_intr_scratch = nullptr;   // initial intrinsics [9] on the device
    LmState *d_st = nullptr, *d_st_scratch = nullptr;
    TraceRec *d_trace = nullptr;
    int max_trace = 4096;
    // grow-only device scratch + pinned, device-mapped staging of the map / metric / warp entry points
    void *d_scratch = nullptr, *h_pin = nullptr;
    size_t scratch_cap = 0;
    void *h_stage = nullptr;           // grow-only pinned staging of mid-size transfers
    size_t stage_cap = 0;
    unsigned long long pin_seq = 0;    // sequence number of the zero-copy path's completion word
    std::vector<double> h_rt_stage;
    // pinned host mirrors
    LmState *h_st = nullptr;
    int *h_flags = nullptr;
    // stage bookkeeping
    bool have_problem = false, have_img = false, have_intr = false, have_ext = false, have_result = false;
    long long *d_foc_acc = nullptr;    // [2][LM_FOC_ACC_WORDS] exact focal sums of the intrinsics stage (single GPU): two buffers rotate
    int foc_buf = 0;                   // the buffer the next intrinsics stage adds into
    bool foc_dirty = true;             // the buffers are not known to be zero
    bool focal_pending = false;        // pipeline: the focal solve of the last intrinsics stage has been left to the pose kernel's prologue
    bool have_H = false;               // d_H holds k_init_homography's output for the current image points
    bool fail0_clean = false;          // d_fail[0] is zero (no memset needed in front of the pose stage)
    bool intr_distorted = false;       // the initial intrinsics carry distortion (a caller's guess): no shortcut through d_H
    int init_hmode = 1;                // how k_init_extrinsics obtains the normalised homography when d_H is valid: 1 = OpenCV's sequence
                                       // with the DLT reused, 2 = the polished pixel-space homography decomposed directly (ccal_set_pose_init)
    double intr0[CCAL_NINTR] = {0};
    ccal_stats stats{};
    // multi-GPU
    ccal_allreduce_fn ar_fn = nullptr;
    void *ar_user = nullptr;
    int world = 1, rank = 0;
    Comm *comm = nullptr;              // native RCCL communicator (ccal_comm_init_rank)
    P2P *p2p = nullptr;                // direct xGMI row exchange (ccal_p2p_export / ccal_p2p_connect)
    bool p2p_no_fuse = false;          // ranks of several processes share this GPU: one-workgroup exchange, never whole grids polling
    bool p2p_shared_detected = false;  // ccal_p2p_connect2 found two ranks with the same device uuid
    int p2p_mode = 0;                  // ccal_p2p_set_exchange: 0 automatic, 1 one-workgroup exchange, 2 fused whatever was detected
    bool p2p_split = false;            // shards of one host thread share this GPU AND its stream: send / receive as two kernels
    Detector *detector = nullptr;      // corner detector buffers (ccal_detect_corners)
    static constexpr int DET_SLOTS = 3; // ccal_detect_corners_many: images in flight, each with a detector and a stream of its own
    Detector *det_slot[DET_SLOTS] = {nullptr, nullptr, nullptr};
    hipStream_t det_stream[DET_SLOTS] = {nullptr, nullptr, nullptr};
    int64_t n_global = 0;
    int lm_batch = 8;
    // optional per-sweep event timing
    bool time_sweeps = false;
    std::vector<hipEvent_t> ev;   // 2 * MAX_KINDS events (+ one pair for the terminating launch), created on demand
    int ev_used = 0;
    bool ev_final_used = false;
    // one solve = one replayable graph: begin + lm_batch x (sweep, solve) + state read-back
    hipGraphExec_t g_first = nullptr;
    struct GraphKey {             // everything baked into the captured launches
        const void *obj, *img, *blk, *st, *acc, *red, *trace, *intr0, *ext0, *host_poses;
        int N, P, batch, R, wpi;
        LmInit init;
    } graph_key{};
    bool run_merged = false;      // this solve's launches were merged with other contexts' (ccal_stats.sweep_form bit 16)
    hipEvent_t ev_join = nullptr; // merged launches run on another context's stream: whatever is still pending on this one is ordered in front of them
    bool join_needed = false;     // ... which can only be the case after a deferred wait (staged inputs copied by a launch nobody waited for) or a
                                  // solve of its own (the host is released by the progress word, ahead of the terminating launch's last stores)
    bool recording = false;       // ccal_fit_many is recording this context's launches for a merged launch (kernel by kernel, no graph)
    bool use_merge = true;        // (CCAL_NO_MERGE=1: ccal_fit_many enqueues every context's pipeline on its own stream)
    bool defer_sync = false;      // one-shot path: stage syncs and checks are batched at the end
    double dbg_enqueue_s = 0, dbg_total_s = 0;
    int dbg_runs = 0;
    bool graph_dirty = true, use_graph = true, adaptive_batch = true, spin_wait = true;
    int wpi = 1;                  // waves per image of the sweep, resolved once per problem (lm_waves_per_image)
    bool wpi_sharded = false;     // ... for a fit that was / was not image-sharded at that time (re-resolved when a transport is attached)
    // small drop-in calls (ccal_calibrate): inputs staged in the pinned block and copied by a kernel, poses written to it by the launch
    // that ends the solve -- no copy command, no read-back launch, one wait.  Layout of h_pin: template | corners | poses
    bool small_call = false;
    CopySeg *defer_copy = nullptr; // ccal_calibrate_many: the staged inputs' copy is described here instead of launched (one launch for all)
    size_t obj_staged = 0;        // words of the template waiting in the pinned block for the corners' copy launch (small calls only)
    double *host_poses = nullptr;
    bool use_solo = true;         // fits of a few images on one GPU run all their trials in ONE launch (CCAL_NO_SOLO=1: launch per trial)
    bool run_solo = false;
    // a solve in flight (lm_begin .. lm_end)
    double run_t0 = 0;
    long run_max_trials = 0, run_trials = 0;
    bool run_graphed = false;
};

// a pipeline that left its focal solve to the pose kernel and never launched it: the accumulators are in an unknown state
static void drop_pending_focal(ccal_ctx *c) {
    if (c->focal_pending) {
        c->focal_pending = false;
        c->foc_dirty = true;
    }
}

// d_intr0 is only complete once the pose stage that carries the pending focal solve has been enqueued: entry points that read the
// initial intrinsics refuse a context whose pipeline was abandoned between its two initialisation stages
static bool intr_incomplete(const ccal_ctx *c) { return c->focal_pending; }
#define REFUSE_PENDING_FOCAL(c, who)                                                                                              \
    do {                                                                                                                          \
        if (intr_incomplete(c))                                                                                                   \
            return fail(CCAL_E_STATE, who ": the intrinsics stage of an abandoned pipeline is incomplete (call ccal_init_intrinsics again)"); \
    } while (0)

static void free_problem_buffers(ccal_ctx *c) {
    (void)hipFree(c->d_obj); (void)hipFree(c->d_img_own); (void)hipFree(c->d_ext0); (void)hipFree(c->d_blk);
    (void)hipFree(c->d_H); (void)hipFree(c->d_foc_partials); (void)hipFree(c->d_gram);
    c->d_obj = nullptr; c->d_img_own = nullptr; c->d_ext0 = c->d_blk = nullptr;
    c->d_H = c->d_foc_partials = c->d_gram = nullptr;
    c->img_cap = 0; c->n_cap = 0; c->p_cap = 0;
}

extern "C" {

int ccal_version(void) { return CCAL_VERSION; }

static bool sharded(const ccal_ctx *c);
// Waves per image of the sweep, once per problem (and again when a transport is attached: an image-sharded fit keeps one-pass
// boards at one wave per image -- the split's 3 % are not worth four times as many workgroups polling a fused exchange)
static void resolve_wpi(ccal_ctx *c) {
    c->wpi_sharded = sharded(c);
    if (c->P <= 0 || c->N <= 0) { c->wpi = 1; return; }
    const int w = (c->wpi_sharded && c->P <= 64) ? 1 : lm_waves_per_image(c->N, c->P, c->wpi_sharded);
    if (w != c->wpi) c->graph_dirty = true;
    c->wpi = w;
}
const char *ccal_last_error(void) { return g_err.c_str(); }

int ccal_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

int ccal_create(int device, ccal_ctx **out) {
    if (!out) return fail(CCAL_E_ARG, "ccal_create: out == NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(CCAL_E_DEVICE, "ccal_create: no HIP device visible (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(CCAL_E_ARG, "ccal_create: bad device index");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CCAL_E_DEVICE, std::string("ccal_create: kernels are built for gfx950 only, device is ") +
                                       prop.gcnArchName);
    HIPCHK(hipSetDevice(device));
    ccal_ctx *c = new ccal_ctx();
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIPCHK(hipMalloc(&c->d_st, sizeof(LmState) * 2));          // S_k lives in d_st[k & 1]
    HIPCHK(hipMalloc(&c->d_st_scratch, sizeof(LmState) * 2));
    HIPCHK(hipMalloc(&c->d_red, sizeof(double) * NRED));
    HIPCHK(hipMalloc(&c->d_acc, ACC_BYTES));
    // (clears on the context's own stream, not on the legacy default stream: an operation there fails outright while another host
    //  thread has a stream in capture mode)
    HIPCHK(hipMemsetAsync(c->d_acc, 0, ACC_BYTES, c->own_stream));                 // a new context starts in the steady state every solve leaves behind (accumulators
    c->acc_dirty = false;                                      // all zero): its first fit needs no clears and can share merged launches
    HIPCHK(hipMalloc(&c->d_acc_scratch, ACC_BYTES));
    HIPCHK(hipMemsetAsync(c->d_acc_scratch, 0, ACC_BYTES, c->own_stream));
    HIPCHK(hipMalloc(&c->d_foc_red, sizeof(double) * NFOC));
    HIPCHK(hipMalloc(&c->d_foc_acc, sizeof(long long) * 2 * LM_FOC_ACC_WORDS));
    HIPCHK(hipMemsetAsync(c->d_foc_acc, 0, sizeof(long long) * 2 * LM_FOC_ACC_WORDS, c->own_stream));
    c->foc_dirty = false;
    HIPCHK(hipMalloc(&c->d_fail, sizeof(int) * 4));
    HIPCHK(hipMemsetAsync(c->d_fail, 0, sizeof(int) * 4, c->own_stream));
    HIPCHK(hipStreamSynchronize(c->own_stream));
    HIPCHK(hipMalloc(&c->d_intr0, sizeof(double) * 9));
    HIPCHK(hipMalloc(&c->d_intr_scratch, sizeof(double) * 9));
    HIPCHK(hipMalloc(&c->d_trace, sizeof(TraceRec) * c->max_trace));
    HIPCHK(hipHostMalloc(&c->h_pin, 64 * 1024, hipHostMallocMapped));
    memset(c->h_pin, 0, 64 * 1024);
    HIPCHK(hipHostMalloc(&c->h_st, sizeof(LmState), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(&c->h_flags, sizeof(int) * 64, hipHostMallocDefault));   // flags | intr staging | progress
    const char *b = getenv("CCAL_LM_BATCH");
    if (b && atoi(b) > 0) { c->lm_batch = atoi(b); c->adaptive_batch = false; }
    const char *g = getenv("CCAL_NO_GRAPH");
    if (g && atoi(g) != 0) c->use_graph = false;
    if (const char *so = getenv("CCAL_NO_SOLO")) c->use_solo = atoi(so) == 0;
    if (const char *so = getenv("CCAL_NO_MERGE")) c->use_merge = atoi(so) == 0;
    const char *sp = getenv("CCAL_NO_SPIN");
    if (sp && atoi(sp) != 0) c->spin_wait = false;
    if (const char *hm = getenv("CCAL_INIT_HMODE")) {       // experiments: 0 = OpenCV's own sequence (DLT + polish), 1, 2
        const int v = atoi(hm);
        if (v >= 0 && v <= 2) c->init_hmode = v;
    }
    *out = c;
    return CCAL_OK;
}

void ccal_destroy(ccal_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_problem_buffers(c);
    (void)hipFree(c->d_st); (void)hipFree(c->d_st_scratch); (void)hipFree(c->d_red); (void)hipFree(c->d_foc_red); (void)hipFree(c->d_foc_acc); (void)hipFree(c->d_fail);
    (void)hipFree(c->d_acc); (void)hipFree(c->d_acc_scratch);
    (void)hipFree(c->d_trace); (void)hipFree(c->d_intr0); (void)hipFree(c->d_intr_scratch);
    if (c->g_first) (void)hipGraphExecDestroy(c->g_first);
    (void)hipHostFree(c->h_st); (void)hipHostFree(c->h_flags); (void)hipHostFree(c->h_pin); (void)hipFree(c->d_scratch);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (auto &e : c->ev) (void)hipEventDestroy(e);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    delete c->comm;
    delete c->p2p;
    delete c->detector;
    for (auto &d : c->det_slot) delete d;
    for (auto &st : c->det_stream) if (st) (void)hipStreamDestroy(st);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int ccal_set_stream(ccal_ctx *c, void *hip_stream) {
    if (!c) return fail(CCAL_E_ARG, "ccal_set_stream: ctx == NULL");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return CCAL_OK;
}

int ccal_set_allreduce(ccal_ctx *c, ccal_allreduce_fn fn, void *user, int32_t world_size, int32_t rank,
                       int64_t n_images_global) {
    if (!c) return fail(CCAL_E_ARG, "ccal_set_allreduce: ctx == NULL");
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(CCAL_E_ARG, "ccal_set_allreduce: bad rank/world");
    if (world_size > 1 && (!fn || n_images_global < 1)) return fail(CCAL_E_ARG, "ccal_set_allreduce: hook missing");
    if (fn && n_images_global < 1) return fail(CCAL_E_ARG, "ccal_set_allreduce: n_images_global missing");
    c->graph_dirty = true;
    delete c->comm;
    c->comm = nullptr;
    delete c->p2p;
    c->p2p = nullptr;
    c->ar_fn = fn;                      // a hook with world_size == 1 is honoured (used by tests)
    c->ar_user = user;
    c->world = world_size;
    c->rank = rank;
    c->n_global = n_images_global > 0 ? n_images_global : 0;
    return CCAL_OK;
}

int ccal_comm_unique_id(char id_out[128]) {
    if (!id_out) return fail(CCAL_E_ARG, "ccal_comm_unique_id: NULL argument");
    std::string err;
    if (Comm::unique_id(id_out, err) != 0) return fail(CCAL_E_COMM, err);
    return CCAL_OK;
}

int ccal_comm_init_rank(ccal_ctx *c, int32_t world_size, int32_t rank, const char id[128], int64_t n_images_global) {
    if (!c || !id) return fail(CCAL_E_ARG, "ccal_comm_init_rank: NULL argument");
    if (world_size < 1 || rank < 0 || rank >= world_size || n_images_global < 1)
        return fail(CCAL_E_ARG, "ccal_comm_init_rank: bad rank/world/image count");
    HIPCHK(hipSetDevice(c->device));
    if (!c->comm) c->comm = new Comm();
    std::string err;
    if (c->comm->init(world_size, rank, id, err) != 0) {
        delete c->comm;
        c->comm = nullptr;
        return fail(CCAL_E_COMM, err);
    }
    c->graph_dirty = true;
    c->ar_fn = nullptr;
    delete c->p2p;
    c->p2p = nullptr;
    c->world = world_size;
    c->rank = rank;
    c->n_global = n_images_global;
    return CCAL_OK;
}

static bool via_p2p_ready(const ccal_ctx *c) { return c->p2p && c->p2p->ready(); }

int ccal_p2p_export(ccal_ctx *c, int32_t world_size, int32_t rank, char handle_out[64]) {
    if (!c || !handle_out) return fail(CCAL_E_ARG, "ccal_p2p_export: NULL argument");
    if (world_size < 1 || world_size > P2P_MAX_RANKS || rank < 0 || rank >= world_size)
        return fail(CCAL_E_ARG, "ccal_p2p_export: bad rank/world (1..16 ranks of one node)");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    delete c->p2p;
    c->graph_dirty = true;             // a captured solve may hold the old inbox pointers
    c->p2p_shared_detected = false;
    c->p2p_no_fuse = c->p2p_mode == 1;
    c->p2p = new P2P();
    std::string err;
    if (c->p2p->create(world_size, rank, handle_out, err) != 0) {
        delete c->p2p;
        c->p2p = nullptr;
        return fail(CCAL_E_COMM, err);
    }
    return CCAL_OK;
}

int ccal_p2p_connect(ccal_ctx *c, const char *handles, int64_t n_images_global) {
    if (!c || !handles) return fail(CCAL_E_ARG, "ccal_p2p_connect: NULL argument");
    if (!c->p2p) return fail(CCAL_E_STATE, "ccal_p2p_connect: ccal_p2p_export has not been called");
    if (n_images_global < 1) return fail(CCAL_E_ARG, "ccal_p2p_connect: bad image count");
    HIPCHK(hipSetDevice(c->device));
    std::string err;
    if (c->p2p->connect(handles, err) != 0) {
        delete c->p2p;
        c->p2p = nullptr;
        return fail(CCAL_E_COMM, err);
    }
    c->graph_dirty = true;
    c->ar_fn = nullptr;
    delete c->comm;
    c->comm = nullptr;
    c->world = c->p2p->world();
    c->rank = c->p2p->rank();
    c->n_global = n_images_global;
    return CCAL_OK;
}

// The same pair with the rank's device id travelling inside the blob (64-byte inbox handle + 16-byte device uuid): the
// library sees for itself when two ranks share a GPU and takes the one-workgroup exchange then (a waiting rank must hold one
// compute unit, not a grid that keeps the sender off the chip) -- callers no longer have to gather the ids and call
// ccal_p2p_set_exchange themselves; ccal_p2p_set_exchange stays as the override.
int ccal_p2p_export2(ccal_ctx *c, int32_t world_size, int32_t rank, char blob_out[CCAL_P2P_BLOB_BYTES]) {
    if (!blob_out) return fail(CCAL_E_ARG, "ccal_p2p_export2: NULL argument");
    memset(blob_out, 0, CCAL_P2P_BLOB_BYTES);
    int rc = ccal_p2p_export(c, world_size, rank, blob_out);
    if (rc != CCAL_OK) return rc;
    return ccal_device_uuid(c->device, blob_out + P2P_HANDLE_BYTES);
}

int ccal_p2p_connect2(ccal_ctx *c, const char *blobs, int64_t n_images_global) {
    if (!c || !blobs) return fail(CCAL_E_ARG, "ccal_p2p_connect2: NULL argument");
    if (!c->p2p) return fail(CCAL_E_STATE, "ccal_p2p_connect2: ccal_p2p_export2 has not been called");
    const int W = c->p2p->world();
    std::vector<char> handles((size_t)W * P2P_HANDLE_BYTES);
    bool shared = false;
    for (int r = 0; r < W; r++) {
        memcpy(handles.data() + (size_t)r * P2P_HANDLE_BYTES, blobs + (size_t)r * CCAL_P2P_BLOB_BYTES, P2P_HANDLE_BYTES);
        for (int q = 0; q < r; q++)
            shared = shared || memcmp(blobs + (size_t)r * CCAL_P2P_BLOB_BYTES + P2P_HANDLE_BYTES,
                                      blobs + (size_t)q * CCAL_P2P_BLOB_BYTES + P2P_HANDLE_BYTES, 16) == 0;
    }
    const int rc = ccal_p2p_connect(c, handles.data(), n_images_global);
    if (rc != CCAL_OK) return rc;
    c->p2p_shared_detected = shared;
    c->p2p_no_fuse = c->p2p_mode == 1 || (c->p2p_mode == 0 && shared);
    return CCAL_OK;
}

int ccal_p2p_shared_device(ccal_ctx *c) { return (c && c->p2p_shared_detected) ? 1 : 0; }

int ccal_p2p_selftest(ccal_ctx *c, int32_t rounds) {
    if (!c || rounds < 1) return fail(CCAL_E_ARG, "ccal_p2p_selftest: bad argument");
    if (!via_p2p_ready(c)) return fail(CCAL_E_STATE, "ccal_p2p_selftest: ccal_p2p_connect has not been called");
    HIPCHK(hipSetDevice(c->device));
    const int W = c->p2p->world(), R = c->p2p->rank();
    double *d = nullptr, h[P2P_MAX_DOUBLES];
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&d), sizeof h));
    int bad = 0;
    for (int t = 0; t < rounds && !bad; t++) {
        // every rank contributes (rank + 1)(j + 1) + t / 7: the sum is exact in binary64 only if the
        // ranks' rows are added in rank order on every rank, which is also what is checked
        for (int j = 0; j < P2P_MAX_DOUBLES; j++) h[j] = (double)(R + 1) * (double)(j + 1) + (double)t / 7.0;
        hipError_t e = hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_p2p_allreduce(c->stream, c->p2p->dev(), d, P2P_MAX_DOUBLES);
        if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(d); HIPCHK(e); }
        for (int j = 0; j < P2P_MAX_DOUBLES; j++) {
            double want = 0.0;
            for (int r = 0; r < W; r++) want += (double)(r + 1) * (double)(j + 1) + (double)t / 7.0;
            if (h[j] != want) bad++;
        }
    }
    (void)hipFree(d);
    std::string err;
    if (c->p2p->check_and_clear(err)) return fail(CCAL_E_COMM, "ccal_p2p_selftest: " + err);
    if (bad) return fail(CCAL_E_COMM, "ccal_p2p_selftest: exchanged sums differ from the expected values");
    return CCAL_OK;
}

int ccal_p2p_set_exchange(ccal_ctx *c, int32_t mode) {
    if (!c) return fail(CCAL_E_ARG, "ccal_p2p_set_exchange: ctx == NULL");
    if (mode < 0 || mode > 2)
        return fail(CCAL_E_ARG, "ccal_p2p_set_exchange: mode is 0 (automatic), 1 (one-workgroup exchange) or 2 (fused even on a shared device)");
    c->p2p_mode = mode;
    c->p2p_no_fuse = mode == 1 || (mode == 0 && c->p2p_shared_detected);
    c->graph_dirty = true;             // a captured solve holds the launches of the old mode
    return CCAL_OK;
}

int ccal_device_uuid(int32_t device, char uuid_out[16]) {
    if (!uuid_out) return fail(CCAL_E_ARG, "ccal_device_uuid: NULL argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return fail(CCAL_E_ARG, "ccal_device_uuid: bad device index");
    hipUUID id;
    HIPCHK(hipDeviceGetUuid(&id, device));
    static_assert(sizeof id.bytes == 16, "uuid size");
    memcpy(uuid_out, id.bytes, 16);
    return CCAL_OK;
}

int ccal_set_problem(ccal_ctx *c, const float *obj, int32_t P, int32_t N, int32_t width, int32_t height,
                     double aspect, uint32_t flags, int32_t max_iter, double eps) {
    if (!c || !obj) return fail(CCAL_E_ARG, "ccal_set_problem: NULL argument");
    if (P < 4 || N < 1 || max_iter < 1 || !(eps >= 0)) return fail(CCAL_E_ARG, "ccal_set_problem: bad sizes/criteria");
    const uint32_t known = CCAL_USE_INTRINSIC_GUESS | CCAL_FIX_ASPECT_RATIO | CCAL_FIX_PRINCIPAL_POINT |
                           CCAL_ZERO_TANGENT_DIST | CCAL_FIX_FOCAL_LENGTH | CCAL_FIX_K1 | CCAL_FIX_K2 | CCAL_FIX_K3 |
                           CCAL_FIX_TANGENT_DIST;
    if (flags & ~known)
        return fail(CCAL_E_ARG, "ccal_set_problem: unsupported flag (rational / thin-prism / tilted models are "
                                "outside the reference's use, src/detect_fit.jl:40)");
    if ((flags & CCAL_FIX_ASPECT_RATIO) && !(flags & CCAL_USE_INTRINSIC_GUESS) && !(aspect >= 0.01 && aspect <= 100))
        return fail(CCAL_E_ARG, "ccal_set_problem: aspect ratio outside [0.01, 100]");
    HIPCHK(hipSetDevice(c->device));
    bool planar = true;
    for (int p = 0; p < P; p++)
        if (std::fabs(obj[3 * p + 2]) > 1e-5f) planar = false;
    c->planar_z0 = planar;
    if ((size_t)N > c->n_cap || P > c->p_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        free_problem_buffers(c);
        const size_t n = (size_t)N;
        HIPCHK(hipMalloc(&c->d_obj, sizeof(float) * 3 * P));
        HIPCHK(hipMalloc(&c->d_ext0, sizeof(double) * n * EXT));
        HIPCHK(hipMalloc(&c->d_blk, sizeof(double) * 2 * n * BLK));
        HIPCHK(hipMalloc(&c->d_H, sizeof(double) * n * 18));    // per image: polished homography | DLT result
        HIPCHK(hipMalloc(&c->d_foc_partials, sizeof(double) * n * NFOC));
        c->n_cap = n;
        c->p_cap = P;
    }
    if (c->small_call && sizeof(float) * 3 * P <= PIN_OBJ_BYTES) {
        memcpy(static_cast<char *>(c->h_pin) + PIN_OBJ_OFF, obj, sizeof(float) * 3 * P);
        c->obj_staged = (size_t)3 * P;     // (copied by the launch that copies the corners: ccal_set_image_points follows at once)
    } else {
        HIPCHK(hipMemcpyAsync(c->d_obj, obj, sizeof(float) * 3 * P, hipMemcpyHostToDevice, c->stream));
    }
    // (no clearing of blk / ext0: a solve's first launch reads the six pose words of ext0 and nothing of blk, and every sweep
    //  writes its images' blocks whole -- two memset launches per call were a tenth of a reference-sized fit)
    if (!c->defer_sync) HIPCHK(hipStreamSynchronize(c->stream)); else c->join_needed = true;   // `obj` may be released by the caller after return
    c->P = P; c->N = N; c->width = width; c->height = height; c->aspect = aspect; c->flags = flags;
    c->max_iter = max_iter; c->eps = eps;
    resolve_wpi(c);
    const int sweep_grid = std::min(lm_sweep_blocks(N), lm_max_grid());
    const int old_R = c->acc_R;
    c->acc_R = lm_acc_replicas(sweep_grid);
    if (const char *r = getenv("CCAL_ACC_REPLICAS")) {      // experiments: 1, 2, 4, ... 64
        const int v = atoi(r);
        if (v >= 1 && v <= ACC_MAX_REPLICAS && (v & (v - 1)) == 0 && (sweep_grid + v - 1) / v <= 2000) c->acc_R = v;
    }
    // the buffer layout depends on the replica count; with the same count the last solve's terminating launch has left
    // every buffer cleared (acc_dirty is false then) and the next solve needs no 0.5 MB memset
    if (c->acc_R != old_R) c->acc_dirty = true;
    c->have_problem = true;
    c->have_img = c->have_intr = c->have_ext = c->have_result = false;
    c->have_H = false;
    drop_pending_focal(c);
    c->d_img = nullptr;
    memset(&c->stats, 0, sizeof c->stats);
    return CCAL_OK;
}

int ccal_set_image_points(ccal_ctx *c, const float *img_host) {
    if (!c || !img_host) return fail(CCAL_E_ARG, "ccal_set_image_points: NULL argument");
    if (!c->have_problem) return fail(CCAL_E_STATE, "ccal_set_image_points: call ccal_set_problem first");
    HIPCHK(hipSetDevice(c->device));
    const size_t need = sizeof(float) * 2 * (size_t)c->P * c->N;
    if (need > c->img_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_img_own);
        c->d_img_own = nullptr;
        HIPCHK(hipMalloc(&c->d_img_own, need));
        c->img_cap = need;
    }
    if (c->small_call && need <= PIN_IMG_BYTES) {
        memcpy(static_cast<char *>(c->h_pin) + PIN_IMG_OFF, img_host, need);
        if (c->defer_copy != nullptr) {
            *c->defer_copy = CopySeg{reinterpret_cast<const uint32_t *>(static_cast<char *>(c->h_pin) + PIN_OBJ_OFF), reinterpret_cast<uint32_t *>(c->d_obj),
                                     c->obj_staged, reinterpret_cast<const uint32_t *>(static_cast<char *>(c->h_pin) + PIN_IMG_OFF),
                                     reinterpret_cast<uint32_t *>(c->d_img_own), need / 4};
        } else {
            HIPCHK(launch_copy_words2(c->stream, static_cast<char *>(c->h_pin) + PIN_OBJ_OFF, c->d_obj, c->obj_staged,
                                      static_cast<char *>(c->h_pin) + PIN_IMG_OFF, c->d_img_own, need / 4));
        }
        c->obj_staged = 0;
    } else {
        if (c->obj_staged) {
            HIPCHK(launch_copy_words(c->stream, static_cast<char *>(c->h_pin) + PIN_OBJ_OFF, c->d_obj, c->obj_staged));
            c->obj_staged = 0;
        }
        HIPCHK(hipMemcpyAsync(c->d_img_own, img_host, need, hipMemcpyHostToDevice, c->stream));
    }
    if (!c->defer_sync) HIPCHK(hipStreamSynchronize(c->stream)); else c->join_needed = true;
    c->d_img = c->d_img_own;
    c->have_img = true;
    c->have_H = false;
    drop_pending_focal(c);
    c->have_result = false;
    return CCAL_OK;
}

int ccal_set_image_points_device(ccal_ctx *c, const float *img_device) {
    if (!c || !img_device) return fail(CCAL_E_ARG, "ccal_set_image_points_device: NULL argument");
    if (!c->have_problem) return fail(CCAL_E_STATE, "ccal_set_image_points_device: call ccal_set_problem first");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, img_device) != hipSuccess || attr.type != hipMemoryTypeDevice) {
        (void)hipGetLastError();
        return fail(CCAL_E_ARG, "ccal_set_image_points_device: not a device pointer");
    }
    if (attr.device != c->device)
        return fail(CCAL_E_ARG, "ccal_set_image_points_device: the pointer lives on device " + std::to_string(attr.device) +
                                    ", the context on device " + std::to_string(c->device));
    c->d_img = img_device;
    c->have_img = true;
    c->have_H = false;                 // (the same pointer may hold other corners now)
    drop_pending_focal(c);
    c->have_result = false;
    return CCAL_OK;
}

int ccal_set_intrinsics(ccal_ctx *c, const double intr[CCAL_NINTR]) {
    if (!c || !intr) return fail(CCAL_E_ARG, "ccal_set_intrinsics: NULL argument");
    if (!c->have_problem) return fail(CCAL_E_STATE, "ccal_set_intrinsics: call ccal_set_problem first");
    for (int a = 0; a < CCAL_NINTR; a++)
        if (!std::isfinite(intr[a])) return fail(CCAL_E_ARG, "ccal_set_intrinsics: non-finite value");
    if (!(intr[0] > 0 && intr[1] > 0)) return fail(CCAL_E_ARG, "ccal_set_intrinsics: focal lengths must be > 0");
    memcpy(c->intr0, intr, sizeof c->intr0);
    c->intr_distorted = false;
    drop_pending_focal(c);
    for (int a = 4; a < CCAL_NINTR; a++) c->intr_distorted = c->intr_distorted || intr[a] != 0.0;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d_intr0, c->intr0, sizeof c->intr0, hipMemcpyHostToDevice, c->stream));
    if (!c->defer_sync) HIPCHK(hipStreamSynchronize(c->stream)); else c->join_needed = true;
    c->have_intr = true;
    c->have_result = false;
    return CCAL_OK;
}

int ccal_set_extrinsics(ccal_ctx *c, const double *rt_host) {
    if (!c || !rt_host) return fail(CCAL_E_ARG, "ccal_set_extrinsics: NULL argument");
    if (!c->have_problem) return fail(CCAL_E_STATE, "ccal_set_extrinsics: call ccal_set_problem first");
    HIPCHK(hipSetDevice(c->device));
    std::vector<double> padded((size_t)c->N * EXT, 0.0);
    for (int i = 0; i < c->N; i++)
        for (int k = 0; k < 6; k++) padded[(size_t)i * EXT + k] = rt_host[(size_t)i * 6 + k];
    HIPCHK(hipMemcpyAsync(c->d_ext0, padded.data(), sizeof(double) * padded.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_ext = true;
    c->have_result = false;
    return CCAL_OK;
}

// ---- A4 / A5: closed-form initialisation on the device ---------------------------------
static bool via_p2p(const ccal_ctx *c) { return c->p2p && c->p2p->ready(); }
static bool sharded(const ccal_ctx *c) { return c->ar_fn != nullptr || (c->comm && c->comm->ready()) || via_p2p(c); }

static int run_allreduce(ccal_ctx *c, double *buf, int n) {
    if (via_p2p(c)) {
        HIPCHK(launch_p2p_allreduce(c->stream, c->p2p->dev(), buf, n));
        return CCAL_OK;
    }
    if (c->comm && c->comm->ready()) {
        std::string err;
        if (c->comm->all_reduce_sum(buf, n, c->stream, err) != 0) return fail(CCAL_E_COMM, err);
        return CCAL_OK;
    }
    if (!c->ar_fn) return CCAL_OK;
    const int rc = c->ar_fn(c->ar_user, buf, n, (void *)c->stream);
    if (rc != 0) return fail(CCAL_E_COMM, "all-reduce hook returned " + std::to_string(rc));
    return CCAL_OK;
}

static LmInit make_init(const ccal_ctx *c) {
    LmInit in;
    memset(&in, 0, sizeof in);
    in.max_iter = c->max_iter;
    in.eps = c->eps;
    in.max_trace = c->max_trace;
    in.P = c->P;
    in.N_local = c->N;
    in.nblocks = lm_sweep_blocks(c->N);
    in.N_global = (sharded(c) && c->n_global > 0) ? c->n_global : c->N;
    // parameter mask from the flag word
    const uint32_t f = c->flags;
    for (int a = 0; a < 9; a++) in.mask[a] = 1;
    if (f & CCAL_FIX_ASPECT_RATIO) in.mask[0] = 0;
    if (f & CCAL_FIX_FOCAL_LENGTH) in.mask[0] = in.mask[1] = 0;
    if (f & CCAL_FIX_PRINCIPAL_POINT) in.mask[2] = in.mask[3] = 0;
    if (f & (CCAL_ZERO_TANGENT_DIST | CCAL_FIX_TANGENT_DIST)) in.mask[6] = in.mask[7] = 0;
    if (f & CCAL_FIX_K1) in.mask[4] = 0;
    if (f & CCAL_FIX_K2) in.mask[5] = 0;
    if (f & CCAL_FIX_K3) in.mask[8] = 0;
    in.zero_tangent = (f & CCAL_ZERO_TANGENT_DIST) ? 1 : 0;
    in.aspect = (f & CCAL_FIX_ASPECT_RATIO) ? c->aspect : 0.0;
    in.aspect_from_guess = ((f & CCAL_FIX_ASPECT_RATIO) && (f & CCAL_USE_INTRINSIC_GUESS)) ? 1 : 0;
    return in;
}

// read back what the host mirrors and the failure flags of the initialisation kernels: enqueue ...
static int finish_init_enqueue(ccal_ctx *c) {
    double *h_intr = reinterpret_cast<double *>(c->h_flags + 4);   // pinned staging (flags: ints 0..3)
    HIPCHK(launch_init_readback(c->stream, c->d_intr0, c->d_fail, h_intr, c->h_flags));
    return CCAL_OK;
}

// ... and, once the copies have landed, check
static int finish_init_check(ccal_ctx *c, bool intr, bool extr) {
    const double *h_intr = reinterpret_cast<const double *>(c->h_flags + 4);
    if (via_p2p(c)) {              // a silent peer explains (and outranks) the NaN it left behind
        std::string err;
        if (c->p2p->check_and_clear(err)) return fail(CCAL_E_COMM, "ccal_init_intrinsics: " + err);
    }
    if (intr) {
        if (c->h_flags[1] != 0) return fail(CCAL_E_DEGENERATE, "ccal_init_intrinsics: homography estimation failed");
        for (int a = 0; a < 9; a++) c->intr0[a] = h_intr[a];
        for (int a = 0; a < 4; a++)
            if (!std::isfinite(c->intr0[a])) return fail(CCAL_E_DEGENERATE, "ccal_init_intrinsics: non-finite focal length");
    }
    if (extr && c->h_flags[0] != 0)
        return fail(CCAL_E_DEGENERATE, "ccal_init_extrinsics: pose initialisation failed for " +
                                           std::to_string(c->h_flags[0]) + " image(s)");
    return CCAL_OK;
}

static int signal_and_spin(ccal_ctx *c);

// (the wait is a spin on a completion word in pinned memory: a blocking stream synchronisation costs tens of
//  microseconds of wake-up latency, comparable to the initialisation kernels themselves)
static int finish_init(ccal_ctx *c, bool intr, bool extr) {
    int rc = finish_init_enqueue(c);
    if (rc) return rc;
    if ((rc = signal_and_spin(c)) != CCAL_OK) return rc;
    return finish_init_check(c, intr, extr);
}

// The initialisation kernels come in a latency form and a throughput form whose trial errors are summed differently; which
// one runs is decided from the GLOBAL image count, so that every shard of a fit and the unsharded fit take the same one.
static bool init_spec(const ccal_ctx *c) {
    const int64_t ng = (sharded(c) && c->n_global > 0) ? c->n_global : c->N;
    return ng <= (int64_t)init_simd_count();
}

// Not absurdly many images on this GPU: its share of the focal least squares is summed exactly by k_init_homography itself
// (single GPU: solved from those sums; image-sharded: the decoded sums are what the ranks add up -- one rank reproduces the
// single-GPU bits)
static bool focal_exact(const ccal_ctx *c) { return c->N <= LM_FOC_MAX_IMAGES; }

// A4 in two halves, so that shards sharing one in-order stream (p2p_split) can put every shard's send in front
// of every shard's receive:  a = homographies + local focal sums (+ send),  b = (receive | all-reduce) + 2x2 solve
static int init_intrinsics_a(ccal_ctx *c) {
    if (!c) return fail(CCAL_E_ARG, "ccal_init_intrinsics: ctx == NULL");
    if (!c->have_problem || !c->have_img) return fail(CCAL_E_STATE, "ccal_init_intrinsics: problem/image points not set");
    if (!c->planar_z0)
        return fail(CCAL_E_NONPLANAR, "For non-planar calibration rigs the initial intrinsic matrix must be specified");
    HIPCHK(hipSetDevice(c->device));
    c->run_t0 = now_s();
    // this GPU holds every image: the focal sums are formed exactly by the homography kernel itself (focal_exact)
    long long *foc_acc = nullptr;
    if (focal_exact(c)) {
        if (c->focal_pending) c->foc_dirty = true;         // (a pipeline was abandoned between its two init stages)
        c->focal_pending = false;
        if (c->foc_dirty) {
            HIPCHK(hipMemsetAsync(c->d_foc_acc, 0, sizeof(long long) * 2 * LM_FOC_ACC_WORDS, c->stream));
            c->foc_dirty = false;
            c->foc_buf = 0;
        }
        foc_acc = c->d_foc_acc + (size_t)c->foc_buf * LM_FOC_ACC_WORDS;
    }
    HIPCHK(launch_init_homography(c->stream, c->d_obj, c->d_img, c->d_H, c->d_foc_partials, foc_acc,
                                  (foc_acc != nullptr && c->defer_sync) ? c->d_fail : nullptr, c->N, c->P, c->width, c->height, init_spec(c)));
    c->have_H = true;
    if (sharded(c)) {
        if (foc_acc != nullptr) {
            HIPCHK(launch_init_focal_acc_row(c->stream, foc_acc, c->d_foc_acc + (size_t)(c->foc_buf ^ 1) * LM_FOC_ACC_WORDS, c->d_foc_red));
            c->foc_buf ^= 1;
        } else {
            HIPCHK(launch_init_focal_reduce(c->stream, c->d_foc_partials, c->d_foc_red, c->N));
        }
        if (via_p2p(c) && c->p2p_split) HIPCHK(launch_p2p_send(c->stream, c->p2p->dev(), c->d_foc_red, NFOC));
    }
    return CCAL_OK;
}

static int init_intrinsics_b(ccal_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    const double aspect = (c->flags & CCAL_FIX_ASPECT_RATIO) ? c->aspect : 0.0;
    int rc = CCAL_OK;
    if (focal_exact(c) && !sharded(c)) {
        // the sums are in the exact accumulators.  Stages as one pipeline: the pose kernel that follows solves in its prologue
        // (no kernel here); called on its own: one small kernel.  Same sums, same arithmetic, same bits.
        if (c->defer_sync && c->init_hmode != 0) {
            c->focal_pending = true;
        } else {
            long long *cur = c->d_foc_acc + (size_t)c->foc_buf * LM_FOC_ACC_WORDS, *other = c->d_foc_acc + (size_t)(c->foc_buf ^ 1) * LM_FOC_ACC_WORDS;
            HIPCHK(launch_init_focal_acc(c->stream, cur, other, c->d_intr0, c->d_fail, c->width, c->height, aspect));
            c->foc_buf ^= 1;
        }
    } else if (!sharded(c)) {
        // this GPU holds every image: sums and the 2x2 solve in one launch
        HIPCHK(launch_init_focal_reduce_solve(c->stream, c->d_foc_partials, c->d_foc_red, c->N, c->d_intr0, c->d_fail, c->width,
                                              c->height, aspect));
    } else {
        if (via_p2p(c) && c->p2p_split) HIPCHK(launch_p2p_recv(c->stream, c->p2p->dev(), c->d_foc_red, NFOC));
        else if ((rc = run_allreduce(c, c->d_foc_red, NFOC)) != CCAL_OK) return rc;
        HIPCHK(launch_init_focal_solve(c->stream, c->d_foc_red, c->d_intr0, c->d_fail, c->width, c->height, aspect));
    }
    if (!c->defer_sync && (rc = finish_init(c, true, false)) != CCAL_OK) return rc;
    c->have_intr = true;
    c->fail0_clean = true;             // (focal_solve zeroes the pose stage's failure counter)
    c->intr_distorted = false;         // cvInitIntrinsicParams2D leaves the distortion at zero
    c->have_result = false;
    c->stats.t_init_intr_s = now_s() - c->run_t0;
    return CCAL_OK;
}

int ccal_init_intrinsics(ccal_ctx *c) {
    const int rc = init_intrinsics_a(c);
    return rc ? rc : init_intrinsics_b(c);
}

int ccal_init_extrinsics(ccal_ctx *c) {
    if (!c) return fail(CCAL_E_ARG, "ccal_init_extrinsics: ctx == NULL");
    if (!c->have_problem || !c->have_img || !c->have_intr)
        return fail(CCAL_E_STATE, "ccal_init_extrinsics: intrinsics / image points not set");
    if (!c->planar_z0) return fail(CCAL_E_NONPLANAR, "ccal_init_extrinsics: only planar z = 0 templates are supported");
    HIPCHK(hipSetDevice(c->device));
    const double t0 = now_s();
    // word 0: failed poses (word 1: focal solve); the focal solve kernel of the intrinsics stage has already zeroed it
    if (!c->fail0_clean) HIPCHK(hipMemsetAsync(c->d_fail, 0, sizeof(int), c->stream));
    c->fail0_clean = false;
    // the pixel-space homographies of the intrinsics stage stand in for the second, normalised homography OpenCV computes
    // here whenever they exist for these corners and the intrinsics carry no distortion (k_init_extrinsics, HMODE)
    const bool use_H = c->have_H && !c->intr_distorted && c->init_hmode != 0;
    InitFocal focal{};
    const bool fused_focal = c->focal_pending;
    if (fused_focal) {
        if (!use_H) return fail(CCAL_E_STATE, "ccal_init_extrinsics: pipeline state lost between the two initialisation stages");
        focal.acc = c->d_foc_acc + (size_t)c->foc_buf * LM_FOC_ACC_WORDS;
        focal.acc_clear = c->d_foc_acc + (size_t)(c->foc_buf ^ 1) * LM_FOC_ACC_WORDS;
        focal.intr0_out = c->d_intr0;
        focal.width = c->width; focal.height = c->height;
        focal.aspect = (c->flags & CCAL_FIX_ASPECT_RATIO) ? c->aspect : 0.0;
        c->foc_buf ^= 1;
        c->focal_pending = false;
    }
    HIPCHK(launch_init_extrinsics(c->stream, c->d_obj, c->d_img, c->d_intr0, use_H ? c->d_H : nullptr, c->init_hmode, init_spec(c),
                                  c->d_ext0, c->d_fail, c->N, c->P, fused_focal ? &focal : nullptr));
    int rc;
    if (!c->defer_sync && (rc = finish_init(c, false, true)) != CCAL_OK) return rc;
    c->stats.t_init_extr_s = now_s() - t0;
    c->have_ext = true;
    c->have_result = false;
    return CCAL_OK;
}

// ---- A6-A9: global LM ------------------------------------------------------------------
// (trials << 32) | done, written to pinned host memory by workgroup 0 of every k_lm_trial launch that advances the state
static unsigned long long *progress_word(ccal_ctx *c) { return reinterpret_cast<unsigned long long *>(c->h_flags + 32); }

// Wait until the device reports `done` or `target_trials` executed trials.  Spinning on the
// pinned progress word avoids the wake-up latency of hipStreamSynchronize (tens of microseconds,
// comparable to a whole small solve); after 200 us of spinning the solve is long enough for that
// latency not to matter and the wait becomes a blocking stream synchronisation.
static int wait_progress(ccal_ctx *c, long target_trials) {
    volatile unsigned long long *p = progress_word(c);
    if (c->spin_wait) {
        const double t_end = now_s() + 200e-6;
        for (long spins = 0;; spins++) {
            const unsigned long long v = *p;
            if ((v & 1ull) || (long)(v >> 32) >= target_trials) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return CCAL_OK;
            }
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 255) == 255 && now_s() > t_end) break;
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

static TrialLaunch base_launch(ccal_ctx *c) {
    TrialLaunch t;
    t.obj = c->d_obj; t.img = c->d_img; t.blk = c->d_blk; t.st = c->d_st; t.acc = c->d_acc; t.R = c->acc_R;
    t.trace = c->d_trace; t.host_result = c->h_st; t.host_progress = progress_word(c);
    t.intr0 = c->d_intr0; t.ext0 = c->d_ext0; t.ext0_stride = EXT; t.init = make_init(c);
    t.N = c->N; t.P = c->P; t.wpi = c->wpi;
    // the first launch of every solve reports the initialisation's flags and start intrinsics to the host (finish_init_check)
    t.host_poses = c->host_poses;
    t.init_fail = c->d_fail; t.host_init_intr = reinterpret_cast<double *>(c->h_flags + 4); t.host_init_fail = c->h_flags;
    return t;
}

// the exchange is fused into the trial launch unless the ranks of this fit run many workgroups each (every workgroup polls)
static bool xchg_fused(const ccal_ctx *c) {
    if (c->p2p_no_fuse) return false;  // other processes' ranks share this GPU: a waiting rank must hold one CU, not a grid
    const int64_t ng = c->n_global > 0 ? c->n_global : c->N;
    const int64_t per_rank = (ng + c->world - 1) / (c->world > 0 ? c->world : 1);
    return lm_sweep_blocks((int)per_rank) <= FUSED_XCHG_MAX_BLOCKS;
}

// Launch trial k (k_lm_trial): its prologue consumes the exact accumulators of trial k - 1 -- directly
// (single GPU, or the fused xGMI exchange), or as one row of doubles that was summed across GPUs
// between the two launches (k_lm_row -> RCCL / host hook, or k_lm_exchange).
static int enqueue_trial(ccal_ctx *c, hipStream_t stream, int k, bool final_only) {
    TrialLaunch t = base_launch(c);
    t.k = k;
    t.first = k == 0;
    t.final_only = final_only ? 1 : 0;
    if (via_p2p(c)) t.p2p = &c->p2p->dev();
    if (k > 0) {
        if (via_p2p(c)) {
            if (c->p2p_split) {
                // this shard's row was sent by enqueue_send(k), like every other shard's of this stream: collect them
                HIPCHK(launch_lm_recv(stream, c->d_st, c->d_red, c->p2p->dev(), k));
                t.row_in = c->d_red;
            } else if (xchg_fused(c)) {
                t.xchg_fused = 1;      // the prologue decodes the local row and exchanges it itself
            } else {
                // one launch: local row, exchange with the peers over xGMI, sum in rank order
                HIPCHK(launch_lm_exchange(stream, c->d_acc, c->acc_R, c->d_st, c->d_red, c->p2p->dev(), k));
                t.row_in = c->d_red;
            }
        } else if (sharded(c)) {
            // local row (one workgroup) -> 68 doubles summed across GPUs on the launch stream
            HIPCHK(launch_lm_row(stream, c->d_acc, c->acc_R, c->d_st, k, c->d_red));
            const int rc = run_allreduce(c, c->d_red, NRED);
            if (rc) return rc;
            t.row_in = c->d_red;
        }
    }
    if (!final_only && c->time_sweeps && c->ev_used < MAX_KINDS) {
        t.ev_start = c->ev[2 * c->ev_used];
        t.ev_stop = c->ev[2 * c->ev_used + 1];
        c->ev_used++;
    } else if (final_only && c->time_sweeps) {      // (the last terminating launch of a solve is the one that counts; prologue flows only)
        t.ev_start = c->ev[2 * MAX_KINDS];
        t.ev_stop = c->ev[2 * MAX_KINDS + 1];
        c->ev_final_used = true;
    }
    HIPCHK(launch_lm_trial(stream, t));
    return CCAL_OK;
}

// p2p_split only: the first half of exchange k (every shard of a stream sends before any of them receives)
static int enqueue_send(ccal_ctx *c, hipStream_t stream, int k) {
    if (k > 0 && via_p2p(c) && c->p2p_split) HIPCHK(launch_lm_send(stream, c->d_acc, c->acc_R, c->d_st, c->p2p->dev(), k));
    return CCAL_OK;
}

// Capture trials 0 .. lm_batch-1 plus the terminating prologue-only launch into one graph.
static int capture_graph(ccal_ctx *c, hipGraphExec_t *out) {
    hipGraph_t graph = nullptr;
    hipStream_t cs = c->own_stream;
    HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    int rc = CCAL_OK;
    for (int k = 0; k < c->lm_batch && rc == CCAL_OK; k++) rc = enqueue_trial(c, cs, k, false);
    if (rc == CCAL_OK) rc = enqueue_trial(c, cs, c->lm_batch, true);
    const hipError_t e2 = hipStreamEndCapture(cs, &graph);
    if (rc != CCAL_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    HIPCHK(e2);
    if (*out) { (void)hipGraphExecDestroy(*out); *out = nullptr; }
    HIPCHK(hipGraphInstantiate(out, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    return CCAL_OK;
}

// A solve in pieces, so that one host thread can drive several devices (ccal_calibrate_multi): their
// kernels wait for each other's rows, hence every device's batch must be enqueued before any is waited for.
//   lm_begin   first batch (one graph launch, or trial by trial) + terminating prologue-only launch
//   lm_wait    until the batch has been consumed or the solve has ended
//   lm_more    the graph was sized for another problem: another batch
//   lm_end     books, errors, size of the next graph
static int lm_prepare(ccal_ctx *c) {
    if (!c) return fail(CCAL_E_ARG, "ccal_lm_run: ctx == NULL");
    if (!c->have_problem || !c->have_img || !c->have_intr || !c->have_ext)
        return fail(CCAL_E_STATE, "ccal_lm_run: initial intrinsics / extrinsics / image points not set");
    REFUSE_PENDING_FOCAL(c, "ccal_lm_run");
    HIPCHK(hipSetDevice(c->device));
    if (c->wpi_sharded != sharded(c)) resolve_wpi(c);   // a transport was attached or dropped since the problem was set
    c->run_t0 = now_s();
    // upper bound on trials: every accepted iteration may be preceded by lambda climbing
    // from -16 to 16 (rejections + re-damping sweeps)
    c->run_max_trials = 8L + (long)c->max_iter * 70L;
    if (via_p2p(c) && c->run_max_trials > 65000L) c->run_max_trials = 65000L;   // the exchange tag carries the trial index in 16 bits
    c->run_trials = 0;
    c->ev_used = 0;
    c->ev_final_used = false;
    c->h_st->done = 0;
    *progress_word(c) = 0ull;
    if (c->acc_dirty) {
        // first solve of a problem, or the last one did not end with its terminating launch (which leaves
        // the accumulators cleared): start from zeros
        HIPCHK(hipMemsetAsync(c->d_acc, 0, ACC_BYTES, c->stream));
    }
    c->acc_dirty = true;               // until this solve is seen to have ended with its terminating launch
    c->run_graphed = false;
    c->run_merged = c->recording;
    return CCAL_OK;
}

static int lm_begin(ccal_ctx *c) {
    int rc0 = lm_prepare(c);
    if (rc0) return rc0;
    if (via_p2p(c) && c->p2p_split)
        return fail(CCAL_E_STATE, "ccal_lm_run: this context shares its stream with other shards (ccal_calibrate_multi drives it)");
    // A few images on one GPU: every trial in ONE launch of one workgroup (k_lm_trial<.., SOLO>): the trial-to-trial hand-off is a
    // workgroup barrier instead of a kernel boundary + accumulator round trip; same bits as the launch-per-trial form
    c->run_solo = c->use_solo && !sharded(c) && !c->time_sweeps && c->N <= lm_solo_max_images();
    if (!c->recording) c->join_needed = true;      // (launches of its own on this stream: see fit_begin_merged)
    if (c->run_solo) {
        const double t_enq0 = now_s();
        TrialLaunch t = base_launch(c);
        t.k = 0; t.first = 1; t.solo = 1;
        t.max_trials = (int)c->run_max_trials;
        HIPCHK(launch_lm_trial(c->stream, t));
        c->run_trials = c->run_max_trials;          // (the launch reports once: done, or budget exhausted)
        c->dbg_enqueue_s += now_s() - t_enq0;
        return CCAL_OK;
    }
    c->run_graphed = c->use_graph && !c->recording && !c->time_sweeps && (!sharded(c) || via_p2p(c));
    if (c->run_graphed) {
        ccal_ctx::GraphKey key;
        memset(&key, 0, sizeof key);
        key.obj = c->d_obj; key.img = c->d_img; key.blk = c->d_blk; key.st = c->d_st; key.acc = c->d_acc; key.R = c->acc_R;
        key.red = c->d_red; key.trace = c->d_trace; key.intr0 = c->d_intr0; key.ext0 = c->d_ext0; key.host_poses = c->host_poses;
        key.N = c->N; key.P = c->P; key.wpi = c->wpi; key.batch = c->lm_batch; key.init = make_init(c);
        if (c->graph_dirty || !c->g_first || memcmp(&key, &c->graph_key, sizeof key) != 0) {
            const int rc = capture_graph(c, &c->g_first);
            if (rc) {
                // a stream capture can be invalidated by what other host threads do meanwhile (allocations, synchronous copies):
                // this solve is launched kernel by kernel -- same kernels, same bits -- and the next one captures again
                (void)hipGetLastError();
                if (c->g_first) { (void)hipGraphExecDestroy(c->g_first); c->g_first = nullptr; }
                c->run_graphed = false;
                c->graph_dirty = true;
            } else {
                c->graph_key = key;
                c->graph_dirty = false;
            }
        }
    }
    const double t_enq0 = now_s();
    if (c->run_graphed) {
        HIPCHK(hipGraphLaunch(c->g_first, c->stream));
        c->run_trials = c->lm_batch;
    } else {
        for (int k = 0; k < c->lm_batch; k++) {
            const int rc = enqueue_trial(c, c->stream, k, false);
            if (rc) return rc;
        }
        c->run_trials = c->lm_batch;
        const int rc = enqueue_trial(c, c->stream, (int)c->run_trials, true);
        if (rc) return rc;
    }
    c->dbg_enqueue_s += now_s() - t_enq0;
    return CCAL_OK;
}

static int lm_wait(ccal_ctx *c) { return wait_progress(c, c->run_trials); }

static bool lm_unfinished(const ccal_ctx *c) { return !c->h_st->done && c->run_trials < c->run_max_trials; }

static int lm_more(ccal_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    c->join_needed = true;
    int rc;
    const int more = c->lm_batch < 4 ? 4 : c->lm_batch;
    for (int k = 0; k < more; k++)
        if ((rc = enqueue_trial(c, c->stream, (int)c->run_trials + k, false)) != CCAL_OK) return rc;
    c->run_trials += more;
    return enqueue_trial(c, c->stream, (int)c->run_trials, true);
}

static int lm_end(ccal_ctx *c) {
    HIPCHK(hipSetDevice(c->device));
    const bool graphed = c->run_graphed;
    const double t0 = c->run_t0;
    if (via_p2p(c)) {
        std::string err;
        if (c->p2p->check_and_clear(err)) return fail(CCAL_E_COMM, "ccal_lm_run: " + err);
    }
    const LmState &r = *c->h_st;
    if (r.done) c->acc_dirty = false;  // the terminating launch cleared every accumulator buffer
    c->stats.t_lm_s = now_s() - t0;
    c->dbg_total_s += c->stats.t_lm_s;
    if (++c->dbg_runs == 200 && getenv("CCAL_DEBUG_TIMING"))
        fprintf(stderr, "[ccal] lm_run x200: enqueue %.2f us, total %.2f us per call (graph %d, batch %d)\n",
                1e6 * c->dbg_enqueue_s / 200, 1e6 * c->dbg_total_s / 200, (int)graphed, c->lm_batch);
    c->stats.iters = r.iters; c->stats.sweeps = r.sweeps; c->stats.reschur = r.reschur;
    c->stats.rejected = r.rejected; c->stats.lambda_lg10 = r.lambda_lg10; c->stats.n_bad = r.n_bad;
    c->stats.final_err = r.prev_err; c->stats.rel_step = r.rel_step;
    c->stats.rms = std::sqrt(r.prev_err / ((double)r.N_global * (double)c->P));
    c->stats.sweep_form = lm_sweep_form(c->N, c->wpi, via_p2p(c), c->run_solo) | (c->run_merged ? 1 << 16 : 0);
    c->stats.sweep_us = 0.0;
    c->stats.timed_sweeps = 0;
    c->stats.solve_kernels_us = 0.0;
    if (!r.done) return fail(CCAL_E_STATE, "ccal_lm_run: trial budget exhausted without termination");
    if (c->time_sweeps) {
        HIPCHK(hipStreamSynchronize(c->stream));
        double sum_ms = 0.0, all_ms = 0.0;
        int n = 0;
        for (int t = 0; t < c->ev_used; t++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * t], c->ev[2 * t + 1]));
            all_ms += ms;
            if (r.kinds[t] != 1 + MODE_INIT && r.kinds[t] != 1 + MODE_EVAL) continue;   // re-damping / surplus launches
            sum_ms += ms;
            n++;
        }
        if (c->ev_final_used) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * MAX_KINDS], c->ev[2 * MAX_KINDS + 1]));
            all_ms += ms;
        }
        c->stats.solve_kernels_us = 1000.0 * all_ms;
        c->stats.timed_sweeps = n;
        c->stats.sweep_us = n ? 1000.0 * sum_ms / n : 0.0;
    }
    c->have_result = true;
    // size the next graph to the trials this problem needed (a surplus launch costs a kernel
    // boundary); a different problem simply continues trial by trial above
    if (c->adaptive_batch && !c->run_solo && r.trials != c->lm_batch) {
        c->lm_batch = r.trials < 1 ? 1 : (r.trials > 48 ? 48 : r.trials);
        c->graph_dirty = true;
    }
    if (r.status == -4) return fail(CCAL_E_NONFINITE, "ccal_lm_run: non-finite reprojection error on the device");
    if (r.status == -3) return fail(CCAL_E_DEGENERATE, "ccal_lm_run: reduced intrinsic system not positive definite");
    return (r.status == 1 ? CCAL_W_MAX_ITER : CCAL_OK) | (r.n_bad > 0 ? CCAL_W_RANK_DEFICIENT : 0);
}

int ccal_lm_run(ccal_ctx *c) {
    int rc = lm_begin(c);
    if (rc) return rc;
    if ((rc = lm_wait(c)) != CCAL_OK) return rc;
    while (lm_unfinished(c)) {   // the graph was sized for another problem: continue (the next solve re-captures)
        if ((rc = lm_more(c)) != CCAL_OK) return rc;
        if ((rc = lm_wait(c)) != CCAL_OK) return rc;
    }
    return lm_end(c);
}

int ccal_set_pose_init(ccal_ctx *c, int32_t mode) {
    if (!c) return fail(CCAL_E_ARG, "ccal_set_pose_init: ctx == NULL");
    if (mode < 0 || mode > 2) return fail(CCAL_E_ARG, "ccal_set_pose_init: mode is 0, 1 or 2");
    c->init_hmode = mode;
    return CCAL_OK;
}

int ccal_set_graph(ccal_ctx *c, int32_t enable) {
    if (!c) return fail(CCAL_E_ARG, "ccal_set_graph: ctx == NULL");
    c->use_graph = enable != 0;
    return CCAL_OK;
}

int ccal_set_sweep_timing(ccal_ctx *c, int32_t enable) {
    if (!c) return fail(CCAL_E_ARG, "ccal_set_sweep_timing: ctx == NULL");
    HIPCHK(hipSetDevice(c->device));
    if (enable && c->ev.empty()) {
        c->ev.resize(2 * MAX_KINDS + 2);
        for (auto &e : c->ev) HIPCHK(hipEventCreate(&e));
    }
    c->time_sweeps = enable != 0;
    return CCAL_OK;
}

// rvec,tvec sit at words BLK_EXT.. of each image's block of the accepted parity: N rows of 8 doubles -> dst (host)
static int enqueue_pose_rows(ccal_ctx *c, double *dst) {
    HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * 8, c->d_blk + (size_t)c->h_st->parity * c->N * BLK + BLK_EXT,
                            sizeof(double) * BLK, sizeof(double) * 8, (size_t)c->N, hipMemcpyDeviceToHost, c->stream));
    return CCAL_OK;
}

static void unpack_pose_rows(const ccal_ctx *c, const double *e, double *rvecs, double *tvecs) {
    for (int i = 0; i < c->N; i++)
        for (int k = 0; k < 3; k++) {
            if (rvecs) rvecs[(size_t)3 * i + k] = e[(size_t)i * 8 + k];
            if (tvecs) tvecs[(size_t)3 * i + k] = e[(size_t)i * 8 + 3 + k];
        }
}

int ccal_get_result(ccal_ctx *c, double intr[CCAL_NINTR], double *rvecs, double *tvecs, ccal_stats *stats) {
    if (!c) return fail(CCAL_E_ARG, "ccal_get_result: ctx == NULL");
    if (!c->have_result) return fail(CCAL_E_STATE, "ccal_get_result: no finished fit");
    HIPCHK(hipSetDevice(c->device));
    const LmState &r = *c->h_st;
    if (intr) memcpy(intr, r.intr_prev, sizeof(double) * 9);
    if (rvecs || tvecs) {
        std::vector<double> e((size_t)c->N * 8);
        const int rc = enqueue_pose_rows(c, e.data());
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(c->stream));
        unpack_pose_rows(c, e.data(), rvecs, tvecs);
    }
    if (stats) *stats = c->stats;
    return CCAL_OK;
}

int ccal_get_trace(ccal_ctx *c, ccal_trace *out, int32_t max_entries, int32_t *n_entries) {
    if (!c || !n_entries) return fail(CCAL_E_ARG, "ccal_get_trace: NULL argument");
    if (!c->have_result) return fail(CCAL_E_STATE, "ccal_get_trace: no finished fit");
    HIPCHK(hipSetDevice(c->device));
    int n = c->h_st->n_trace < c->max_trace ? c->h_st->n_trace : c->max_trace;
    if (n > max_entries) n = max_entries;
    *n_entries = n;
    if (out && n > 0) {
        static_assert(sizeof(ccal_trace) == sizeof(TraceRec), "trace record layouts differ");
        HIPCHK(hipMemcpyAsync(out, c->d_trace, sizeof(TraceRec) * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return CCAL_OK;
}

int ccal_get_init(ccal_ctx *c, double intr[CCAL_NINTR], double *rt_host) {
    if (!c) return fail(CCAL_E_ARG, "ccal_get_init: ctx == NULL");
    HIPCHK(hipSetDevice(c->device));
    if (intr) {
        if (!c->have_intr) return fail(CCAL_E_STATE, "ccal_get_init: no initial intrinsics");
        REFUSE_PENDING_FOCAL(c, "ccal_get_init");
        memcpy(intr, c->intr0, sizeof c->intr0);
    }
    if (rt_host) {
        if (!c->have_ext) return fail(CCAL_E_STATE, "ccal_get_init: no initial extrinsics");
        std::vector<double> e((size_t)c->N * EXT);
        HIPCHK(hipMemcpyAsync(e.data(), c->d_ext0, sizeof(double) * e.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int i = 0; i < c->N; i++)
            for (int k = 0; k < 6; k++) rt_host[(size_t)6 * i + k] = e[(size_t)i * EXT + k];
    }
    return CCAL_OK;
}

// ---- one-shot drop-in ------------------------------------------------------------------
// One context per (host thread, device) is kept alive between ccal_calibrate calls: creating the
// stream, ~15 device buffers and the launch graph costs far more than a small fit.
static thread_local ccal_ctx *t_cached[16] = {nullptr};
static thread_local std::vector<ccal_ctx *> t_pool[16];   // ccal_calibrate_many's contexts, per device
static thread_local CopySeg *t_segs = nullptr;            // ... and its table of staged-input copies (pinned, device-visible)
static thread_local size_t t_segs_cap = 0;
struct MergeArena {
    int device = -1;
    char *d = nullptr, *h = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
};
static thread_local MergeArena t_arena[16][2];   // per device: tables of the merged initialisation launches | of the merged solves

// ... and one set of contexts per (host thread, device list) for ccal_calibrate_multi
struct MultiSet {
    std::vector<int> devices;
    std::vector<ccal_ctx *> ctx;
    bool shared = false;               // some device carries more than one shard (one stream per device, split exchange)
};
static thread_local std::vector<MultiSet> t_multi;

static void multi_destroy(MultiSet &m) {
    // every inbox stays mapped until no context of the set can still be sending
    for (ccal_ctx *c : m.ctx)
        if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    for (ccal_ctx *c : m.ctx)
        if (c) c->stream = c->own_stream;          // (shards of one device shared the first one's stream)
    for (ccal_ctx *c : m.ctx)
        if (c) ccal_destroy(c);
    m.ctx.clear();
}

void ccal_thread_cleanup(void) {
    for (auto &c : t_cached) { if (c) ccal_destroy(c); c = nullptr; }
    for (auto &pool : t_pool) { for (auto c : pool) ccal_destroy(c); pool.clear(); }
    if (t_segs) { (void)hipHostFree(t_segs); t_segs = nullptr; t_segs_cap = 0; }
    for (auto &dev_arenas : t_arena)
        for (auto &A : dev_arenas)
            if (A.d) {
                (void)hipSetDevice(A.device);
                (void)hipDeviceSynchronize();
                (void)hipFree(A.d); (void)hipHostFree(A.h);
                if (A.ev) (void)hipEventDestroy(A.ev);
                A = MergeArena();
            }
    for (auto &m : t_multi) multi_destroy(m);
    t_multi.clear();
}

}  // extern "C"

// The cached contexts of a host thread die with the thread (a Julia task pool or a Python worker thread that
// comes and goes would otherwise leak one stream, ~25 device buffers and a launch graph per thread).  A
// thread_local object's destructor runs at thread exit -- for the main thread before any atexit handler, i.e.
// while the HIP runtime is still up.
namespace {
struct ThreadCacheGuard {
    ~ThreadCacheGuard() { ccal_thread_cleanup(); }
};
thread_local ThreadCacheGuard t_cache_guard;
}  // namespace

extern "C" {

static int stage_need(ccal_ctx *c, size_t bytes);
static int signal_and_spin(ccal_ctx *c);

int ccal_calibrate(const float *obj, const float *img, int32_t P, int32_t N, int32_t width, int32_t height,
                   double aspect, uint32_t flags, int32_t max_iter, double eps, int32_t device, double K4[4],
                   double dist5[5], double *rvecs, double *tvecs, double *rms, int32_t *iters, ccal_stats *stats) {
    if (!obj || !img) return fail(CCAL_E_ARG, "ccal_calibrate: NULL points");
    if (device < 0 || device >= 16) return fail(CCAL_E_ARG, "ccal_calibrate: bad device index");
    ccal_ctx *c = t_cached[device];
    int rc;
    if (!c) {
        if ((rc = ccal_create(device, &c)) < 0) return rc;
        t_cached[device] = c;
        (void)&t_cache_guard;          // instantiate this thread's guard: the cache is released at thread exit
    }
    double intr[9];
    int lm_rc = 0;
    const bool guess = (flags & CCAL_USE_INTRINSIC_GUESS) != 0;
    c->defer_sync = true;          // the whole pipeline is enqueued, then checked once
    // A small call (the reference's own regime: a handful of images) stages its inputs in the context's pinned block, from where a
    // kernel copies them (no copy commands from pageable memory), and has the launch that ends the solve write the poses back
    // into the same block (no read-back launch, no second wait).
    c->small_call = P >= 1 && N >= 1 && sizeof(float) * 3 * (size_t)P <= PIN_OBJ_BYTES && sizeof(float) * 2 * (size_t)P * N <= PIN_IMG_BYTES &&
                    sizeof(double) * 8 * (size_t)N <= PIN_POSE_BYTES && !getenv("CCAL_NO_SMALL_CALL");
    c->host_poses = (c->small_call && (rvecs || tvecs)) ? reinterpret_cast<double *>(static_cast<char *>(c->h_pin) + PIN_POSE_OFF) : nullptr;
    do {
        if ((rc = ccal_set_problem(c, obj, P, N, width, height, aspect, flags, max_iter, eps)) < 0) break;
        if ((rc = ccal_set_image_points(c, img)) < 0) { c->have_problem = false; c->obj_staged = 0; break; }   // (the template was not uploaded)
        if (guess) {
            if (!K4 || !dist5) { rc = fail(CCAL_E_ARG, "ccal_calibrate: intrinsic guess requested but K4/dist5 NULL"); break; }
            const double g[9] = {K4[0], K4[1], K4[2], K4[3], dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]};
            if ((rc = ccal_set_intrinsics(c, g)) < 0) break;
        } else if ((rc = ccal_init_intrinsics(c)) < 0) break;
        if ((rc = ccal_init_extrinsics(c)) < 0) break;
        lm_rc = ccal_lm_run(c);
        const std::string lm_err = g_err;
        // One read-back for everything the host still needs -- the initialisation's flags and values, the poses --
        // through pinned memory, and one wait on a completion word instead of two stream synchronisations.
        // (the initialisation's flags and values came to the host with the solve's first launch)
        const bool poses = lm_rc >= 0 && c->have_result && (rvecs || tvecs);
        const double *pose_rows = nullptr;
        if (poses && c->host_poses != nullptr) {
            pose_rows = c->host_poses;         // written by the launch that found the solve done, ahead of its progress word
        } else if (poses) {
            if ((rc = stage_need(c, sizeof(double) * 8 * (size_t)c->N)) < 0) break;
            if ((rc = enqueue_pose_rows(c, static_cast<double *>(c->h_stage))) < 0) break;
            if ((rc = signal_and_spin(c)) < 0) break;
            pose_rows = static_cast<const double *>(c->h_stage);
        } else if (lm_rc < 0) {
            // the LM may not have got as far as its first launch, let alone waiting for its completion: fetch the flags the old way
            (void)finish_init_enqueue(c);
            (void)hipStreamSynchronize(c->stream);
        }
        // initialisation failures explain (and outrank) whatever the LM reported on garbage
        if ((rc = finish_init_check(c, !guess, true)) < 0) break;
        if ((rc = lm_rc) < 0) { g_err = lm_err; break; }
        ccal_stats st;
        if ((rc = ccal_get_result(c, intr, nullptr, nullptr, &st)) < 0) break;
        if (poses) unpack_pose_rows(c, pose_rows, rvecs, tvecs);
        if (K4) memcpy(K4, intr, sizeof(double) * 4);
        if (dist5) memcpy(dist5, intr + 4, sizeof(double) * 5);
        if (rms) *rms = st.rms;
        if (iters) *iters = st.iters;
        if (stats) *stats = st;
        rc = lm_rc;
    } while (0);
    c->defer_sync = false;
    c->small_call = false;
    c->host_poses = nullptr;
    if (rc < 0) {                  // leave no half-finished work behind on the cached stream
        const std::string keep = g_err;
        (void)hipStreamSynchronize(c->stream);
        g_err = keep;
    }
    return rc;
}

// Initialisation + LM on the problem and image points already set on this context (host copy or device pointer), as
// one pipeline: every stage is enqueued behind the previous one, the host waits once (the LM's progress word); the
// initialisation's flags and values are read back in front of the LM's launches, so they have landed by then.
// ccal_fit in two halves: everything enqueued (both initialisation stages and the LM's launches) | the one wait, the LM's
// continuation if its graph was sized for another problem, the checks.  ccal_fit_many runs the first half of every context
// before the second half of any.
struct FitPhase { bool guess = false; int rc = CCAL_OK; int lm_rc = CCAL_OK; bool lm_begun = false; std::string err; };

// (the enqueueing half itself in two steps -- the initialisation stages | the solve's launches --, so that merged launches can put the
//  first step of every context on the device before the second step of any is even recorded)
static void fit_begin_init(ccal_ctx *c, FitPhase &ph) {
    ph = FitPhase();
    if (!c) { ph.rc = fail(CCAL_E_ARG, "ccal_fit: ctx == NULL"); ph.err = g_err; return; }
    if (!c->have_problem || !c->have_img) { ph.rc = fail(CCAL_E_STATE, "ccal_fit: problem / image points not set"); ph.err = g_err; return; }
    ph.guess = (c->flags & CCAL_USE_INTRINSIC_GUESS) != 0;
    if (ph.guess && !c->have_intr) { ph.rc = fail(CCAL_E_STATE, "ccal_fit: USE_INTRINSIC_GUESS without ccal_set_intrinsics"); ph.err = g_err; return; }
    c->defer_sync = true;
    if (ph.guess || (ph.rc = ccal_init_intrinsics(c)) >= 0) ph.rc = ccal_init_extrinsics(c);
    c->defer_sync = false;
    if (ph.rc < 0) ph.err = g_err;
}

static void fit_begin_lm(ccal_ctx *c, FitPhase &ph) {
    if (!c || ph.rc < 0) return;
    ph.lm_rc = lm_begin(c);            // (its first launch carries the init stages' flags and intrinsics to the host)
    ph.lm_begun = ph.lm_rc == CCAL_OK;
    if (ph.lm_rc < 0) ph.err = g_err;
}

static void fit_begin(ccal_ctx *c, FitPhase &ph) {
    fit_begin_init(c, ph);
    fit_begin_lm(c, ph);
}

static int fit_end(ccal_ctx *c, FitPhase &ph) {
    if (!c) { g_err = ph.err; return ph.rc; }
    int rc = ph.rc;
    if (rc >= 0) {
        int lm_rc = ph.lm_rc;
        if (ph.lm_begun) {
            (void)hipSetDevice(c->device);
            lm_rc = lm_wait(c);
            while (lm_rc == CCAL_OK && lm_unfinished(c))   // the graph was sized for another problem: continue (the next solve re-captures)
                if ((lm_rc = lm_more(c)) == CCAL_OK) lm_rc = lm_wait(c);
            if (lm_rc == CCAL_OK) lm_rc = lm_end(c);
            if (lm_rc < 0) ph.err = g_err;
        }
        if (lm_rc < 0) {               // (the LM may not have reached its first launch, and did not wait for its own completion)
            (void)finish_init_enqueue(c);
            (void)hipStreamSynchronize(c->stream);
        }
        // initialisation failures explain (and outrank) whatever the LM reported on garbage
        if ((rc = finish_init_check(c, !ph.guess, true)) >= 0 && (rc = lm_rc) < 0) g_err = ph.err;
    } else {
        g_err = ph.err;
    }
    if (rc < 0) {
        const std::string keep = g_err;
        (void)hipStreamSynchronize(c->stream);
        g_err = keep;
    }
    return rc;
}

int ccal_fit(ccal_ctx *c) {
    FitPhase ph;
    fit_begin(c, ph);
    return fit_end(c, ph);
}

// ---- ccal_fit_many: merged launches ---------------------------------------------------------------------------------------------
// Contexts in their steady state (a fit before, nothing pending, accumulators known clean -- so that a pipeline is exactly: the
// homography kernel, the pose kernel, the solve's trial launches) whose solve needs at most one workgroup per compute unit do not
// get launches of their own: their enqueueing half runs with a recorder installed (ccal_kernels.hpp), and the records of all of
// them are merged position by position into ONE launch per kernel instantiation, whose workgroups look their own fit's arguments
// up in a device table.  Eight launches then carry 32 six-image fits instead of 256, and the chip runs them side by side; the
// host-side bookkeeping, the per-context progress words and results are exactly those of ccal_fit.
static bool merge_ok(const ccal_ctx *c) {
    if (!c || !c->use_merge || c->time_sweeps || sharded(c) || c->p2p_split || !focal_exact(c) || c->init_hmode == 0 || !c->planar_z0 ||
        (c->flags & CCAL_USE_INTRINSIC_GUESS) != 0 || c->foc_dirty || c->focal_pending || c->acc_dirty || !c->have_problem || !c->have_img)
        return false;
    if ((size_t)32 * c->P + sizeof(double) * 2200 > 60 * 1024) return false;               // (the pose kernel's staged-correspondence form)
    // (a fit of more workgroups fills half the chip alone: merged with others every launch boundary waits for the slowest of them, and
    //  streams of their own overlap better -- 1000 x 54 measured 67 us per fit on streams against 96 merged, six at a time)
    return ((long)c->N * c->wpi + 3) / 4 <= 128;
}

static int arena_need(MergeArena &A, int device, size_t bytes) {
    if (A.device == device && bytes <= A.cap) return CCAL_OK;
    if (A.d) { (void)hipSetDevice(A.device); (void)hipDeviceSynchronize(); (void)hipFree(A.d); (void)hipHostFree(A.h); A.d = A.h = nullptr; A.cap = 0; }
    HIPCHK(hipSetDevice(device));
    if (!A.ev || A.device != device) {
        if (A.ev) (void)hipEventDestroy(A.ev);
        HIPCHK(hipEventCreateWithFlags(&A.ev, hipEventDisableTiming));
    }
    const size_t cap = bytes + bytes / 2 + 65536;
    (void)&t_cache_guard;                         // (the arena is released with this thread's caches: ccal_thread_cleanup)
    HIPCHK(hipMalloc(&A.d, cap));
    HIPCHK(hipHostMalloc(&A.h, cap, hipHostMallocDefault));
    A.cap = cap;
    A.device = device;
    return CCAL_OK;
}

static void merge_poison(ccal_ctx *c) {           // a recording was abandoned: nothing its bookkeeping says has happened
    c->have_H = c->have_intr = c->have_ext = c->have_result = false;
    c->focal_pending = false;
    c->foc_dirty = true;
    c->acc_dirty = true;
}

// the enqueueing half of every context in `idx` (same device, all merge_ok) as merged launches on the first one's stream, in two
// steps: the initialisation stages of ALL contexts are recorded, merged and launched; while they run the solves are recorded, merged
// and launched behind them.
// false: nothing was launched and every context of `idx` is back in a state the ordinary fit_begin handles.
static bool fit_begin_merged(ccal_ctx *const *ctxs, const std::vector<int> &idx, std::vector<FitPhase> &ph) {
    const int m = (int)idx.size();
    std::vector<MergeRecorder> rec((size_t)m);
    static const bool dbg = getenv("CCAL_DEBUG_TIMING") != nullptr;
    ccal_ctx *c0 = ctxs[idx[0]];
    MergeArena *arena = t_arena[c0->device];      // (a thread's merged calls on another device keep their own tables: no release / re-allocation per call)
    struct Launch { int kernel, key, grid, k, first, final_only; size_t pos, dyn, off_map, off_first, off_nblk; std::vector<int> who; };

    // records [lo, hi) of every context -> tables (one per kernel kind in `kinds`, entry = context) + launches, through arena A
    auto merge_and_launch = [&](MergeArena &A, size_t lo, size_t hi, std::initializer_list<int> kinds, size_t &n_launches) -> hipError_t {
        std::vector<Launch> launches;
        size_t off_tab[3] = {0, 0, 0}, bytes = 0;
        for (int t : kinds) {
            size_t ab = 0;
            for (int e = 0; e < m && ab == 0; e++)
                for (size_t j = lo; j < std::min(hi, rec[e].recs.size()); j++)
                    if (rec[e].recs[j].kernel == t) { ab = rec[e].recs[j].args_bytes; break; }
            off_tab[t] = bytes;
            bytes += ((ab * (size_t)m + 255) / 256) * 256;
        }
        for (size_t j = lo; j < hi; j++)
            for (int e = 0; e < m; e++) {
                if (j >= rec[e].recs.size()) continue;
                const MergeRecord &r = rec[e].recs[j];
                Launch *L = nullptr;
                for (auto &q : launches)
                    if (q.pos == j && q.kernel == r.kernel && q.key == r.key && q.k == r.k && q.first == r.first && q.final_only == r.final_only) { L = &q; break; }
                if (!L) {
                    launches.push_back(Launch{r.kernel, r.key, 0, r.k, r.first, r.final_only, j, 0, 0, 0, 0, {}});
                    L = &launches.back();
                }
                L->who.push_back(e);
                L->grid += r.grid;
                L->dyn = std::max(L->dyn, r.dyn_lds);
            }
        for (auto &L : launches) {
            L.off_map = bytes; bytes += (((size_t)L.grid * 4 + 255) / 256) * 256;
            L.off_first = bytes; bytes += (((size_t)m * 4 + 255) / 256) * 256;
            L.off_nblk = bytes; bytes += (((size_t)m * 4 + 255) / 256) * 256;
        }
        if (arena_need(A, c0->device, bytes) != CCAL_OK) return hipErrorOutOfMemory;
        if (A.ev != nullptr && hipEventSynchronize(A.ev) != hipSuccess) return hipErrorUnknown;     // (the last call's launches have read the old tables)
        for (int t : kinds)
            for (int e = 0; e < m; e++)
                for (size_t j = lo; j < std::min(hi, rec[e].recs.size()); j++)
                    if (rec[e].recs[j].kernel == t) {     // (the first record of that kind: a solve's later launches differ in k / first / final_only only)
                        memcpy(A.h + off_tab[t] + rec[e].recs[j].args_bytes * (size_t)e, rec[e].recs[j].args, rec[e].recs[j].args_bytes);
                        break;
                    }
        for (auto &L : launches) {
            int *map = reinterpret_cast<int *>(A.h + L.off_map), *first = reinterpret_cast<int *>(A.h + L.off_first),
                *nblk = reinterpret_cast<int *>(A.h + L.off_nblk);
            for (int e = 0; e < m; e++) first[e] = nblk[e] = 0;
            int b = 0;
            for (int e : L.who) {
                const int g = rec[e].recs[L.pos].grid;
                first[e] = b; nblk[e] = g;
                for (int q = 0; q < g; q++) map[b + q] = e;
                b += g;
            }
        }
        hipError_t he = hipMemcpyAsync(A.d, A.h, bytes, hipMemcpyHostToDevice, c0->stream);
        for (auto &L : launches) {
            if (he != hipSuccess) break;
            const int *map = reinterpret_cast<const int *>(A.d + L.off_map), *first = reinterpret_cast<const int *>(A.d + L.off_first),
                      *nblk = reinterpret_cast<const int *>(A.d + L.off_nblk);
            if (L.kernel == MERGE_HOMOGRAPHY) he = launch_init_homography_multi(c0->stream, L.key, A.d + off_tab[MERGE_HOMOGRAPHY], map, first, L.grid);
            else if (L.kernel == MERGE_EXTRINSICS) he = launch_init_extrinsics_multi(c0->stream, L.key, A.d + off_tab[MERGE_EXTRINSICS], map, first, L.grid, L.dyn);
            else he = launch_lm_trial_multi(c0->stream, L.key, A.d + off_tab[MERGE_LM], map, first, nblk, L.grid, L.k, L.first, L.final_only);
        }
        if (he == hipSuccess) he = hipEventRecord(A.ev, c0->stream);
        n_launches += launches.size();
        return he;
    };
    auto fail_all = [&](hipError_t he) {
        (void)hipStreamSynchronize(c0->stream);
        for (int e = 0; e < m; e++) {
            merge_poison(ctxs[idx[e]]);
            ph[idx[e]] = FitPhase();
            ph[idx[e]].rc = fail(CCAL_E_DEVICE, std::string("ccal_fit_many: merged launch failed: ") + hipGetErrorString(he));
            ph[idx[e]].err = g_err;
        }
    };

    // ---- step 1: the two initialisation stages of every context
    const double t0 = now_s();
    bool ok = hipSetDevice(c0->device) == hipSuccess;
    for (int e = 0; e < m; e++) {
        ccal_ctx *c = ctxs[idx[e]];
        merge_recorder() = &rec[e];
        c->recording = true;
        fit_begin_init(c, ph[idx[e]]);
        c->recording = false;
        merge_recorder() = nullptr;
        if (rec[e].unsupported || ph[idx[e]].rc < 0 || rec[e].recs.size() != 2 || rec[e].recs[0].kernel != MERGE_HOMOGRAPHY ||
            rec[e].recs[1].kernel != MERGE_EXTRINSICS)
            ok = false;
    }
    if (!ok) {
        for (int e = 0; e < m; e++) merge_poison(ctxs[idx[e]]);     // nothing was launched: the ordinary path starts every one of them afresh
        return false;
    }
    // The merged launches run on c0's stream only.  Whatever a participant may still have pending on a stream of its OWN -- the copy
    // launch of inputs staged by ccal_calibrate_many when fewer than four calibrations were small, a deferred upload, the last stores of
    // the terminating launch of an earlier ccal_fit -- is ordered in front of them: an event on that stream, waited for by c0's.  The
    // library knows when that can be the case for a stream it owns (join_needed; never in the steady state of ccal_fit_many, whose
    // contexts were last used by merged launches); a caller's stream (ccal_set_stream) is always joined.
    std::vector<hipStream_t> joined;              // (ccal_calibrate_many puts all small calibrations on ONE stream: one join for all of them)
    for (int e = 1; e < m && ok; e++) {
        ccal_ctx *c = ctxs[idx[e]];
        if (c->stream == c0->stream || (!c->join_needed && c->stream == c->own_stream)) continue;
        if (std::find(joined.begin(), joined.end(), c->stream) != joined.end()) continue;
        joined.push_back(c->stream);
        if (c->ev_join == nullptr && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { ok = false; break; }
        if (hipEventRecord(c->ev_join, c->stream) != hipSuccess || hipStreamWaitEvent(c0->stream, c->ev_join, 0) != hipSuccess) ok = false;
    }
    if (!ok) {
        (void)hipGetLastError();
        for (int e = 0; e < m; e++) merge_poison(ctxs[idx[e]]);
        return false;
    }
    for (int e = 0; e < m; e++) ctxs[idx[e]]->join_needed = false;
    size_t n_launches = 0;
    static const bool one_step = getenv("CCAL_MERGE_ONE_STEP") != nullptr;     // (experiments: everything recorded before anything is launched)
    hipError_t he = one_step ? hipSuccess : merge_and_launch(arena[0], 0, 2, {MERGE_HOMOGRAPHY, MERGE_EXTRINSICS}, n_launches);
    // (from here on launches are in flight: a failure is a HIP error and fails the contexts, it cannot fall back)
    if (he != hipSuccess) { fail_all(he); return true; }
    const double t1 = now_s();
    // ---- step 2: the solves, recorded while the initialisation runs
    size_t len = 0;
    for (int e = 0; e < m; e++) {
        ccal_ctx *c = ctxs[idx[e]];
        merge_recorder() = &rec[e];
        c->recording = true;
        fit_begin_lm(c, ph[idx[e]]);
        c->recording = false;
        merge_recorder() = nullptr;
        bool good = !rec[e].unsupported && ph[idx[e]].lm_begun && rec[e].recs.size() >= 3;
        for (size_t j = 2; j < rec[e].recs.size(); j++) good = good && rec[e].recs[j].kernel == MERGE_LM;
        if (!good) {
            // (cannot happen for a context merge_ok has passed; if it does, the context fails alone: its solve is not launched)
            rec[e].recs.resize(2);
            merge_poison(c);
            if (ph[idx[e]].rc >= 0 && ph[idx[e]].lm_rc >= 0) { ph[idx[e]].lm_rc = fail(CCAL_E_STATE, "ccal_fit_many: the solve has no merged form"); ph[idx[e]].err = g_err; }
            ph[idx[e]].lm_begun = false;
        }
        len = std::max(len, rec[e].recs.size());
    }
    if (one_step && (he = merge_and_launch(arena[0], 0, 2, {MERGE_HOMOGRAPHY, MERGE_EXTRINSICS}, n_launches)) != hipSuccess) { fail_all(he); return true; }
    if (len > 2) he = merge_and_launch(arena[1], 2, len, {MERGE_LM}, n_launches);
    for (int e = 1; e < m && he == hipSuccess; e++)
        if (ctxs[idx[e]]->stream != c0->stream) he = hipStreamWaitEvent(ctxs[idx[e]]->stream, arena[1].ev ? arena[1].ev : arena[0].ev, 0);
    if (dbg) fprintf(stderr, "[ccal] fit_many merged: %d contexts, %zu launches; initialisation recorded + launched in %.1f us, solves in %.1f us\n", m,
                     n_launches, 1e6 * (t1 - t0), 1e6 * (now_s() - t1));
    if (he != hipSuccess) fail_all(he);
    return true;
}

// (errs != NULL: every failing context's own message, for ccal_calibrate_many)
static int fit_many_impl(ccal_ctx *const *ctxs, int32_t n, int32_t *status_out, std::vector<std::string> *errs) {
    if (n < 0 || (n > 0 && !ctxs)) return fail(CCAL_E_ARG, "ccal_fit_many: bad arguments");
    for (int a = 0; a < n; a++)
        for (int b = 0; b < a; b++)
            if (ctxs[a] && ctxs[a] == ctxs[b]) return fail(CCAL_E_ARG, "ccal_fit_many: a context is listed twice");
    for (int a = 0; a < n; a++)
        if (ctxs[a] && (sharded(ctxs[a]) || ctxs[a]->p2p_split))
            return fail(CCAL_E_STATE, "ccal_fit_many: a context is a shard of a joint fit (its peers' launches would have to be enqueued too)");
    std::vector<FitPhase> ph((size_t)n);
    const double t0 = now_s();
    // steady-state contexts of one device with small fits: merged launches; the others: their own pipelines on their own streams
    // (enqueued by this one thread: helper threads enqueueing beside it were measured -- 2 and 4 threads, 32 contexts -- and changed
    //  nothing, the runtime serialises launch calls; a pipeline of its own costs the host ~14 us of launch calls, a recorded one 1.5)
    std::vector<char> merged((size_t)n, 0);
    for (int dev = 0; dev < 16 && n > 1; dev++) {
        std::vector<int> idx;
        for (int a = 0; a < n; a++)
            if (ctxs[a] && ctxs[a]->device == dev && merge_ok(ctxs[a])) idx.push_back(a);
        if (idx.size() < 4) continue;                 // (two or three: their own streams are as fast -- tools/many_timing.py)
        if (fit_begin_merged(ctxs, idx, ph))
            for (int a : idx) merged[a] = 1;
    }
    for (int a = 0; a < n; a++)
        if (!merged[a]) fit_begin(ctxs[a], ph[a]);
    static const bool dbg = getenv("CCAL_DEBUG_TIMING") != nullptr;
    static double dbg_enq = 0.0; static long dbg_calls = 0, dbg_n = 0;
    if (dbg) {
        dbg_enq += now_s() - t0; dbg_n += n;
        if (++dbg_calls % 20 == 0) { fprintf(stderr, "[ccal] fit_many: enqueue %.2f us per context over the last 20 calls\n", 1e6 * dbg_enq / (double)dbg_n); dbg_enq = 0.0; dbg_n = 0; }
    }
    int worst = CCAL_OK;
    std::string worst_err;
    const double t_end0 = now_s();
    for (int a = 0; a < n; a++) {
        const int rc = fit_end(ctxs[a], ph[a]);
        if (status_out) status_out[a] = rc;
        if (rc < 0 && errs) (*errs)[(size_t)a] = g_err;
        if (rc < 0 && worst >= 0) { worst = rc; worst_err = "context " + std::to_string(a) + ": " + g_err; }
        else if (rc > 0 && worst >= 0) worst |= rc;    // the positive statuses are warning bits: the call returns their union
    }
    if (dbg && dbg_calls % 20 == 0) fprintf(stderr, "[ccal] fit_many: waits + checks %.1f us for %d contexts\n", 1e6 * (now_s() - t_end0), n);
    if (worst < 0) g_err = worst_err;
    return worst;
}

int ccal_fit_many(ccal_ctx *const *ctxs, int32_t n, int32_t *status_out) { return fit_many_impl(ctxs, n, status_out, nullptr); }

// fit_model for n calibrations of one caller (shared template, image size, flags, criteria), host arrays in and out like
// ccal_calibrate: a pool of contexts of this thread, small inputs staged through each context's pinned block on ONE stream (a copy
// launch each, no waits), the fits through ccal_fit_many (merged launches), the poses written to the pinned blocks by the launches
// that end the solves -- one wait per calibration on a word that is long since set.
int ccal_calibrate_many(const float *obj, int32_t P, int32_t n, const float *const *img, const int32_t *N, int32_t width, int32_t height,
                        double aspect, uint32_t flags, int32_t max_iter, double eps, int32_t device, double *const *out,
                        int32_t *status_out) {
    if (!obj || n < 0 || (n > 0 && (!img || !N || !out))) return fail(CCAL_E_ARG, "ccal_calibrate_many: NULL argument");
    if (device < 0 || device >= 16) return fail(CCAL_E_ARG, "ccal_calibrate_many: bad device index");
    if (flags & CCAL_USE_INTRINSIC_GUESS) return fail(CCAL_E_ARG, "ccal_calibrate_many: no intrinsic guess in the batched form (use ccal_calibrate)");
    auto &pool = t_pool[device];
    while ((int)pool.size() < n) {
        ccal_ctx *c = nullptr;
        const int rc = ccal_create(device, &c);
        if (rc < 0) return rc;
        pool.push_back(c);
        (void)&t_cache_guard;          // (released at thread exit)
    }
    if (n == 0) return CCAL_OK;
    HIPCHK(hipSetDevice(device));
    if ((size_t)n > t_segs_cap) {
        HIPCHK(hipStreamSynchronize(pool[0]->own_stream));      // (the last call's copy launch has read the old table)
        if (t_segs) (void)hipHostFree(t_segs);
        t_segs = nullptr; t_segs_cap = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&t_segs), sizeof(CopySeg) * ((size_t)n + 64), hipHostMallocMapped));
        t_segs_cap = (size_t)n + 64;
    }
    int n_segs = 0;
    hipStream_t shared = pool[0]->own_stream;
    std::vector<int> st((size_t)n, CCAL_OK);
    std::vector<std::string> errs((size_t)n);
    std::vector<ccal_ctx *> live;
    std::vector<int> live_i;
    static const bool no_small = getenv("CCAL_NO_SMALL_CALL") != nullptr;
    auto is_small = [&](int i) {
        return P >= 1 && N[i] >= 1 && sizeof(float) * 3 * (size_t)P <= PIN_OBJ_BYTES && sizeof(float) * 2 * (size_t)P * N[i] <= PIN_IMG_BYTES &&
               sizeof(double) * 8 * (size_t)N[i] <= PIN_POSE_BYTES && !no_small;
    };
    // four or more small calibrations will share merged launches: they are staged on the one stream those run on, and one kernel
    // copies all of them; fewer keep their own streams (their pipelines overlap there) and their own copy launches
    int n_small = 0;
    for (int i = 0; i < n; i++) n_small += (img[i] && out[i] && is_small(i)) ? 1 : 0;
    const bool share = n_small >= 4;
    for (int i = 0; i < n; i++) {
        ccal_ctx *c = pool[i];
        if (!img[i] || !out[i] || N[i] < 1) { st[i] = fail(CCAL_E_ARG, "ccal_calibrate_many: NULL / empty calibration"); errs[i] = g_err; continue; }
        c->small_call = is_small(i);
        c->host_poses = c->small_call ? reinterpret_cast<double *>(static_cast<char *>(c->h_pin) + PIN_POSE_OFF) : nullptr;
        // small inputs: staged and copied on the one stream the merged launches will run on, nothing waited for; larger ones: their own
        // stream, uploaded before this loop goes on (the merged launches would not wait for another stream's copies)
        c->stream = (c->small_call && share) ? shared : c->own_stream;
        c->defer_sync = c->small_call;
        c->defer_copy = (c->small_call && share) ? t_segs + n_segs : nullptr;     // (one copy launch for all of them, below)
        int rc = ccal_set_problem(c, obj, P, N[i], width, height, aspect, flags, max_iter, eps);
        if (rc >= 0 && (rc = ccal_set_image_points(c, img[i])) < 0) { c->have_problem = false; c->obj_staged = 0; }
        c->defer_sync = false;
        if (rc >= 0 && c->defer_copy != nullptr) n_segs++;
        c->defer_copy = nullptr;
        if (rc < 0) { st[i] = rc; errs[i] = g_err; continue; }
        live.push_back(c);
        live_i.push_back(i);
    }
    if (n_segs > 0 && launch_copy_multi(shared, t_segs, n_segs) != hipSuccess) {
        (void)hipStreamSynchronize(shared);
        for (int i = 0; i < n; i++) { pool[i]->small_call = false; pool[i]->host_poses = nullptr; pool[i]->stream = pool[i]->own_stream; pool[i]->have_problem = false; }
        return fail(CCAL_E_DEVICE, "ccal_calibrate_many: copy launch failed");
    }
    std::vector<int32_t> st_live(live.size(), 0);
    std::vector<std::string> errs_live(live.size());
    (void)fit_many_impl(live.data(), (int32_t)live.size(), st_live.data(), &errs_live);
    for (size_t q = 0; q < live.size(); q++) {
        ccal_ctx *c = live[q];
        const int i = live_i[q];
        int rc = st_live[q];
        if (rc < 0) { st[i] = rc; errs[i] = errs_live[q]; continue; }
        double *o = out[i];
        const double *pose_rows = c->host_poses;      // written by the launch that found the solve done, ahead of its progress word
        if (pose_rows == nullptr) {
            int r2;
            if ((r2 = stage_need(c, sizeof(double) * 8 * (size_t)c->N)) < 0 || (r2 = enqueue_pose_rows(c, static_cast<double *>(c->h_stage))) < 0 ||
                (r2 = signal_and_spin(c)) < 0) { st[i] = r2; errs[i] = g_err; continue; }
            pose_rows = static_cast<const double *>(c->h_stage);
        }
        double intr[9];
        ccal_stats s9;
        int r3 = ccal_get_result(c, intr, nullptr, nullptr, &s9);
        if (r3 < 0) { st[i] = r3; errs[i] = g_err; continue; }
        memcpy(o, intr, sizeof(double) * 9);
        o[9] = s9.rms;
        o[10] = (double)s9.iters;
        unpack_pose_rows(c, pose_rows, o + 11, o + 11 + (size_t)3 * c->N);
        st[i] = rc;
    }
    int worst = CCAL_OK;
    for (int i = 0; i < n; i++) {
        ccal_ctx *c = pool[i];
        c->small_call = false;
        c->host_poses = nullptr;
        c->stream = c->own_stream;
        if (status_out) status_out[i] = st[i];
        if (st[i] < 0 && worst >= 0) { worst = st[i]; g_err = "calibration " + std::to_string(i) + ": " + errs[i]; }
        else if (st[i] > 0 && worst >= 0) worst |= st[i];    // (union of the warning bits)
    }
    return worst;
}

static MultiSet *multi_get(int n_dev, const int32_t *dev_ids) {
    std::vector<int> want(dev_ids, dev_ids + n_dev);
    for (auto &m : t_multi)
        if (m.devices == want) return &m;
    MultiSet m;
    m.devices = want;
    m.ctx.assign(n_dev, nullptr);
    bool ok = true;
    // A device listed more than once carries several shards.  Their kernels wait for each other's rows, and a kernel
    // that waits must never sit in front of the kernel that sends in an in-order hardware queue (streams of one
    // process may share one): all shards of a device therefore run on ONE stream, and the exchange is split into a
    // send and a receive kernel, every shard's send enqueued before any shard's receive (k_lm_send / k_lm_recv).
    // Nothing then waits for anything that is queued behind it, whatever the number of shards.
    for (int g = 0; g < n_dev; g++)
        for (int h = 0; h < g; h++) m.shared = m.shared || dev_ids[h] == dev_ids[g];
    for (int g = 0; g < n_dev && ok; g++) {
        ok = ccal_create(dev_ids[g], &m.ctx[g]) == CCAL_OK;
        if (!ok) break;
        m.ctx[g]->p2p_split = m.shared;
        for (int h = 0; h < g; h++)
            if (dev_ids[h] == dev_ids[g]) { m.ctx[g]->stream = m.ctx[h]->stream; break; }
    }
    std::vector<P2P *> boxes(n_dev, nullptr);
    std::string err;
    for (int g = 0; g < n_dev && ok; g++) {
        ccal_ctx *c = m.ctx[g];
        char unused[P2P_HANDLE_BYTES];
        ok = hipSetDevice(c->device) == hipSuccess;
        if (!ok) { fail(CCAL_E_DEVICE, "ccal_calibrate_multi: hipSetDevice failed"); break; }
        c->p2p = new P2P();
        if (c->p2p->create(n_dev, g, unused, err) != 0) { fail(CCAL_E_COMM, "ccal_calibrate_multi: " + err); ok = false; }
        boxes[g] = c->p2p;
    }
    for (int g = 0; g < n_dev && ok; g++) {
        if (m.ctx[g]->p2p->connect_local(boxes.data(), err) != 0) { fail(CCAL_E_COMM, "ccal_calibrate_multi: " + err); ok = false; }
        m.ctx[g]->world = n_dev;
        m.ctx[g]->rank = g;
    }
    if (!ok) {
        const std::string keep = g_err;
        multi_destroy(m);
        g_err = keep;
        return nullptr;
    }
    t_multi.push_back(m);
    (void)&t_cache_guard;
    return &t_multi.back();
}

// shared-device sets: trials k0 .. k0 + count - 1 of EVERY shard plus the terminating prologue-only launch, trial by
// trial -- first every shard's send of exchange k, then every shard's receive + trial k (see multi_get)
static int multi_enqueue(MultiSet *m, int k0, int count) {
    int rc = CCAL_OK;
    for (int k = k0; k <= k0 + count && rc == CCAL_OK; k++) {
        const bool final_only = k == k0 + count;
        for (ccal_ctx *c : m->ctx) {
            HIPCHK(hipSetDevice(c->device));
            if ((rc = enqueue_send(c, c->stream, k)) != CCAL_OK) return rc;
        }
        for (ccal_ctx *c : m->ctx) {
            HIPCHK(hipSetDevice(c->device));
            if ((rc = enqueue_trial(c, c->stream, k, final_only)) != CCAL_OK) return rc;
        }
    }
    for (ccal_ctx *c : m->ctx) c->run_trials = k0 + count;
    return rc;
}

int ccal_calibrate_multi(const float *obj, const float *img, int32_t P, int32_t N, int32_t width, int32_t height,
                         double aspect, uint32_t flags, int32_t max_iter, double eps, int32_t n_dev,
                         const int32_t *dev_ids, double K4[4], double dist5[5], double *rvecs, double *tvecs,
                         double *rms, int32_t *iters, ccal_stats *stats) {
    if (!obj || !img || !dev_ids) return fail(CCAL_E_ARG, "ccal_calibrate_multi: NULL argument");
    if (n_dev < 1 || n_dev > P2P_MAX_RANKS) return fail(CCAL_E_ARG, "ccal_calibrate_multi: 1..16 devices");
    if (N < n_dev) return fail(CCAL_E_ARG, "ccal_calibrate_multi: fewer images than devices");
    for (int g = 0; g < n_dev; g++)
        if (dev_ids[g] < 0 || dev_ids[g] >= 16) return fail(CCAL_E_ARG, "ccal_calibrate_multi: bad device id");
    MultiSet *m = multi_get(n_dev, dev_ids);
    if (!m) return CCAL_E_COMM;
    const bool guess = (flags & CCAL_USE_INTRINSIC_GUESS) != 0;
    if (guess && (!K4 || !dist5)) return fail(CCAL_E_ARG, "ccal_calibrate_multi: intrinsic guess requested but K4/dist5 NULL");
    std::vector<int> lo(n_dev + 1, 0);
    for (int g = 0; g < n_dev; g++) lo[g + 1] = lo[g] + N / n_dev + (g < N % n_dev ? 1 : 0);   // contiguous shards
    int rc = CCAL_OK, lm_rc = CCAL_OK;
    std::string lm_err;
    // every stage is enqueued on ALL devices before anything waits: the devices' kernels wait for each other's rows
    for (ccal_ctx *c : m->ctx) c->defer_sync = true;
    do {
        for (int g = 0; g < n_dev && rc >= 0; g++) {
            ccal_ctx *c = m->ctx[g];
            c->n_global = N;
            if ((rc = ccal_set_problem(c, obj, P, lo[g + 1] - lo[g], width, height, aspect, flags, max_iter, eps)) < 0) break;
            if ((rc = ccal_set_image_points(c, img + (size_t)lo[g] * 2 * P)) < 0) break;
            if (guess) {
                const double gs[9] = {K4[0], K4[1], K4[2], K4[3], dist5[0], dist5[1], dist5[2], dist5[3], dist5[4]};
                rc = ccal_set_intrinsics(c, gs);
            }
        }
        if (rc < 0) break;
        if (!guess) {
            for (int g = 0; g < n_dev && rc >= 0; g++) rc = init_intrinsics_a(m->ctx[g]);
            for (int g = 0; g < n_dev && rc >= 0; g++) rc = init_intrinsics_b(m->ctx[g]);
        }
        for (int g = 0; g < n_dev && rc >= 0; g++) rc = ccal_init_extrinsics(m->ctx[g]);
        if (m->shared) {
            // trial by trial over all shards: sends of exchange k, then (receive, trial k) of every shard
            for (int g = 0; g < n_dev && rc >= 0; g++) rc = lm_prepare(m->ctx[g]);
            if (rc >= 0) rc = multi_enqueue(m, 0, m->ctx[0]->lm_batch);
        } else {
            for (int g = 0; g < n_dev && rc >= 0; g++) rc = lm_begin(m->ctx[g]);
        }
        for (int g = 0; g < n_dev && rc >= 0; g++) rc = lm_wait(m->ctx[g]);
        while (rc >= 0 && lm_unfinished(m->ctx[0])) {   // identical state on every device: one decision for all
            if (m->shared) {
                rc = multi_enqueue(m, (int)m->ctx[0]->run_trials, m->ctx[0]->lm_batch < 4 ? 4 : m->ctx[0]->lm_batch);
            } else {
                for (int g = 0; g < n_dev && rc >= 0; g++) rc = lm_more(m->ctx[g]);
            }
            for (int g = 0; g < n_dev && rc >= 0; g++) rc = lm_wait(m->ctx[g]);
        }
        if (rc < 0) break;
        for (int g = 0; g < n_dev; g++) {
            const int e = lm_end(m->ctx[g]);
            if (g == 0 || (e < 0 && lm_rc >= 0)) { lm_rc = e; lm_err = g_err; }
        }
        // initialisation failures explain (and outrank) whatever the LM reported on garbage
        for (int g = 0; g < n_dev && rc >= 0; g++) rc = finish_init(m->ctx[g], !guess, true);
        if (rc < 0) break;
        if ((rc = lm_rc) < 0) { g_err = lm_err; break; }
        double intr[9];
        ccal_stats st;
        for (int g = 0; g < n_dev && rc >= 0; g++) {
            ccal_stats sg;
            rc = ccal_get_result(m->ctx[g], g == 0 ? intr : nullptr, rvecs ? rvecs + (size_t)3 * lo[g] : nullptr,
                                 tvecs ? tvecs + (size_t)3 * lo[g] : nullptr, &sg);
            if (g == 0) st = sg;
        }
        if (rc < 0) break;
        if (K4) memcpy(K4, intr, sizeof(double) * 4);
        if (dist5) memcpy(dist5, intr + 4, sizeof(double) * 5);
        if (rms) *rms = st.rms;
        if (iters) *iters = st.iters;
        if (stats) *stats = st;
        rc = lm_rc;
    } while (0);
    for (ccal_ctx *c : m->ctx) c->defer_sync = false;
    if (rc < 0) {                  // leave no half-finished work behind on the cached streams
        const std::string keep = g_err;
        for (ccal_ctx *c : m->ctx) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
        for (ccal_ctx *c : m->ctx) { std::string e; if (c->p2p) (void)c->p2p->check_and_clear(e); }
        g_err = keep;
    }
    return rc;
}

// ---- point maps and error metrics (A11-A13) ---------------------------------------------
// Device scratch of these entry points lives in the context and only ever grows: no hipMalloc / hipFree per
// call.  Small calls (the reference's `c(rc, i)` for one point, src/meta.jl:82,88) do not even copy: input,
// poses and output sit in one pinned, device-mapped staging block that the kernel reads and writes over PCIe,
// so such a call is a memcpy, one launch and one stream synchronisation.
static constexpr size_t PIN_BYTES = 64 * 1024 - 64;   // (the last 64 bytes of the staging block hold the completion word)

static int scratch_need(ccal_ctx *c, size_t bytes) {
    if (bytes <= c->scratch_cap) return CCAL_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_cap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIPCHK(hipMalloc(&c->d_scratch, cap));
    c->scratch_cap = cap;
    return CCAL_OK;
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// grow-only pinned host staging for mid-size transfers (caller buffers are pageable: copied through here, the
// DMA itself always runs pinned <-> device); beyond STAGE_MAX the runtime's own pageable path is used
static constexpr size_t STAGE_MAX = 16u << 20;
// Wait for everything enqueued on the context's stream so far: a one-thread kernel behind it raises a sequence number
// in pinned memory and the host spins on that word (a blocking stream synchronisation costs tens to hundreds of
// microseconds of wake-up latency per call); after 2 ms of spinning the wait becomes a stream synchronisation.
static unsigned long long *pin_flag(ccal_ctx *c) { return reinterpret_cast<unsigned long long *>(static_cast<char *>(c->h_pin) + PIN_BYTES); }

// wait for the completion word to take the value `want` (written by k_signal, or by a one-workgroup map kernel itself)
static int spin_flag(ccal_ctx *c, unsigned long long want) {
    volatile unsigned long long *flag = pin_flag(c);
    const double t_end = now_s() + 2e-3;
    bool seen = false;
    for (long spins = 0; !seen; spins++) {
        seen = *flag == want;
        if (!seen && (spins & 255) == 255 && now_s() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

static int signal_and_spin(ccal_ctx *c) {
    volatile unsigned long long *flag = pin_flag(c);
    const unsigned long long want = ++c->pin_seq;
    HIPCHK(launch_signal(c->stream, const_cast<unsigned long long *>(flag), want));
    const double t_end = now_s() + 2e-3;
    bool seen = false;
    for (long spins = 0; !seen; spins++) {
        seen = *flag == want;
        if (!seen && (spins & 255) == 255 && now_s() > t_end) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

static int stage_need(ccal_ctx *c, size_t bytes) {
    if (bytes <= c->stage_cap) return CCAL_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->stage_cap = 0;
    const size_t cap = bytes + bytes / 2 + 4096;
    HIPCHK(hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
    c->stage_cap = cap;
    return CCAL_OK;
}

// in [in_w, M, N] (or one [3, M] template when !per_image) -> out [out_w, M, N], image n through pose n
static int map_many(ccal_ctx *c, const double *in, int32_t M, int32_t N, bool per_image, const double f[2], const double cc[2],
                    double k, const double *rvecs, const double *tvecs, double cs, double *out, bool forward) {
    if (!c || !in || !out || !f || !cc || !rvecs || !tvecs) return fail(CCAL_E_ARG, "ccal map: NULL argument");
    if (M < 0 || N < 1 || !(cs != 0.0)) return fail(CCAL_E_ARG, "ccal map: bad size / checker_size");
    if (M == 0) return CCAL_OK;
    HIPCHK(hipSetDevice(c->device));
    const int in_w = forward ? 3 : 2, out_w = forward ? 2 : 3;
    const size_t n_in = (size_t)in_w * M * (per_image ? N : 1), n_out = (size_t)out_w * M * N, n_rt = (size_t)6 * N;
    const size_t b_rt = align256(8 * n_rt), b_in = align256(8 * n_in), b_out = align256(8 * n_out);
    MapIntr mi;
    mi.f[0] = f[0]; mi.f[1] = f[1]; mi.c[0] = cc[0]; mi.c[1] = cc[1]; mi.k = k; mi.cs = cs;
    const bool small = b_rt + b_in + b_out <= PIN_BYTES;
    char *base;
    if (small) {
        base = static_cast<char *>(c->h_pin);
    } else {
        const int rc = scratch_need(c, b_rt + b_in + b_out);
        if (rc) return rc;
        base = static_cast<char *>(c->d_scratch);
    }
    double *d_rt = reinterpret_cast<double *>(base), *d_in = reinterpret_cast<double *>(base + b_rt),
           *d_out = reinterpret_cast<double *>(base + b_rt + b_in);
    // poses: [3,N] + [3,N] -> [6,N]
    double *h_rt = small ? d_rt : (c->h_rt_stage.resize(n_rt), c->h_rt_stage.data());
    for (int i = 0; i < N; i++)
        for (int a = 0; a < 3; a++) { h_rt[(size_t)6 * i + a] = rvecs[(size_t)3 * i + a]; h_rt[(size_t)6 * i + 3 + a] = tvecs[(size_t)3 * i + a]; }
    const bool staged = !small && b_rt + b_in + b_out <= STAGE_MAX;
    char *hs = nullptr;
    if (staged) {
        const int rc = stage_need(c, b_rt + b_in + b_out);
        if (rc) return rc;
        hs = static_cast<char *>(c->h_stage);
    }
    if (small) {
        memcpy(d_in, in, 8 * n_in);
    } else if (staged) {
        memcpy(hs, h_rt, 8 * n_rt);
        memcpy(hs + b_rt, in, 8 * n_in);
        HIPCHK(hipMemcpyAsync(d_rt, hs, b_rt + 8 * n_in, hipMemcpyHostToDevice, c->stream));   // poses + points: one DMA
    } else {
        HIPCHK(hipMemcpyAsync(d_rt, h_rt, 8 * n_rt, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_in, in, 8 * n_in, hipMemcpyHostToDevice, c->stream));
    }
    // a point-wise call (one workgroup on the pinned block): the map kernel writes the completion word itself -- one launch
    const bool own_flag = small && map_is_one_workgroup(M, N);
    const unsigned long long want = own_flag ? ++c->pin_seq : 0ull;
    if (forward) HIPCHK(launch_map_forward(c->stream, d_in, per_image ? (long long)3 * M : 0ll, d_out, M, N, mi, d_rt, own_flag ? pin_flag(c) : nullptr, want));
    else HIPCHK(launch_map_inverse(c->stream, d_in, d_out, M, N, mi, d_rt, own_flag ? pin_flag(c) : nullptr, want));
    if (!small) {
        HIPCHK(hipMemcpyAsync(staged ? static_cast<void *>(hs + b_rt + b_in) : static_cast<void *>(out), d_out, 8 * n_out,
                              hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (staged) memcpy(out, hs + b_rt + b_in, 8 * n_out);
        return CCAL_OK;
    }
    // zero-copy path: wait for the map kernel without a stream synchronisation
    const int rc = own_flag ? spin_flag(c, want) : signal_and_spin(c);
    if (rc) return rc;
    memcpy(out, d_out, 8 * n_out);
    return CCAL_OK;
}

int ccal_project(ccal_ctx *c, const double *xyz, int32_t M, const double f[2], const double cc[2], double k,
                 const double rvec[3], const double tvec[3], double checker_size, double *rowcol) {
    return map_many(c, xyz, M, 1, true, f, cc, k, rvec, tvec, checker_size, rowcol, true);
}

int ccal_unproject(ccal_ctx *c, const double *rowcol, int32_t M, const double f[2], const double cc[2], double k,
                   const double rvec[3], const double tvec[3], double checker_size, double *xyz) {
    return map_many(c, rowcol, M, 1, true, f, cc, k, rvec, tvec, checker_size, xyz, false);
}

int ccal_project_many(ccal_ctx *c, const double *xyz, int32_t M, int32_t N, int32_t per_image, const double f[2],
                      const double cc[2], double k, const double *rvecs, const double *tvecs, double checker_size,
                      double *rowcol) {
    return map_many(c, xyz, M, N, per_image != 0, f, cc, k, rvecs, tvecs, checker_size, rowcol, true);
}

int ccal_unproject_many(ccal_ctx *c, const double *rowcol, int32_t M, int32_t N, const double f[2], const double cc[2],
                        double k, const double *rvecs, const double *tvecs, double checker_size, double *xyz) {
    return map_many(c, rowcol, M, N, true, f, cc, k, rvecs, tvecs, checker_size, xyz, false);
}

int ccal_errors(ccal_ctx *c, int32_t N, int32_t n1, int32_t n2, const double *obj_scaled, const double *img,
                const double f[2], const double cc[2], double k, const double *rvecs, const double *tvecs,
                double checker_size, int32_t S, const double *inv_pts, double out[4]) {
    if (!c || !obj_scaled || !img || !f || !cc || !rvecs || !tvecs || !out)
        return fail(CCAL_E_ARG, "ccal_errors: NULL argument");
    if (N < 1 || n1 < 2 || n2 < 2 || S < 0 || !(checker_size != 0.0)) return fail(CCAL_E_ARG, "ccal_errors: bad sizes");
    const int P = n1 * n2;
    if ((size_t)P * 24 > 60000) return fail(CCAL_E_ARG, "ccal_errors: board too large for the per-image LDS tile");
    HIPCHK(hipSetDevice(c->device));
    const bool inv = S > 0 && inv_pts;
    const size_t b_obj = align256(8 * (size_t)3 * P), b_img = align256(8 * (size_t)2 * P * N), b_rt = align256(8 * (size_t)6 * N),
                 b_inv = inv ? align256(8 * (size_t)2 * S * N) : 0, b_part = align256(8 * (size_t)4 * (N + 1));
    int rc = scratch_need(c, b_obj + b_img + b_rt + b_inv + b_part);
    if (rc) return rc;
    char *base = static_cast<char *>(c->d_scratch);
    double *d_obj = reinterpret_cast<double *>(base), *d_img = reinterpret_cast<double *>(base + b_obj),
           *d_rt = reinterpret_cast<double *>(base + b_obj + b_img), *d_inv = reinterpret_cast<double *>(base + b_obj + b_img + b_rt),
           *d_part = reinterpret_cast<double *>(base + b_obj + b_img + b_rt + b_inv);
    c->h_rt_stage.resize((size_t)6 * N);
    for (int i = 0; i < N; i++)
        for (int a = 0; a < 3; a++) { c->h_rt_stage[(size_t)6 * i + a] = rvecs[(size_t)3 * i + a]; c->h_rt_stage[(size_t)6 * i + 3 + a] = tvecs[(size_t)3 * i + a]; }
    MapIntr mi;
    mi.f[0] = f[0]; mi.f[1] = f[1]; mi.c[0] = cc[0]; mi.c[1] = cc[1]; mi.k = k; mi.cs = checker_size;
    HIPCHK(hipMemcpyAsync(d_obj, obj_scaled, sizeof(double) * 3 * P, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_img, img, sizeof(double) * 2 * (size_t)P * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_rt, c->h_rt_stage.data(), sizeof(double) * 6 * (size_t)N, hipMemcpyHostToDevice, c->stream));
    if (inv) HIPCHK(hipMemcpyAsync(d_inv, inv_pts, sizeof(double) * 2 * (size_t)S * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_errors(c->stream, d_obj, d_img, mi, d_rt, inv ? d_inv : nullptr, d_part, N, n1, n2, S));
    double *sums = static_cast<double *>(c->h_pin);     // pinned: the copy is a plain DMA, no staging
    HIPCHK(hipMemcpyAsync(sums, d_part + (size_t)4 * N, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // normalisation of src/buildcalibrations.jl:60-65
    const double n = (double)P * N;
    out[0] = std::sqrt(sums[0] / n);
    out[1] = std::sqrt(sums[1] / n);
    out[2] = std::sqrt(sums[2] / ((double)(n1 - 1) * (n2 - 1)) / N);
    out[3] = inv ? std::sqrt(sums[3] / S / N) : 0.0;
    return CCAL_OK;
}

// ---- rectified image -----------------------------------------
int ccal_rectify_image(ccal_ctx *c, const uint8_t *img, int32_t rows, int32_t cols, int32_t channels, int64_t stride_row,
                       int64_t stride_col, int64_t stride_ch, const double f[2], const double cc[2], double k,
                       const double rvec[3], const double tvec[3], double checker_size, double ratio, int32_t a0,
                       int32_t b0, int32_t out_rows, int32_t out_cols, double index_offset, uint8_t *out) {
    if (!c || !img || !out || !f || !cc || !rvec || !tvec) return fail(CCAL_E_ARG, "ccal_rectify_image: NULL argument");
    if (rows < 1 || cols < 1 || channels < 1 || channels > 4 || out_rows < 1 || out_cols < 1 || !(ratio > 0.0) ||
        !(checker_size != 0.0) || stride_row < 1 || stride_col < 1 || stride_ch < 1)
        return fail(CCAL_E_ARG, "ccal_rectify_image: bad sizes / strides");
    HIPCHK(hipSetDevice(c->device));
    const size_t span = (size_t)((rows - 1) * stride_row + (cols - 1) * stride_col + (channels - 1) * stride_ch + 1);
    const size_t out_bytes = (size_t)out_rows * out_cols * channels;
    const int rc = scratch_need(c, align256(span) + align256(out_bytes));
    if (rc) return rc;
    uint8_t *d_in = static_cast<uint8_t *>(c->d_scratch), *d_out = d_in + align256(span);
    const RectPar par = {{f[0], f[1], cc[0], cc[1], k, rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2],
                          checker_size, ratio, index_offset, 0, 0}};
    HIPCHK(hipMemcpyAsync(d_in, img, span, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_rectify(c->stream, d_in, rows, cols, channels, stride_row, stride_col, stride_ch, a0, b0, out_rows, out_cols,
                          par, d_out));
    HIPCHK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

// ---- corner detection ----------------------------------------
int ccal_detect_corners(ccal_ctx *c, const uint8_t *gray, int32_t rows, int32_t cols, int64_t stride_row,
                        int64_t stride_col, int32_t n1, int32_t n2, double *corners, int32_t *found,
                        double *stage_out) {
    if (!c || !gray || !corners || !found) return fail(CCAL_E_ARG, "ccal_detect_corners: NULL argument");
    if (rows < 32 || cols < 32 || n1 < 2 || n2 < 2 || (int64_t)n1 * n2 > Detector::MAX_PEAKS)
        return fail(CCAL_E_ARG, "ccal_detect_corners: bad image or board size");
    HIPCHK(hipSetDevice(c->device));
    if (!c->detector) c->detector = new Detector();
    std::string err;
    int f = 0;
    const int rc = c->detector->detect(c->stream, gray, rows, cols, stride_row, stride_col, n1, n2, corners, &f,
                                       stage_out, err);
    if (rc == -1) return fail(CCAL_E_ARG, "ccal_detect_corners: " + err);
    if (rc < 0) return fail(CCAL_E_DEVICE, err);
    *found = f;
    return CCAL_OK;
}

// The images of one calibration (one size), three in flight: while the device works on image t the host sorts the peaks of image
// t - 1 and launches its refinement, and runs the lattice search of image t - 2 -- each slot has a detector (buffers, pinned block)
// and a stream of its own.  Same arithmetic per image as ccal_detect_corners: same corners.
int ccal_detect_corners_many(ccal_ctx *c, const uint8_t *const *gray, int32_t n, int32_t rows, int32_t cols, int64_t stride_row,
                             int64_t stride_col, int32_t n1, int32_t n2, double *const *corners, int32_t *found) {
    if (!c || n < 0 || (n > 0 && (!gray || !corners || !found))) return fail(CCAL_E_ARG, "ccal_detect_corners_many: NULL argument");
    if (rows < 32 || cols < 32 || n1 < 2 || n2 < 2 || (int64_t)n1 * n2 > Detector::MAX_PEAKS)
        return fail(CCAL_E_ARG, "ccal_detect_corners_many: bad image or board size");
    for (int i = 0; i < n; i++)
        if (!gray[i] || !corners[i]) return fail(CCAL_E_ARG, "ccal_detect_corners_many: NULL image / output");
    HIPCHK(hipSetDevice(c->device));
    constexpr int S = ccal_ctx::DET_SLOTS;
    for (int k = 0; k < S; k++) {
        if (!c->det_slot[k]) c->det_slot[k] = new Detector();
        if (!c->det_stream[k]) HIPCHK(hipStreamCreateWithFlags(&c->det_stream[k], hipStreamNonBlocking));
    }
    std::string err;
    int rc = 0;
    for (int t = 0; t < n + 2 && rc == 0; t++) {
        if (t < n) rc = c->det_slot[t % S]->begin(c->det_stream[t % S], gray[t], rows, cols, stride_row, stride_col, err);
        if (rc == 0 && t >= 1 && t - 1 < n) rc = c->det_slot[(t - 1) % S]->mid(c->det_stream[(t - 1) % S], n1, n2, nullptr, err);
        if (rc == 0 && t >= 2) {
            int f = 0;
            rc = c->det_slot[(t - 2) % S]->end(c->det_stream[(t - 2) % S], n1, n2, corners[t - 2], &f, err);
            found[t - 2] = f;
        }
    }
    if (rc != 0) {
        for (int k = 0; k < S; k++) (void)hipStreamSynchronize(c->det_stream[k]);     // nothing of this call stays in flight
        return rc == -1 ? fail(CCAL_E_ARG, "ccal_detect_corners_many: " + err) : fail(CCAL_E_DEVICE, err);
    }
    return CCAL_OK;
}

// ---- diagnostics -----------------------------------------------------------------------
int ccal_debug_image_grams(ccal_ctx *c, const double intr[CCAL_NINTR], const double *rt_host, double *G_host) {
    if (!c || !intr || !rt_host || !G_host) return fail(CCAL_E_ARG, "ccal_debug_image_grams: NULL argument");
    if (!c->have_problem || !c->have_img) return fail(CCAL_E_STATE, "ccal_debug_image_grams: problem not set");
    HIPCHK(hipSetDevice(c->device));
    int rc = ccal_set_extrinsics(c, rt_host);
    if (rc) return rc;
    if (!c->d_gram) HIPCHK(hipMalloc(&c->d_gram, sizeof(double) * c->n_cap * 256));
    HIPCHK(hipMemcpyAsync(c->d_intr_scratch, intr, sizeof(double) * 9, hipMemcpyHostToDevice, c->stream));
    TrialLaunch t = base_launch(c);
    t.st = c->d_st_scratch; t.trace = nullptr; t.host_result = nullptr; t.host_progress = nullptr;
    t.acc = c->d_acc_scratch;          // diagnostics launches never touch the accumulators of a solve
    t.intr0 = c->d_intr_scratch; t.gram_out = c->d_gram; t.first = 1; t.k = 0;
    HIPCHK(launch_lm_trial(c->stream, t));
    HIPCHK(hipMemcpyAsync(G_host, c->d_gram, sizeof(double) * (size_t)c->N * 256, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->have_result = false;
    return CCAL_OK;
}

int ccal_time_sweep(ccal_ctx *c, int32_t reps, double *avg_us) {
    if (!c || !avg_us || reps < 1) return fail(CCAL_E_ARG, "ccal_time_sweep: bad argument");
    if (!c->have_problem || !c->have_img || !c->have_intr || !c->have_ext)
        return fail(CCAL_E_STATE, "ccal_time_sweep: needs image points and a parameter point");
    REFUSE_PENDING_FOCAL(c, "ccal_time_sweep");
    HIPCHK(hipSetDevice(c->device));
    // a first-trial launch (state boot + full sweep) at the last result if there is one, else at the
    // initial values; back-to-back launches, so the mean includes one kernel boundary
    TrialLaunch t = base_launch(c);
    t.st = c->d_st_scratch; t.trace = nullptr; t.host_result = nullptr; t.host_progress = nullptr;
    t.acc = c->d_acc_scratch;
    t.first = 1; t.k = 0;
    if (c->have_result) {
        const LmState *fin = c->d_st + (c->h_st->trials & 1);
        t.intr0 = reinterpret_cast<const double *>(reinterpret_cast<const char *>(fin) + offsetof(LmState, intr_prev));
        t.ext0 = c->d_blk + (size_t)c->h_st->parity * c->N * BLK + BLK_EXT;
        t.ext0_stride = BLK;
    }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; w++) HIPCHK(launch_lm_trial(c->stream, t));
    HIPCHK(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; r++) HIPCHK(launch_lm_trial(c->stream, t));
    HIPCHK(hipEventRecord(e1, c->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->have_result = false;   // blk[0] heads now hold the timed point, not the fit's buffers
    *avg_us = 1000.0 * ms / reps;
    return CCAL_OK;
}

int ccal_debug_p2p_world(ccal_ctx *c, int32_t world, int32_t rounds, uint32_t seq0, const double *rows, double *sums) {
    if (!c || !rows || !sums) return fail(CCAL_E_ARG, "ccal_debug_p2p_world: NULL argument");
    if (world < 1 || world > P2P_MAX_RANKS || rounds < 1 || rounds > 4096) return fail(CCAL_E_ARG, "ccal_debug_p2p_world: 1..16 ranks, 1..4096 rounds");
    HIPCHK(hipSetDevice(c->device));
    int timed_out = 0;
    HIPCHK(p2p_world_debug(c->stream, world, rounds, seq0, rows, sums, &timed_out));
    if (timed_out) return fail(CCAL_E_COMM, "ccal_debug_p2p_world: an exchange timed out");
    return CCAL_OK;
}

int ccal_debug_multi_set_epoch(int32_t n_dev, const int32_t *dev_ids, uint32_t epoch) {
    if (!dev_ids || n_dev < 1) return fail(CCAL_E_ARG, "ccal_debug_multi_set_epoch: bad argument");
    const std::vector<int> want(dev_ids, dev_ids + n_dev);
    for (auto &m : t_multi) {
        if (m.devices != want) continue;
        for (ccal_ctx *x : m.ctx) {
            HIPCHK(hipSetDevice(x->device));
            HIPCHK(hipStreamSynchronize(x->stream));
            if (x->p2p) HIPCHK(x->p2p->set_epoch(epoch));
        }
        return CCAL_OK;
    }
    return fail(CCAL_E_STATE, "ccal_debug_multi_set_epoch: no such device set on this thread (call ccal_calibrate_multi first)");
}

int ccal_debug_get_homographies(ccal_ctx *c, double *H_host) {
    if (!c || !H_host) return fail(CCAL_E_ARG, "ccal_debug_get_homographies: NULL argument");
    if (!c->have_problem || !c->have_H) return fail(CCAL_E_STATE, "ccal_debug_get_homographies: no intrinsics stage has run on these corners");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(H_host, c->d_H, sizeof(double) * 18 * (size_t)c->N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

int ccal_debug_acc_sum(ccal_ctx *c, const double *rows, int32_t n_rows, int32_t replicas, double out[68]) {
    if (!c || !rows || !out) return fail(CCAL_E_ARG, "ccal_debug_acc_sum: NULL argument");
    if (n_rows < 1 || n_rows > 1024 || replicas < 1 || replicas > ACC_MAX_REPLICAS || (replicas & (replicas - 1)) != 0)
        return fail(CCAL_E_ARG, "ccal_debug_acc_sum: 1..1024 rows, replicas a power of two up to 64");
    HIPCHK(hipSetDevice(c->device));
    const size_t b_in = align256(sizeof(double) * (size_t)n_rows * NRED);
    const int rc = scratch_need(c, b_in + 1024);
    if (rc) return rc;
    double *d_in = static_cast<double *>(c->d_scratch), *d_out = reinterpret_cast<double *>(static_cast<char *>(c->d_scratch) + b_in);
    HIPCHK(hipMemcpyAsync(d_in, rows, sizeof(double) * (size_t)n_rows * NRED, hipMemcpyHostToDevice, c->stream));
    HIPCHK(launch_acc_debug(c->stream, d_in, n_rows, c->d_acc_scratch, replicas, d_out));
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(double) * NRED, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CCAL_OK;
}

#ifdef CCAL_STAMPS
int ccal_debug_init_stamps(unsigned long long *out) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(ccal::debug_read_init_stamps(out));
    return CCAL_OK;
}
int ccal_debug_init_blocks(unsigned int *out) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(ccal::debug_read_init_blocks(out));
    return CCAL_OK;
}
int ccal_debug_wg_times(unsigned long long *out) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(ccal::debug_read_wg_times(out));
    return CCAL_OK;
}
int ccal_debug_stamps(unsigned long long *out) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(ccal::debug_read_stamps(out));
    return CCAL_OK;
}
#endif

}  // extern "C"
