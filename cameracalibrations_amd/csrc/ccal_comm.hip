// ccal_comm.hip -- see ccal_comm.hpp
#include "ccal_comm.hpp"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace ccal {

namespace {
struct UniqueId { char internal[128]; };
typedef int (*fn_get_id)(UniqueId *);
typedef int (*fn_init_rank)(void **, int, UniqueId, int);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_destroy)(void *);
typedef const char *(*fn_errstr)(int);
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;

struct Api {
    void *handle = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    std::string load_error;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (a.handle) break;
        }
        if (!a.handle) { a.load_error = "librccl.so.1 not found"; return; }
        a.get_id = (fn_get_id)dlsym(a.handle, "ncclGetUniqueId");
        a.init_rank = (fn_init_rank)dlsym(a.handle, "ncclCommInitRank");
        a.all_reduce = (fn_all_reduce)dlsym(a.handle, "ncclAllReduce");
        a.destroy = (fn_destroy)dlsym(a.handle, "ncclCommDestroy");
        a.errstr = (fn_errstr)dlsym(a.handle, "ncclGetErrorString");
        if (!a.get_id || !a.init_rank || !a.all_reduce || !a.destroy) a.load_error = "RCCL symbols missing";
    });
    return a;
}

std::string nccl_msg(const char *what, int rc) {
    Api &a = api();
    return std::string(what) + ": " + (a.errstr ? a.errstr(rc) : "error") + " (" + std::to_string(rc) + ")";
}
}  // namespace

int Comm::unique_id(char out[128], std::string &err) {
    Api &a = api();
    if (!a.load_error.empty()) { err = a.load_error; return -1; }
    UniqueId id;
    const int rc = a.get_id(&id);
    if (rc != 0) { err = nccl_msg("ncclGetUniqueId", rc); return -1; }
    memcpy(out, id.internal, 128);
    return 0;
}

int Comm::init(int world, int rank, const char id_bytes[128], std::string &err) {
    Api &a = api();
    if (!a.load_error.empty()) { err = a.load_error; return -1; }
    if (comm_) { a.destroy(comm_); comm_ = nullptr; }
    UniqueId id;
    memcpy(id.internal, id_bytes, 128);
    const int rc = a.init_rank(&comm_, world, id, rank);
    if (rc != 0) { comm_ = nullptr; err = nccl_msg("ncclCommInitRank", rc); return -1; }
    return 0;
}

int Comm::all_reduce_sum(double *buf, int n, hipStream_t stream, std::string &err) {
    const int rc = api().all_reduce(buf, buf, (size_t)n, NCCL_DOUBLE, NCCL_SUM, comm_, stream);
    if (rc != 0) { err = nccl_msg("ncclAllReduce", rc); return -1; }
    return 0;
}

Comm::~Comm() {
    if (comm_) api().destroy(comm_);
    comm_ = nullptr;
}

}  // namespace ccal
