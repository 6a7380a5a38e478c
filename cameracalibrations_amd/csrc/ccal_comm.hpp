// ccal_comm.hpp -- native RCCL all-reduce for the image-sharded fit.
//
// The reduced intrinsic system (112 doubles) is summed across GPUs with ncclAllReduce on the
// launch stream, between the reduce and solve halves of k_lm_solve: one tiny, latency-bound
// collective per LM trial over xGMI.  RCCL is resolved at run time (dlopen of librccl.so.1), so
// the single-GPU library has no link-time dependency on it; inside a PyTorch process the
// loader hands back the RCCL that torch already loaded.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace ccal {

struct RcclApi;   // resolved function table

class Comm {
  public:
    static int unique_id(char out[128], std::string &err);
    Comm() = default;
    ~Comm();
    int init(int world, int rank, const char id[128], std::string &err);
    int all_reduce_sum(double *buf, int n, hipStream_t stream, std::string &err);
    bool ready() const { return comm_ != nullptr; }

  private:
    void *comm_ = nullptr;
};

}  // namespace ccal
