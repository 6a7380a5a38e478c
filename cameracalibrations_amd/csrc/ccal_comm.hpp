// ccal_comm.hpp -- native RCCL all-reduce for the image-sharded fit.
//
// The reduced intrinsic system (16 condensed rows of 68 doubles) is summed across GPUs with
// ncclAllReduce on the launch stream, between k_lm_row and the next k_lm_trial: one tiny,
// latency-bound collective per LM trial.  (The default transport is the direct row exchange of
// ccal_p2p.hpp; this one stays for nodes where peer mapping is unavailable.)  RCCL is resolved at run time (dlopen of librccl.so.1), so
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
