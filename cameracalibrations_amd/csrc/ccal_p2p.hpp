// ccal_p2p.hpp -- direct xGMI exchange of the reduced intrinsic system.
//
// The per-trial message of the image-sharded fit is one row of 68 doubles (544 B): a collective
// library's launch + protocol latency (tens of microseconds for RCCL) is several LM trials long,
// so the ranks of one node exchange it themselves.  Every rank owns an INBOX in fine-grained
// device memory, shared with its peers through hipIpc*MemHandle; a sender stores its row
// straight into each peer's inbox over xGMI as 8-byte words {epoch | 32 payload bits} (the
// "LL" idea: a word is valid when its epoch matches, so no fence and no ordering between stores
// is relied on), a receiver polls its own local inbox and adds the rows in rank order -- every
// rank forms bit-identical sums, so every rank advances the same LM state machine.
//
//   inbox[slot][src rank][136 words]   tag = (seq << 16) | k: seq counts solves (and stand-alone
//                                      all-reduces), k is the trial inside the solve;
//                                      slot = 2 (seq & 1) + (k & 1): consecutive exchanges never
//                                      share a slot, and a peer can be at most one exchange
//                                      ahead (it needs my row n+1 before it can send n+2)
//
// In a solve the exchange is part of k_lm_trial's prologue: every workgroup has just summed the
// local partial rows (identical bits), workgroup 0 stores that row into the peers' inboxes and
// all workgroups poll their own GPU's inbox -- no extra launch, no hand-off between workgroups;
// only the fabric latency is exposed.  Stream-ordered kernels only (graph-capturable, no host
// hook, no RCCL call).
// A receiver that does not see its peers within the time budget poisons the sums with NaN and
// raises an error word instead of hanging the queue.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

namespace ccal {

constexpr int P2P_MAX_RANKS = 16;
constexpr int P2P_MAX_DOUBLES = 68;                    // == NRED
constexpr int P2P_WORDS = 2 * P2P_MAX_DOUBLES;         // 8-byte LL words per row
constexpr int P2P_HANDLE_BYTES = 64;                   // sizeof(hipIpcMemHandle_t)
constexpr int P2P_SLOTS = 4;
constexpr size_t P2P_INBOX_WORDS = (size_t)P2P_SLOTS * P2P_MAX_RANKS * P2P_WORDS;

struct P2PDev {
    unsigned long long *inbox;                 // this rank's inbox (local, fine-grained)
    unsigned long long *peer[P2P_MAX_RANKS];   // peers' inboxes (peer[rank] == inbox)
    unsigned int *epoch;                       // seq of the current solve / last stand-alone all-reduce (device memory)
    int *err;                                  // != 0 after a timed-out exchange (pinned host memory)
    long long budget;                          // wall_clock64 ticks (100 MHz) a receiver waits
    int world, rank;
};

class P2P {
  public:
    P2P() = default;
    ~P2P();
    // allocate the inbox and export its handle (64 bytes)
    int create(int world, int rank, char handle_out[P2P_HANDLE_BYTES], std::string &err);
    // map the peers' inboxes; handles = world x 64 bytes in rank order (own entry ignored)
    int connect(const char *handles, std::string &err);
    // same process, one context per device: the peers' inboxes are plain device pointers (peer access
    // between different devices is switched on here); peers[r] for every rank r, own entry ignored
    int connect_local(P2P *const *peers, std::string &err);
    bool ready() const { return connected_; }
    int world() const { return dev_.world; }
    int rank() const { return dev_.rank; }
    const P2PDev &dev() const { return dev_; }
    // 0 = no exchange has timed out since the last call (clears the flag); stream must be idle
    int check_and_clear(std::string &err);
    // diagnostics: overwrite the solve sequence number (every rank of a fit must be given the same value; streams idle)
    hipError_t set_epoch(unsigned int value) { return hipMemcpy(dev_.epoch, &value, sizeof value, hipMemcpyHostToDevice); }

  private:
    P2PDev dev_{};
    bool created_ = false, connected_ = false, ipc_ = false;
    int device_ = 0;
};

// diagnostics (ccal_debug_p2p_world): `world` workgroups of one launch as the ranks of a node, `rounds` exchanges numbered like
// consecutive solves from sequence number seq0; rows / sums [rounds][world][68] on the host
hipError_t p2p_world_debug(hipStream_t stream, int world, int rounds, unsigned int seq0, const double *rows_host, double *sums_host,
                           int *timed_out);
// buf[0..n) <- sum over ranks (n <= 68), one workgroup on `stream`
hipError_t launch_p2p_allreduce(hipStream_t stream, const P2PDev &p, double *buf, int n);
// the same all-reduce as two kernels, for ranks whose kernels share one in-order queue: every such rank's send is
// enqueued before any of their receives (a kernel that waits must never sit in front of the kernel that sends)
hipError_t launch_p2p_send(hipStream_t stream, const P2PDev &p, const double *buf, int n);
hipError_t launch_p2p_recv(hipStream_t stream, const P2PDev &p, double *buf, int n);

}  // namespace ccal
