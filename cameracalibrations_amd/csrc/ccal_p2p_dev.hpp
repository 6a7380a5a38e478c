// ccal_p2p_dev.hpp -- device side of the xGMI row exchange (see ccal_p2p.hpp).
#pragma once

#include "ccal_p2p.hpp"

namespace ccal {

// tag of an exchange and the inbox slot it uses: seq = sequence number of the solve (or of a
// stand-alone all-reduce), k = trial index inside the solve (0 for a stand-alone one)
__device__ __forceinline__ unsigned int p2p_tag(unsigned int seq, unsigned int k) { return (seq << 16) | (k & 0xffffu); }
__device__ __forceinline__ int p2p_slot(unsigned int seq, unsigned int k) { return (int)(((seq & 1u) << 1) | (k & 1u)); }

// One exchange, called by all threads of a workgroup (>= 2 n threads).  s_val[0..n) (LDS) holds
// this rank's row on entry -- the same bits in every workgroup that calls it -- and the sum
// over all ranks, added in rank order, on return.  Only the workgroup with `sender` set stores
// the row into the peers' inboxes; every caller polls the local inbox.  s_all = LDS scratch of
// P2P_MAX_RANKS * P2P_MAX_DOUBLES doubles.  A peer that stays silent for p.budget ticks turns the
// sums into NaN and raises *p.err (false is returned): nothing ever spins for good.
__device__ inline bool p2p_exchange_rows(const P2PDev &p, unsigned int tag, int slot, double *s_val, double *s_all,
                                         int n, int tid, bool sender) {
    const size_t base = (size_t)slot * P2P_MAX_RANKS;
    const int w = tid;
    const bool act = w < 2 * n;
    if (act && sender) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s_val[w >> 1]);
        const unsigned long long half = (w & 1) ? (bits >> 32) : (bits & 0xffffffffull);
        const unsigned long long word = ((unsigned long long)tag << 32) | half;
        for (int r = 0; r < p.world; r++)
            if (r != p.rank)
                __hip_atomic_store(p.peer[r] + (base + p.rank) * P2P_WORDS + w, word, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid < n) s_all[p.rank * P2P_MAX_DOUBLES + tid] = s_val[tid];
    bool ok = true;
    if (act) {
        // all peers' words are requested together (one uncached round trip when the rows are already
        // there, not one per peer); only the ones still missing are asked for again
        const long long t0 = wall_clock64();
        unsigned int *dst = reinterpret_cast<unsigned int *>(s_all);
        unsigned int pending = ((p.world >= 32 ? 0u : (1u << p.world)) - 1u) & ~(1u << p.rank);
        while (pending != 0u) {
            unsigned long long v[P2P_MAX_RANKS];
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; r++)
                if (pending & (1u << r))
                    v[r] = __hip_atomic_load(p.inbox + (base + r) * P2P_WORDS + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; r++)
                if ((pending & (1u << r)) && (unsigned int)(v[r] >> 32) == tag) {
                    dst[r * (2 * P2P_MAX_DOUBLES) + w] = (unsigned int)v[r];   // little-endian halves of double w / 2
                    pending &= ~(1u << r);
                }
            if (pending == 0u) break;
            if (wall_clock64() - t0 > p.budget) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    const int all_ok = __syncthreads_and(ok ? 1 : 0);
    if (tid < n) {
        double t = 0.0;
        for (int r = 0; r < p.world; r++) t += s_all[r * P2P_MAX_DOUBLES + tid];
        s_val[tid] = all_ok ? t : __longlong_as_double(0x7ff8000000000000ll);
    }
    if (tid == 0 && !all_ok) *p.err = 1;
    __syncthreads();
    return all_ok != 0;
}

// The same exchange in two stream-ordered halves, for shards whose kernels share ONE in-order queue (several
// shards of a fit on one device, driven by one host thread): a kernel that waits for a row must never sit in
// front of the kernel that sends it, so every shard's SEND is enqueued before any shard's RECEIVE.
//   p2p_send_row   stores the row into EVERY inbox, the sender's own included, and returns (never waits)
//   p2p_recv_rows  waits for every rank's row (own included: it arrived through the own inbox) and adds them in
//                  rank order -- the same bits as p2p_exchange_rows; waits only for ranks of OTHER queues
__device__ inline void p2p_send_row(const P2PDev &p, unsigned int tag, int slot, const double *s_val, int n, int tid) {
    const size_t base = (size_t)slot * P2P_MAX_RANKS;
    if (tid < 2 * n) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(s_val[tid >> 1]);
        const unsigned long long half = (tid & 1) ? (bits >> 32) : (bits & 0xffffffffull);
        const unsigned long long word = ((unsigned long long)tag << 32) | half;
        for (int r = 0; r < p.world; r++)
            __hip_atomic_store(p.peer[r] + (base + p.rank) * P2P_WORDS + tid, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__device__ inline bool p2p_recv_rows(const P2PDev &p, unsigned int tag, int slot, double *s_val, double *s_all, int n, int tid) {
    const size_t base = (size_t)slot * P2P_MAX_RANKS;
    const int w = tid;
    bool ok = true;
    if (w < 2 * n) {
        const long long t0 = wall_clock64();
        unsigned int *dst = reinterpret_cast<unsigned int *>(s_all);
        unsigned int pending = (p.world >= 32 ? 0u : (1u << p.world)) - 1u;
        while (pending != 0u) {
            unsigned long long v[P2P_MAX_RANKS];
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; r++)
                if (pending & (1u << r))
                    v[r] = __hip_atomic_load(p.inbox + (base + r) * P2P_WORDS + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
            for (int r = 0; r < P2P_MAX_RANKS; r++)
                if ((pending & (1u << r)) && (unsigned int)(v[r] >> 32) == tag) {
                    dst[r * (2 * P2P_MAX_DOUBLES) + w] = (unsigned int)v[r];
                    pending &= ~(1u << r);
                }
            if (pending == 0u) break;
            if (wall_clock64() - t0 > p.budget) { ok = false; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    const int all_ok = __syncthreads_and(ok ? 1 : 0);
    if (tid < n) {
        double t = 0.0;
        for (int r = 0; r < p.world; r++) t += s_all[r * P2P_MAX_DOUBLES + tid];
        s_val[tid] = all_ok ? t : __longlong_as_double(0x7ff8000000000000ll);
    }
    if (tid == 0 && !all_ok) *p.err = 1;
    __syncthreads();
    return all_ok != 0;
}

// stand-alone all-reduce by ONE workgroup: takes the next sequence number
__device__ inline bool p2p_allreduce_wg(const P2PDev &p, double *s_val, double *s_all, int n, int tid) {
    const unsigned int seq = __hip_atomic_load(p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const bool ok = p2p_exchange_rows(p, p2p_tag(seq, 0), p2p_slot(seq, 0), s_val, s_all, n, tid, true);
    if (tid == 0) __hip_atomic_store(p.epoch, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}

}  // namespace ccal
