// ccal_wave.hpp -- wave64 building blocks shared by the gfx950 kernels (device only).
//
//   * wave_sum        : butterfly all-reduce over the 64 lanes (every lane gets the same bits)
//   * GramTile        : "rows in lanes" -> LDS -> v_mfma_f64_16x16x4_f64 accumulation of
//                       G = M^T M for a (2*npts) x 16 FP64 matrix M whose two rows per corner
//                       are produced by one lane.  This is the LDS-staged JtJ tile reduction
//                       of the north-star design: rows go through a padded LDS tile so that
//                       the MFMA operand reads are conflict-free 512-B wave reads.
#pragma once

#include <hip/hip_runtime.h>

namespace ccal {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TILE_LD = 34;           // doubles per corner: 2 rows x 16 columns + 2 pad (272 B:
                                      // 16-B aligned b128 stores, 8 consecutive lanes hit 32 distinct banks)
#ifdef CCAL_HALF_TILE
constexpr int TILE_PTS = 32;          // corners staged at a time (two rounds per 64-lane pass): 8.7 KB per wave
#else
constexpr int TILE_PTS = 64;
#endif
constexpr int TILE_D = TILE_PTS * TILE_LD;  // doubles per wave tile (17408 B for 64 corners)

// DPP move of a double (two 32-bit halves); CTRL is a v_mov_b32_dpp control word
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false));
}

// Sum over the 64 lanes, same value (bits) in every lane.  Four DPP steps inside each row of 16
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: register-to-register, no LDS crossbar), then
// the four row sums are read with v_readlane and added in row order.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);    // row_half_mirror
    v += dpp_mov<0x140>(v);    // row_mirror
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    t += __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return t;
}

// Value of lane J of the same row of 16 lanes (row_newbcast: a plain VALU move, no SGPR round trip); J constant.
template <int J>
__device__ __forceinline__ double row_bcast16(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xF, 0xF, true);   // one v_mov_b64_dpp
}

// A wave-uniform double moved to a scalar register pair (two v_readfirstlane_b32).  FP64 arithmetic is vector-only on gfx950, so
// values every lane computes alike (rotation, pose, intrinsics) live in VGPRs unless they are moved by hand; as VOP3 operands they
// cost nothing from there (one scalar source per instruction), and 39 of them are 78 vector registers returned to the rows.
__device__ __forceinline__ double to_sgpr(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Sum inside each row of 16 lanes (lanes 16 r .. 16 r + 15), same value in every lane of the row.
__device__ __forceinline__ double row_sum16(double v) {
    v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);    // row_half_mirror
    v += dpp_mov<0x140>(v);    // row_mirror
    return v;
}

__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp_mov<0xB1>(v));
    v = fmax(v, dpp_mov<0x4E>(v));
    v = fmax(v, dpp_mov<0x141>(v));
    v = fmax(v, dpp_mov<0x140>(v));
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double t = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    t = fmax(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16)));
    t = fmax(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32)));
    t = fmax(t, __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48)));
    return t;
}

// One lane stores its corner's two 16-wide rows (zeros when !valid) into the wave's tile.  Value selects, not two
// exec-masked store blocks: a branch would keep all 32 row values live up to it (+40 VGPRs, spills; measured),
// whereas selects let every value be stored as soon as it is computed.
__device__ __forceinline__ void tile_store_rows(double *tile, int lane, const double mu[16], const double mv[16],
                                                bool valid) {
    double2 *dst = reinterpret_cast<double2 *>(tile + (lane & (TILE_PTS - 1)) * TILE_LD);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        dst[c] = valid ? make_double2(mu[2 * c], mu[2 * c + 1]) : make_double2(0.0, 0.0);
        dst[8 + c] = valid ? make_double2(mv[2 * c], mv[2 * c + 1]) : make_double2(0.0, 0.0);
    }
}

// The same without selects, for rows that are already exact zeros in padding lanes (point_rows_K's vm).
__device__ __forceinline__ void tile_store_rows_masked(double *tile, int lane, const double mu[16], const double mv[16]) {
    double2 *dst = reinterpret_cast<double2 *>(tile + (lane & (TILE_PTS - 1)) * TILE_LD);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        dst[c] = make_double2(mu[2 * c], mu[2 * c + 1]);
        dst[8 + c] = make_double2(mv[2 * c], mv[2 * c + 1]);
    }
}

// Wave-private LDS hand-off: the DS unit executes one wave's instructions in order, so a
// ds_write followed by a ds_read of the SAME wave needs no s_barrier; this only stops the
// compiler from moving LDS accesses of different lanes' addresses across the hand-off.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Accumulate the Gram matrix of the first `npts` corners of the tile.  Each MFMA k-step
// consumes 4 rows = 2 corners; lane l supplies M[4s + (l>>4)][l&15] as BOTH operands
// (A = M^T, B = M share one register).  Operands are fetched four k-steps ahead of the MFMAs
// (all 64 corners' rows are always written, zero when invalid, so rounding the step count up
// to a multiple of 4 only adds zero rows); two accumulators hide the dependent-issue latency.
__device__ __forceinline__ void tile_gram(const double *tile, int lane, int npts, d4 &acc0, d4 &acc1) {
    const int rd_off = (lane >> 5) * TILE_LD + ((lane >> 4) & 1) * 16 + (lane & 15);
    const int nsteps = (((npts + 1) >> 1) + 3) & ~3;
    const double *src = tile + rd_off;
    double a0 = src[0], a1 = src[2 * TILE_LD], a2 = src[4 * TILE_LD], a3 = src[6 * TILE_LD];
    for (int s = 0; s < nsteps; s += 4) {
        const int sn = (s + 4 < nsteps) ? s + 4 : s;          // last group re-reads itself (unused)
        const double *nx = src + sn * 2 * TILE_LD;
        const double b0 = nx[0], b1 = nx[2 * TILE_LD], b2 = nx[4 * TILE_LD], b3 = nx[6 * TILE_LD];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, a2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, a3, acc1, 0, 0, 0);
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    }
}

// ---------------------------------------------------------------------------------------
// Sums over the 64 lanes of NV per-lane values (NV <= 64), every lane gets all NV totals -- for the small normal
// equations of the initialisation (28-36 distinct sums per pass over an image's corners), where a 16-wide MFMA Gram
// computes 256 entries to deliver them and costs a tile store, 14 matrix instructions and a read-back per pass.
// Transpose-reduce through LDS: lane l stores value j at [j][l] (rows padded to SUMS_LD doubles: consecutive lanes,
// consecutive addresses on the way in; 16 rows' 16-byte reads on 64 distinct banks on the way out), then lane
// (j, half) adds up one half of row j (NV <= 32; whole rows by lanes j < NV otherwise) with 16-byte reads, the two halves
// meet through v_permlane32_swap, the totals go back through LDS and every lane reads them all.  Fixed order of additions: the same bits on every run.
// scratch: SUMS_LD * NV + 64 doubles, wave-private.
// ---------------------------------------------------------------------------------------
constexpr int SUMS_LD = 66;
constexpr int sums_scratch_doubles(int nv) { return SUMS_LD * nv + 64; }

// The reduce alone: -> the NV totals, left in LDS (readable by every lane; the caller calls wave_lds_sync() when it has read what
// it needs, before the scratch is used again) -- for callers that want a per-lane selection of the totals (one matrix row per
// lane: ccal_init.hip's lane-parallel solves) instead of all of them in every lane.
template <int NV>
__device__ __forceinline__ const double *wave_sums_lds_reduce(const double (&v)[NV], double *scratch, int lane) {
    static_assert(NV >= 1 && NV <= 64, "one row per lane at most");
#pragma unroll
    for (int j = 0; j < NV; j++) scratch[j * SUMS_LD + lane] = v[j];
    wave_lds_sync();
    double *res = scratch + SUMS_LD * NV;
    if constexpr (NV <= 32) {
        const int j = lane & 31, h = lane >> 5;
        double part = 0.0;
        if (j < NV) {
            const double2 *p = reinterpret_cast<const double2 *>(scratch + j * SUMS_LD + 32 * h);
            double a0 = 0.0, a1 = 0.0;                        // two chains: the adds' latency behind the reads'
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const double2 d0 = p[q], d1 = p[q + 1];
                a0 += d0.x; a1 += d1.x;
                a0 += d0.y; a1 += d1.y;
            }
            part = a0 + a1;
        }
        // the two halves of a row meet in registers: v_permlane32_swap exchanges the upper 32 lanes of one operand with
        // the lower 32 of the other, so with both operands the same value every lane ends up with lane j's + lane (32 + j)'s
        {
            const unsigned int lo = __double2loint(part), hi = __double2hiint(part);
            const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            part = __hiloint2double(sh[0], sl[0]) + __hiloint2double(sh[1], sl[1]);
        }
        if (lane < 32) res[lane] = part;
    } else {
        double part = 0.0;
        if (lane < NV) {
            const double2 *p = reinterpret_cast<const double2 *>(scratch + lane * SUMS_LD);
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int q = 0; q < 32; q += 2) {
                const double2 d0 = p[q], d1 = p[q + 1];
                a0 += d0.x; a1 += d1.x;
                a0 += d0.y; a1 += d1.y;
            }
            part = a0 + a1;
        }
        res[lane] = part;
    }
    wave_lds_sync();
    return res;
}

template <int NV>
__device__ __forceinline__ void wave_sums_lds(const double (&v)[NV], double *scratch, int lane, double (&out)[NV]) {
    const double *res = wave_sums_lds_reduce<NV>(v, scratch, lane);
#pragma unroll
    for (int j2 = 0; j2 < NV; j2++) out[j2] = res[j2];
    wave_lds_sync();                                          // the scratch may be overwritten by the next call
}


// v_mfma_f64_16x16x4_f64 C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg.
__device__ __forceinline__ void gram_to_lds(double *Gs, int lane, const d4 &acc0, const d4 &acc1) {
#pragma unroll
    for (int r = 0; r < 4; r++) Gs[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
}

}  // namespace ccal
