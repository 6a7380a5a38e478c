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
constexpr int TILE_D = 64 * TILE_LD;  // doubles per wave tile (17408 B)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// One lane stores its corner's two 16-wide rows (zeros when !valid) into the wave's tile.
__device__ __forceinline__ void tile_store_rows(double *tile, int lane, const double mu[16], const double mv[16],
                                                bool valid) {
    double2 *dst = reinterpret_cast<double2 *>(tile + lane * TILE_LD);
#pragma unroll
    for (int c = 0; c < 8; c++) {
        dst[c] = valid ? make_double2(mu[2 * c], mu[2 * c + 1]) : make_double2(0.0, 0.0);
        dst[8 + c] = valid ? make_double2(mv[2 * c], mv[2 * c + 1]) : make_double2(0.0, 0.0);
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

// v_mfma_f64_16x16x4_f64 C/D layout: column = lane & 15, row = (lane >> 4) + 4 * reg.
__device__ __forceinline__ void gram_to_lds(double *Gs, int lane, const d4 &acc0, const d4 &acc1) {
#pragma unroll
    for (int r = 0; r < 4; r++) Gs[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc0[r] + acc1[r];
}

}  // namespace ccal
