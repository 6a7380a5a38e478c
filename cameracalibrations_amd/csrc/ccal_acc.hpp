// ccal_acc.hpp -- exact, order-independent accumulation of doubles in fixed-point limbs
// (layout and range: ccal_state.hpp).  Plain integer code, __host__ __device__ so that
// tests/test_acc_host.py can build a host shim of the SAME functions and check them against
// arbitrary-precision arithmetic on the CPU; the product only runs them on the device.
#pragma once

#include <stdint.h>
#include <math.h>
#include <string.h>

#include "ccal_state.hpp"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CCAL_ACC_HD __host__ __device__ __forceinline__
#else
#define CCAL_ACC_HD inline
#endif

namespace ccal {

CCAL_ACC_HD int64_t acc_bits(double x) {
    int64_t b;
    memcpy(&b, &x, 8);
    return b;
}

// x -> limb index j and the two signed addends for limbs j and j + 1 (x == lo * 2^(52 j - 150) +
// hi * 2^(52 (j + 1) - 150) exactly, up to bits below 2^-150 which are dropped towards zero).
// flag: 0 ok, 1 NaN, 2 infinite or |x| >= 2^109 (then lo = hi = 0).
CCAL_ACC_HD void acc_split(double x, int &j, int64_t &lo, int64_t &hi, int &flag) {
    // straight-line (selects, no branches): on the device this runs on the critical wave between the workgroup's
    // last barrier and its atomics, and a nest of divergent early-outs costs more than the arithmetic
    const int64_t bits = acc_bits(x);
    const int ex = (int)((bits >> 52) & 0x7ff);
    const int64_t frac = bits & 0xfffffffffffffll;
    const int s = ex - 1075 - ACC_LSB_EXP;                     // position of the mantissa's lowest bit in the fixed-point number
    const bool special = ex == 0x7ff;
    const bool big = !special && (s + 53 > ACC_LIMBS * ACC_LIMB_BITS - 1);
    flag = special ? (frac ? 1 : 2) : (big ? 2 : 0);
    // |x| = m * 2^(ex - 1075); zero / subnormal inputs are far below the resolution and count as zero, bits below
    // 2^-150 are dropped towards zero (a shift of 53 or more leaves nothing of the 53-bit mantissa)
    const int down = s < 0 ? (-s < 63 ? -s : 63) : 0;
    const int64_t m = (ex == 0 || flag != 0) ? 0 : ((frac | (1ll << 52)) >> down);
    const int sp = s < 0 ? 0 : (s > ACC_LIMBS * ACC_LIMB_BITS - 1 ? 0 : s);
    const int jj = (sp * 1261) >> 16;                          // sp / 52 for 0 <= sp < 260
    const int o = sp - ACC_LIMB_BITS * jj;
    j = flag ? 0 : jj;
    const int64_t l = (int64_t)(((uint64_t)m << o) & ((1ull << ACC_LIMB_BITS) - 1));
    const int64_t h = m >> (ACC_LIMB_BITS - o);
    lo = bits < 0 ? -l : l;
    hi = bits < 0 ? -h : h;
}

// One limb's total as two binary64 halves (an address takes at most ACC_MAX_ADDS adds of 52-bit payloads: no int64
// overflow per replica; the replicas are added up in this form, where nothing can overflow): a == (a >> 32) * 2^32 + (a & 0xffffffff) for two's-complement a, both halves
// convert exactly (one v_cvt each on the device).  From here on the decode is floating point: no 64-bit integer
// carry chain, which gfx950 would have to emulate with two 32-bit instructions per operation.
struct AccPart { double hi, lo; };
CCAL_ACC_HD void acc_part_add(AccPart &p, int64_t a) {
    p.hi += (double)(int32_t)(a >> 32);
    p.lo += (double)(uint32_t)((uint64_t)a & 0xffffffffull);
}

// canonical form of a sum of several acc_part_add()s: 0 <= lo < 2^32, same total (all exact while |total| < 2^84), so that
// what acc_decode() makes of the parts is a function of the TOTAL alone, not of how it was spread over replicas
CCAL_ACC_HD void acc_part_norm(AccPart &p) {
    const double c = floor(p.lo * 0x1p-32);
    p.hi += c;
    p.lo -= c * 0x1p32;
}

// limbs (as AccPart sums) -> double: value = sum_j (hi_j 2^32 + lo_j) 2^(52 j - 150), added from the top limb
// down.  Every term is an exact double.  A partial sum is only rounded when it is not about to be cancelled:
// the terms below limb j are together smaller than 2^12 units of limb j, so either the running sum is that
// small and everything is still exact, or it dominates and the rounding is relative to the result.  The
// result is within a few ulp of the exact total and a pure function of it (same bits in every workgroup).
CCAL_ACC_HD double acc_decode(const AccPart part[ACC_LIMBS]) {
    // 2^(52 j - 150) and 2^32 times that
    const double w[ACC_LIMBS] = {0x1p-150, 0x1p-98, 0x1p-46, 0x1p6, 0x1p58};
    const double w32[ACC_LIMBS] = {0x1p-118, 0x1p-66, 0x1p-14, 0x1p38, 0x1p90};
    double v = 0.0;
    for (int j = ACC_LIMBS - 1; j >= 0; j--) {
        v = fma(part[j].hi, w32[j], v);
        v = fma(part[j].lo, w[j], v);
    }
    return v;
}

}  // namespace ccal
