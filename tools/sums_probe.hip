// sums_probe.hip -- two forms of "NV sums over the 64 lanes, every lane gets them all": the product's LDS transpose
// (wave_sums_lds, ccal_wave.hpp) and a halving butterfly in registers (wave_sums_regs, below: not in the product -- it
// measured the same, profiles/r03_variant_probes.txt); checks both against the host and times one call of each in a
// dependent loop (one wave, s_memtime = shader cycles).
//   hipcc --offload-arch=gfx950 -O3 -I cameracalibrations_amd/csrc -o /tmp/sums_probe tools/sums_probe.hip && /tmp/sums_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "ccal_wave.hpp"

using namespace ccal;

namespace ccal {
// The same sums without the LDS transpose: a halving butterfly in registers.  Level by level the wave splits on one lane
// bit (5, 4, 3, 2, 1): of every pair of values (k, k + half) the lanes with the bit clear keep k, the others k + half, and
// what a lane gives up is what its partner keeps -- so one exchange + one add per PAIR, and the number of live values
// halves (32 -> 16 -> 8 -> 4 -> 2 -> 1).  The exchanges cost no selects where the hardware can mask a move:
//   bit 5  v_permlane32_swap (upper half of one operand <-> lower half of the other), bit 4  v_permlane16_swap (odd rows
//   <-> even rows), bits 3 / 2  v_mov_dpp row_ror:8 / row_shr:4 + row_shl:4 under a bank mask (banks = 4 lanes),
//   bit 1  quad_perm + v_cndmask (no mask that fine), bit 0  plain butterfly of the one value left.
// Lane l ends with the total of value l >> 1; the totals go through LDS once (res[], 32 doubles) so that every lane reads
// them all.  ~115 vector instructions + one LDS round trip against 28 stores + 16 wide reads + 32 adds + two round trips.
// Fixed order of additions: the same bits on every run.
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ double dpp_merge(double old, double src) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK_MASK, false),
                            __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK_MASK, false));
}
__device__ __forceinline__ double halve_bit5(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double halve_bit4(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double halve_bit3(double x, double y) {
    return dpp_merge<0x128, 0xC>(x, y) + dpp_merge<0x128, 0x3>(y, x);          // row_ror:8
}
__device__ __forceinline__ double halve_bit2(double x, double y) {
    return dpp_merge<0x114, 0xA>(x, y) + dpp_merge<0x104, 0x5>(y, x);          // row_shr:4 | row_shl:4
}
__device__ __forceinline__ double halve_bit1(double x, double y, int lane) {
    const bool up = (lane & 2) != 0;
    const double keep = up ? y : x, give = up ? x : y;
    return keep + dpp_mov<0x4E>(give);                                         // quad_perm [2,3,0,1]
}

template <int NV>
__device__ __forceinline__ void wave_sums_regs(const double (&v)[NV], double *res, int lane, double (&out)[NV]) {
    static_assert(NV >= 17 && NV <= 32, "written for 17..32 values");
    double a[16], b[8], c[4], d[2];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (k + 16 < NV) ? halve_bit5(v[k], v[k + 16]) : halve_bit5(v[k], v[k]);   // (upper lanes: unused index)
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = halve_bit4(a[k], a[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = halve_bit3(b[k], b[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; k++) d[k] = halve_bit2(c[k], c[k + 2]);
    double e = halve_bit1(d[0], d[1], lane);
    e += dpp_mov<0xB1>(e);                                                     // quad_perm [1,0,3,2]
    if ((lane & 1) == 0) res[lane >> 1] = e;
    wave_lds_sync();
    const double2 *r2 = reinterpret_cast<const double2 *>(res);
#pragma unroll
    for (int j = 0; j + 1 < NV; j += 2) { const double2 t = r2[j >> 1]; out[j] = t.x; out[j + 1] = t.y; }
    if (NV & 1) out[NV - 1] = res[NV - 1];
}

// (round 4) a third candidate, not in the product: measured 1346 against 1241 cycles for 28 sums
// The same sums with the values travelling in PAIRS (NV <= 32): lane l stores (v[2q], v[2q+1]) as one 16-byte word at [q][l] --
// half the store instructions --, lane (q, r) = (lane & 15, lane >> 4) adds the sixteen words of quarter r of row q (two sums at
// once), the four quarters meet through v_permlane16_swap and v_permlane32_swap, and the totals go back as NV / 2 words.  Rows are
// 65 words apart, so that the sixteen lanes of a read group land on sixteen distinct 16-byte slots.  45 LDS instructions for 28
// sums instead of 59.  Fixed order of additions (another order than wave_sums_lds: the two differ in the last bits).
// scratch: the same sums_scratch_doubles(NV) doubles.
constexpr int SUMS2_LD = 65;                                  // 16-byte words per pair row (64 lanes + 1 pad)
template <int NV>
__device__ __forceinline__ void wave_sums_lds2(const double (&v)[NV], double *scratch, int lane, double (&out)[NV]) {
    static_assert(NV >= 2 && NV <= 32, "pairs of up to 32 values");
    constexpr int NP = (NV + 1) / 2;
    static_assert(2 * (SUMS2_LD * NP + NP) <= sums_scratch_doubles(NV), "fits the scratch of wave_sums_lds");
    double2 *s2 = reinterpret_cast<double2 *>(scratch);
#pragma unroll
    for (int q = 0; q < NP; q++) s2[q * SUMS2_LD + lane] = make_double2(v[2 * q], 2 * q + 1 < NV ? v[2 * q + 1] : 0.0);
    wave_lds_sync();
    const int q = lane & 15, r = lane >> 4;
    double x0 = 0.0, x1 = 0.0, y0 = 0.0, y1 = 0.0;
    if (q < NP) {
        const double2 *p = s2 + q * SUMS2_LD + 16 * r;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const double2 d0 = p[i], d1 = p[i + 1];
            x0 += d0.x; y0 += d0.y;
            x1 += d1.x; y1 += d1.y;
        }
    }
    double x = x0 + x1, y = y0 + y1;
    // quarters r and r ^ 1 (rows of 16 lanes), then halves: with both operands the same register every lane ends up with its own
    // value and its partner's
    auto meet16 = [](double a) {
        const unsigned int lo = __double2loint(a), hi = __double2hiint(a);
        const auto sl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto sh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        return __hiloint2double(sh[0], sl[0]) + __hiloint2double(sh[1], sl[1]);
    };
    auto meet32 = [](double a) {
        const unsigned int lo = __double2loint(a), hi = __double2hiint(a);
        const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        return __hiloint2double(sh[0], sl[0]) + __hiloint2double(sh[1], sl[1]);
    };
    x = meet32(meet16(x));
    y = meet32(meet16(y));
    double2 *res = s2 + SUMS2_LD * NP;
    if (lane < NP) res[lane] = make_double2(x, y);
    wave_lds_sync();
#pragma unroll
    for (int q2 = 0; q2 < NP; q2++) {
        const double2 t = res[q2];
        out[2 * q2] = t.x;
        if (2 * q2 + 1 < NV) out[2 * q2 + 1] = t.y;
    }
    wave_lds_sync();                                          // the scratch may be overwritten by the next call
}


}  // namespace ccal

template <int NV, int KIND>
__global__ void k_sums(const double *in, double *out, unsigned long long *cyc, int reps) {
    __shared__ __attribute__((aligned(16))) double scratch[sums_scratch_doubles(32)];
    const int lane = threadIdx.x;
    double v[NV], t[NV], base[NV];
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = base[j] = in[j * 64 + lane];
    if (KIND == 0) wave_sums_lds<NV>(v, scratch, lane, t); else if (KIND == 2) wave_sums_lds2<NV>(v, scratch, lane, t); else wave_sums_regs<NV>(v, scratch, lane, t);
#pragma unroll
    for (int j = 0; j < NV; j++) out[j * 64 + lane] = t[j];
    // dependent loop: the next call's inputs depend on this call's outputs
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = base[j] * 0.5 + t[(j + 1) % NV] * 1e-3;    // (the inputs die in the call, as in the kernels)
        if (KIND == 0) wave_sums_lds<NV>(v, scratch, lane, t); else if (KIND == 2) wave_sums_lds2<NV>(v, scratch, lane, t); else wave_sums_regs<NV>(v, scratch, lane, t);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NV; j++) s += t[j];
    out[NV * 64 + lane] = s;
    if (lane == 0) *cyc = t1 - t0;
}

template <int NV>
static int run() {
    std::vector<double> h(NV * 64), o((NV + 1) * 64);
    for (int j = 0; j < NV; j++)
        for (int l = 0; l < 64; l++) h[j * 64 + l] = std::sin(0.37 * j + 0.11 * l) * (1 + j);
    double *din, *dout;
    unsigned long long *dc, c;
    (void)hipMalloc(&din, sizeof(double) * h.size());
    (void)hipMalloc(&dout, sizeof(double) * o.size());
    (void)hipMalloc(&dc, 8);
    (void)hipMemcpy(din, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
    int bad = 0;
    for (int kind = 0; kind < 3; kind++) {
        const int reps = 200;
        for (int rep = 0; rep < 2; rep++) {
            if (kind == 0) hipLaunchKernelGGL((k_sums<NV, 0>), dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
            else if (kind == 1) hipLaunchKernelGGL((k_sums<NV, 1>), dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
            else hipLaunchKernelGGL((k_sums<NV, 2>), dim3(1), dim3(64), 0, 0, din, dout, dc, reps);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(o.data(), dout, sizeof(double) * o.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        double worst = 0.0;
        for (int j = 0; j < NV; j++) {
            long double ref = 0;
            for (int l = 0; l < 64; l++) ref += h[j * 64 + l];
            for (int l = 0; l < 64; l++) {
                const double e = std::fabs(o[j * 64 + l] - (double)ref);
                if (e > worst) worst = e;
                if (o[j * 64 + l] != o[j * 64]) bad++;          // every lane must hold the same bits
            }
        }
        if (worst > 1e-12) bad++;
        printf("NV %2d  %-15s max |err| %.2e  lanes disagree %d  %6.1f cycles per call (includes %d multiply-adds per lane)\n",
               NV, kind == 0 ? "wave_sums_lds" : (kind == 1 ? "wave_sums_regs" : "wave_sums_lds2"), worst, bad, (double)c / reps, 2 * NV);
    }
    (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dc);
    return bad;
}

int main() {
    int bad = 0;
    bad += run<24>();
    bad += run<28>();
    bad += run<30>();
    printf(bad ? "FAILED\n" : "OK\n");
    return bad ? 1 : 0;
}
