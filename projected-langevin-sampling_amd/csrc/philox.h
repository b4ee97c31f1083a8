// Counter-based normal generator used by the Langevin noise injection (replaces the CPU
// torch.normal + eigh(I) path of the reference: src/samplers.py:27-44 via basis/orthonormal.py:141-145).
//
// Stream definition (tiling independent, GPU-count independent):
//   element (i, jg) of the (rows x J_global) noise matrix of Langevin step `step`:
//     ibase = i with bit 2 cleared;  ctr = {lo32(ibase), lo32(jg), lo32(step), hi32(step)};  key = {lo32(seed), hi32(seed)}
//     (x0,x1,x2,x3) = Philox4x32-10(ctr, key)
//     u1 = ((x0:x1 as u64) >> 11 + 0.5) * 2^-53,  u2 = ((x2:x3 as u64) >> 11 + 0.5) * 2^-53
//     rad = sqrt(-2 ln u1);   z = (i & 4) ? rad * sin(2 pi u2) : rad * cos(2 pi u2)
//   Rows i and i^4 share one Philox call (both Box-Muller outputs are used); in the f64 MFMA C/D layout
//   (row = (lane>>4) + 4*reg) these two rows live in the same lane, so the fused epilogue pays one call per pair.
// The numpy restatement used by the tests is oracle/philox_ref.py.
#pragma once
#include <hip/hip_runtime.h>

#include "fmath.h"
#include <stdint.h>

namespace plship {

struct PhiloxKey {
  uint32_t k0, k1;
};

__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0;
    uint64_t p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// ---- device-side generator -------------------------------------------------------------------------------------
// Every vector instruction of the noise is paid in matrix-pipe issue slots (the f64 MFMA and the VALU do not overlap,
// DESIGN.md section 3), so the pair costs what its instruction count says.  Same stream as above, to rounding:
//   * Philox: one v_mad_u64_u32 per 32 x 32 -> 64 product and ONE v_bitop3_b32 per three-way xor (4 VALU per round);
//   * u1 from the 53 high bits by two exact conversions and two fma (no 64-bit integer -> double sequence);
//   * sqrt by v_rsq_f64 + Newton (the argument -2 ln u1 lies in [1e-16, 75]: no scaling ladder);
//   * sin / cos of 2 pi u2 without any argument reduction: the top 3 random bits ARE the octant, the next 50 the position
//     inside it (reflected in odd octants by complementing the bits), then fdlibm's k_sin / k_cos kernels on [0, pi/4].
__device__ __forceinline__ uint32_t xor3_u32(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

__device__ __forceinline__ void philox4x32_10_device(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                     uint32_t k1, uint32_t (&out)[4]) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0;
    const uint64_t p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = xor3_u32((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3_u32((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += W0;
    k1 += W1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// sqrt(x) for x well inside the normal range: v_rsq_f64 seed, one coupled Newton step for (sqrt, 1 / (2 sqrt)), one residual
// correction; <= 1 ulp
__device__ __forceinline__ double sqrt_normal(double x) {
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rsq(x);
#else
  const double r = 1.0 / __builtin_sqrt(x);
#endif
  double g = x * r, h = 0.5 * r;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  return fma(fma(-g, g, x), h, g);
}

// Both Box-Muller outputs of the pair that contains row `ibase` (bit 2 of ibase must be clear).
__device__ __forceinline__ void normal_pair(uint64_t seed, uint64_t step, int64_t ibase, int64_t jg, double &z_lo,
                                            double &z_hi) {
#pragma clang fp contract(off)
  uint32_t x[4];
  philox4x32_10_device((uint32_t)ibase, (uint32_t)jg, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                       (uint32_t)(seed >> 32), x);
  // u1 = ((x0:x1 >> 11) + 0.5) * 2^-53: the 53-bit integer is x0 * 2^21 + (x1 >> 11), both parts exact in a double
  const double n1 = fma((double)x[0], 2097152.0, (double)(x[1] >> 11));
  const double u1 = fma(n1, 1.1102230246251565e-16, 5.5511151231257827e-17);
  const double rad = sqrt_normal(-2.0 * fast_log_unit(u1));  // (u1 in [2^-54, 1 - 2^-54]: no special cases)
  // 2 pi u2 = (pi / 4) (o + t), o = top 3 bits of x2, t = ((next 50 bits) + 0.5) * 2^-50 in (0, 1)
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)x[2], 29, 1);  // all ones in odd octants: t -> 1 - t
  const uint32_t hi = __builtin_amdgcn_ubfe(x[2] ^ m, 0, 29), lo = __builtin_amdgcn_ubfe(x[3] ^ m, 11, 21);
#else
  const uint32_t m = (x[2] >> 29) & 1 ? 0xFFFFFFFFu : 0u;
  const uint32_t hi = (x[2] ^ m) & 0x1FFFFFFFu, lo = ((x[3] ^ m) >> 11) & 0x1FFFFFu;
#endif
  const double tf = fma((double)hi, 2097152.0, (double)lo);                  // t * 2^50 - 0.5
  const double a = fma(tf, 6.975736996017264e-16, 3.487868498008632e-16);  // (pi / 4) t,  pi/4 * 2^-50 and * 2^-51
  const double z = a * a;
  // fdlibm k_sin / k_cos on [0, pi/4] (fma_k: one vector instruction per Horner step, fmath.h)
  double ps = fma_k(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma_k(z, ps, 2.75573137070700676789e-06);
  ps = fma_k(z, ps, -1.98412698298579493134e-04);
  ps = fma_k(z, ps, 8.33333333332248946124e-03);
  ps = fma_k(z, ps, -1.66666666666666324348e-01);
  const double s = fma(z * a, ps, a);
  double pc = fma_k(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma_k(z, pc, -2.75573143513906633035e-07);
  pc = fma_k(z, pc, 2.48015872894767294178e-05);
  pc = fma_k(z, pc, -1.38888888888741095749e-03);
  pc = fma_k(z, pc, 4.16666666666666019037e-02);
  const double c = fma(z * z, pc, fma(z, -0.5, 1.0));
  // octant o: swap sin / cos in octants 1, 2, 5, 6; sin < 0 in 4..7; cos < 0 in 2..5
  const bool swap = ((x[2] + 0x20000000u) & 0x40000000u) != 0;
  double sn = swap ? c : s, cs = swap ? s : c;
  const uint32_t sneg = x[2] & 0x80000000u, cneg = (x[2] + 0x40000000u) & 0x80000000u;
  sn = __hiloint2double(__double2hiint(sn) ^ (int)sneg, __double2loint(sn));
  cs = __hiloint2double(__double2hiint(cs) ^ (int)cneg, __double2loint(cs));
  z_lo = rad * cs;
  z_hi = rad * sn;
}

__device__ inline double normal_one(uint64_t seed, uint64_t step, int64_t i, int64_t jg) {
  double a, b;
  normal_pair(seed, step, i & ~(int64_t)4, jg, a, b);
  return (i & 4) ? b : a;
}

}  // namespace plship
