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

// Both Box-Muller outputs of the pair that contains row `ibase` (bit 2 of ibase must be clear).
__device__ inline void normal_pair(uint64_t seed, uint64_t step, int64_t ibase, int64_t jg, double &z_lo,
                                   double &z_hi) {
  uint32_t x[4];
  philox4x32_10((uint32_t)ibase, (uint32_t)jg, (uint32_t)step, (uint32_t)(step >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), x);
  const double two_m53 = 1.1102230246251565e-16;
  uint64_t a = ((uint64_t)x[0] << 32) | x[1];
  uint64_t b = ((uint64_t)x[2] << 32) | x[3];
  double u1 = ((double)(a >> 11) + 0.5) * two_m53;
  double u2 = ((double)(b >> 11) + 0.5) * two_m53;
  double rad = sqrt(-2.0 * fast_log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  z_lo = rad * c;
  z_hi = rad * s;
}

__device__ inline double normal_one(uint64_t seed, uint64_t step, int64_t i, int64_t jg) {
  double a, b;
  normal_pair(seed, step, i & ~(int64_t)4, jg, a, b);
  return (i & 4) ? b : a;
}

}  // namespace plship
