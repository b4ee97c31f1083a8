// fp64 exp / log for the per-element code of the hot kernels.
//
// In these kernels every vector-ALU instruction is paid for in matrix-pipe issue slots (the f64 MFMA and the VALU do
// not overlap), and the library exp/log spend about half their instructions on special-case
// ladders, extended-precision tails and constant moves.  These versions keep the classical argument reductions and
// minimax/Taylor kernels (fdlibm's log, a degree-13 exp) and nothing else: <= 1 ulp on the whole fp64 range including
// subnormals, with the IEEE limits (log 0 = -inf, log of a negative = NaN, exp overflow = +inf, exp underflow = 0,
// NaN in = NaN out).  tests/test_gpu_parity.py::test_device_math_matches_libm pins them.
#pragma once
#include <hip/hip_runtime.h>

namespace plship {

// a * b + k for a compile-time constant k.  Written as v_fma_f64 with the constant in a SCALAR register pair: the
// compiler's own choice for a Horner step is v_mov_b64 (copy the constant) + v_fmac_f64 (two-address form), i.e. two
// vector instructions per coefficient, and every vector instruction of these kernels costs matrix-pipe issue slots.
// Each such constant occupies a scalar register pair for as long as the compiler keeps it live, so this form is for
// translation units whose kernels have scalar registers to spare (PLS_SCALAR_POLY_CONSTANTS = 1: plship.hip -- noise
// generator, Langevin epilogue, Gram build); in the cost-epilogue and small-rank units it pushed the scalar file into
// spilling (84 spilled SGPRs and 340 B/lane of scratch in the Poisson drift+value kernel), so they keep the plain fma.
#ifndef PLS_SCALAR_POLY_CONSTANTS
#define PLS_SCALAR_POLY_CONSTANTS 0
#endif
__device__ __forceinline__ double fma_k(double a, double b, double k) {
#if defined(__HIP_DEVICE_COMPILE__) && PLS_SCALAR_POLY_CONSTANTS
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
#else
  return fma(a, b, k);
#endif
}

// exp(x): n = rint(x log2 e), r = x - n ln 2 (two-piece ln 2), degree-13 Taylor polynomial in Horner form
// (|r| <= 0.347: truncation 4e-18 relative), scaled by 2^n with v_ldexp (gradual underflow as libm).
__device__ __forceinline__ double fast_exp(double x) {
#pragma clang fp contract(off)
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = fma_k(p, r, 2.0876756987868098e-09);
  p = fma_k(p, r, 2.5052108385441720e-08);
  p = fma_k(p, r, 2.7557319223985893e-07);
  p = fma_k(p, r, 2.7557319223985888e-06);
  p = fma_k(p, r, 2.4801587301587302e-05);
  p = fma_k(p, r, 1.9841269841269841e-04);
  p = fma_k(p, r, 1.3888888888888889e-03);
  p = fma_k(p, r, 8.3333333333333332e-03);
  p = fma_k(p, r, 4.1666666666666664e-02);
  p = fma_k(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  double v = ldexp(p, (int)n);
  v = (x < -745.2) ? 0.0 : v;
  v = (x > 709.8) ? __builtin_huge_val() : v;  // (NaN fails both comparisons and has already propagated through p)
  return v;
}

// Newton steps on the v_rcp_f64 seed before the quotient's residual correction.  The seed r0 has a relative error e0; one
// step leaves e0^2 in r, the quotient q = a r inherits it, and the correction q + r (a - b q) squares it again: with the
// ~2^-26 seed of gfx9 one step gives 2^-52 before and 2^-104 after the correction, i.e. the correctly rounded quotient up to
// the rounding of the last fma.  Round 2 took two steps; tests/test_gpu_parity.py::test_device_math_matches_libm holds the
// one-step form to the same <= 1 ulp over the whole normal range.
#ifndef PLS_DIV_NEWTON_STEPS
#define PLS_DIV_NEWTON_STEPS 1
#endif

// a / b for b well inside the normal range (no scaling steps): reciprocal seed + Newton + one residual correction of the
// quotient; <= 1 ulp
__device__ __forceinline__ double fast_div_normal(double a, double b) {
#pragma clang fp contract(off)
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
#if PLS_DIV_NEWTON_STEPS >= 2
  r = fma(fma(-b, r, 1.0), r, r);
#endif
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// a / b for per-element cost code: reciprocal seed, Newton, one residual correction of the quotient (<= 1 ulp
// for |b| and |a / b| inside the normal range), then v_div_fixup_f64, which puts the IEEE results of the special cases
// back (b = 0, infinities, NaN).  7 vector instructions against the ~15 of the compiler's scaled division sequence;
// what is given up is correct rounding and the rescaling of operands within a factor 2^-1022 .. 2^1022 of the limits.
__device__ __forceinline__ double fast_div(double a, double b) {
#pragma clang fp contract(off)
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
#if PLS_DIV_NEWTON_STEPS >= 2
  r = fma(fma(-b, r, 1.0), r, r);
#endif
  const double q = a * r;
  return __builtin_amdgcn_div_fixup(fma(fma(-b, q, a), r, q), b, a);
#else
  return a / b;
#endif
}

// log(x) after fdlibm's e_log.c: x = 2^k (1 + f) with 1 + f in [sqrt(1/2), sqrt(2)), s = f / (2 + f),
// log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2))), R the degree-7 minimax polynomial (error < 2^-58.45).
// fast_log_unit: any positive finite argument (subnormals included) -- in particular the uniform deviates of the noise
// generator, (n + 1/2) 2^-53 with 0 <= n < 2^53, strictly inside (0, 1).  fast_log adds the three special cases (zero,
// infinity, negative: a compare and two 32-bit selects each, 9 vector instructions such an argument can never take).
__device__ __forceinline__ double fast_log_unit(double x) {
  // (contraction written out: the noise generator is inlined into several kernels, and left to the compiler `t2 + t1` or
  // `dk * ln2 - ...` became an fma in one and a multiply-add pair in another -- 1 ulp apart in one normal deviate per ~6000,
  // so a particle's noise depended on the tiling its shard happened to take)
#pragma clang fp contract(off)
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1) (subnormals included)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  k = low ? k - 1 : k;
  const double f = m - 1.0;
  const double dk = (double)k;
  const double s = fast_div_normal(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma_k(w, fma_k(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double p2 =
      fma_k(w, fma_k(w, fma_k(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
            6.666666666666735130e-01);
  const double R = fma(z, p2, t1);
  const double hfsq = 0.5 * f * f;
  return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

__device__ __forceinline__ double fast_log(double x) {
  double v = fast_log_unit(x);
  v = (x == 0.0) ? -__builtin_huge_val() : v;
  v = (x == __builtin_huge_val()) ? x : v;
  v = (x < 0.0) ? __builtin_nan("") : v;
  return v;
}

}  // namespace plship
