// What the inducing-point basis needs in front of the one-launch small-rank step, in ONE launch (reference:
// basis/inducing_point.py:117-150 -- `gpytorch.solve(k(Z,Z), particles)` and `sample_multivariate_normal(cov = k(Z,Z))`):
//     V = k(Z,Z)^-1 U = Linv^T (Linv U)          (Linv = Lc^-1, k(Z,Z) = Lc Lc^T)
//     E = Lc xi,  xi ~ N(0, 1) from the Philox stream of the step (the same draws as normal_fill_kernel, plship.hip)
// for at most 128 inducing points.  Before: two triangular products, a fill and a third product -- four launches of ~3 us of
// GPU work and ~4 us of launch each, on problems whose whole step is 10-40 us.
//
// One workgroup per 16 particle columns and product chain (V; E when noise is drawn), 4 waves.  The three products are triangular in 16 x 16 tiles: row tile t of a lower
// product has t + 1 tiles of contraction, so wave w takes row tiles w and 7 - w (9 tiles of contraction each way, whatever
// w); the upper product (Linv^T T) mirrors it.  A tile of contraction is four v_mfma_f64_16x16x4: A fragments straight from
// global memory (the factor matrices are stored so that the 16 rows of a fragment are 128 contiguous bytes), B fragments
// from LDS (U, T = Linv U and xi as [row][16] images: a fragment is 512 contiguous bytes, conflict-free).  All the A
// fragments of a workgroup's products are requested before anything else happens (72 buffer loads per wave in flight while the
// particles are staged, instead of a round trip to L2 per tile of contraction).
//
// Layouts (v_mfma_f64_16x16x4, /opt/skills/guides/cdna_hip_programming.md): A lane l = A[row l & 15][k = l >> 4],
// B lane l = B[k = l >> 4][col l & 15], D register r of lane l = D[row (l >> 4) + 4 r][col l & 15].
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "step_params.h"

namespace plship {

constexpr int IPB_PREP_MAX_M = 128;
constexpr int IPB_PREP_COLS = 16;

struct IpbPrepP {
  const double *LinvT;  // [k][i] = Linv[i][k]  (zero for k > i)
  int64_t ldlinvt;
  const double *Linv;  // [k][i] = Linv[k][i]  (zero for i > k)
  int64_t ldlinv;
  const double *LcT;  // [k][i] = Lc[i][k]  (zero for k > i); may be NULL when no noise is drawn
  int64_t ldlct;
  const double *U;  // m x j
  int64_t ldu;
  double *V;  // m x j
  int64_t ldv;
  double *E;  // m x j, written when draw != 0
  int64_t lde;
  int m;
  int64_t j;
  int draw;
  NoiseP nz;
};

int launch_ipb_prep(const IpbPrepP &p, hipStream_t st);

}  // namespace plship
