// Internal (C++ linkage) entry to the fp64 MFMA contraction for the other translation units of libplship.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plship {
// C(I x J) = alpha * L^T R + beta * C, L (K x I), R (K x J).  tri = 1: L[k][i] == 0 for k > i (an upper-triangular
// k-major operand, e.g. L_c^T): every output tile only contracts k < its last row + 1; tri = 2: L[k][i] == 0 for k < i
// (lower-triangular, e.g. L_c^-1 as the operand of L_c^-T y): only k >= its first row.  Defined in plship.hip.
// tri_scratch (optional, tri != 0): the caller's scratch for the balanced triangular product on few output tiles
// (pls_tri_scratch_bytes; flag words zero on entry, left zero); NULL: one tile per workgroup.
int gemm_tn_ex(const double *L, int64_t ldl, const double *R, int64_t ldr, double *C, int64_t ldc, int64_t I, int64_t J,
               int64_t K, double alpha, double beta, int tri, hipStream_t st, void *tri_scratch = nullptr,
               size_t tri_scratch_bytes = 0);
}  // namespace plship
