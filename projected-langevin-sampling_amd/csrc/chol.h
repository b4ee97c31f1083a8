// Launchers of chol.hip used by the inducing-point entry points in plship.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/plship.h"

namespace plship {
// V = L^-T L^-1 U (fwd_only: V = L^-1 U) with the substitution operators of pls_chol_factor; one launch.
int chol_solve_launch(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, int fwd_only,
                      hipStream_t st);
// Y = Lc^-1 U / V = Lc^-T Lc^-1 U: products with the inverse factor when the descriptor carries it (and
// PLS_OPT_SOLVE_MODE != 0), block substitution otherwise.  tmp: m x j doubles (ld j) for the two-product form, may be NULL.
int chol_forward_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *Y, int64_t ldy, hipStream_t st);
int chol_full_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, double *tmp,
                    hipStream_t st);
int launch_transpose(const double *in, int64_t ldi, double *out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t st);
int launch_scale_add_diag(const double *in, int64_t ldi, double alpha, double diag, double *out, int64_t ldo, int64_t m,
                          hipStream_t st);
int64_t solve_mode();  // pls_set_option(PLS_OPT_SOLVE_MODE), defined in plship.hip
}  // namespace plship
