// Launchers of chol.hip used by the inducing-point entry points in plship.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/plship.h"

namespace plship {
// V = L^-T L^-1 U (fwd_only: V = L^-1 U) with the substitution operators of pls_chol_factor; one launch.
int chol_solve_launch(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, int fwd_only,
                      hipStream_t st);
}  // namespace plship
