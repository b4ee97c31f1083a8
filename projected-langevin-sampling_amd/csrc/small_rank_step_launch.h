// Entry points of the one-launch small-rank step (small_rank_step.h), one translation unit per variant.
#pragma once
#include <hip/hip_runtime.h>

#include "small_rank_step.h"

namespace plship {
int launch_small_rank_step(const SrStepP &p, hipStream_t st);        // the step alone
int launch_small_rank_step_value(const SrStepP &p, hipStream_t st);  // ... with the energies of the input particles
int launch_small_rank_step_prior(const SrStepP &p, hipStream_t st);        // the same two for an operand that ends with prior rows
int launch_small_rank_step_prior_value(const SrStepP &p, hipStream_t st);  // (SrStepP.Ndata < N)
}  // namespace plship
