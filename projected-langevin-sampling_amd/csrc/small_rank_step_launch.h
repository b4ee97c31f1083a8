// Entry points of the one-launch small-rank step (small_rank_step.h), one translation unit per variant.
#pragma once
#include <hip/hip_runtime.h>

#include "small_rank_step.h"

namespace plship {
int launch_small_rank_step(const SrStepP &p, hipStream_t st);        // the step alone
int launch_small_rank_step_value(const SrStepP &p, hipStream_t st);  // ... with the energies of the input particles
}  // namespace plship
