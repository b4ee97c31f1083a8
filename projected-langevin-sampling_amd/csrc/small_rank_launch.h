// Entry points of the fused small-rank kernels (small_rank.h), one translation unit per mode.
#pragma once
#include <hip/hip_runtime.h>

#include "small_rank.h"

namespace plship {
int launch_small_rank_drift(const SmallRankP &p, int64_t nsplit, hipStream_t st);        // D = Lb^T cost'(Lb V)
int launch_small_rank_value(const SmallRankP &p, int64_t nsplit, hipStream_t st);        // sum_rows cost(Lb V)
int launch_small_rank_drift_value(const SmallRankP &p, int64_t nsplit, hipStream_t st);  // both from the same F
}  // namespace plship
