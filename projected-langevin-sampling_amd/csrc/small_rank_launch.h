// Entry points of the fused small-rank kernels (small_rank.h), one translation unit per mode.
#pragma once
#include <hip/hip_runtime.h>

#include "small_rank.h"

namespace plship {
int launch_small_rank_drift(const SmallRankP &p, int64_t nsplit, hipStream_t st);        // D = Lb^T cost'(Lb V)
int launch_small_rank_value(const SmallRankP &p, int64_t nsplit, hipStream_t st);        // sum_rows cost(Lb V)
int launch_small_rank_drift_value(const SmallRankP &p, int64_t nsplit, hipStream_t st);  // both from the same F
// ranks 129 .. 256 (small_rank2.h: the rank split over wave pairs)
int launch_small_rank2_drift(const SmallRankP &p, int64_t nsplit, hipStream_t st);
int launch_small_rank2_drift_value(const SmallRankP &p, int64_t nsplit, hipStream_t st);
}  // namespace plship
