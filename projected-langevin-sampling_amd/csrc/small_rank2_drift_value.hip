// libplship.so: the SR_MODE_DRIFT_VALUE instantiations of the fused kernel for ranks 129 .. 256 (small_rank2.h).
#include "small_rank2_launch.inc"

namespace plship {
int launch_small_rank2_drift_value(const SmallRankP &p, int64_t nsplit, hipStream_t st) { return launch_small_rank2<SR_MODE_DRIFT_VALUE>(p, nsplit, st); }
}  // namespace plship
