// libplship.so: the SR_MODE_DRIFT instantiations of the fused kernel for ranks 129 .. 256 (small_rank2.h).
#include "small_rank2_launch.inc"

namespace plship {
int launch_small_rank2_drift(const SmallRankP &p, int64_t nsplit, hipStream_t st) { return launch_small_rank2<SR_MODE_DRIFT>(p, nsplit, st); }
}  // namespace plship
