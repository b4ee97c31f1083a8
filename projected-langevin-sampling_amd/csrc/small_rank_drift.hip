// libplship.so: the SR_MODE_DRIFT instantiations of the fused small-rank kernel.
#include "small_rank_launch.inc"

namespace plship {
int launch_small_rank_drift(const SmallRankP &p, int64_t nsplit, hipStream_t st) { return launch_small_rank<SR_MODE_DRIFT>(p, nsplit, st); }
}  // namespace plship
