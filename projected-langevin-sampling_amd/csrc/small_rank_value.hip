// libplship.so: the SR_MODE_VALUE instantiations of the fused small-rank kernel.
#include "small_rank_launch.inc"

namespace plship {
int launch_small_rank_value(const SmallRankP &p, int64_t nsplit, hipStream_t st) { return launch_small_rank<SR_MODE_VALUE>(p, nsplit, st); }
}  // namespace plship
