// libplship.so: the one-launch small-rank step WITHOUT the energy by-product.
#include "small_rank_step_launch.inc"

namespace plship {
int launch_small_rank_step(const SrStepP &p, hipStream_t st) { return launch_small_rank_step_any<false, false>(p, st); }
}  // namespace plship
