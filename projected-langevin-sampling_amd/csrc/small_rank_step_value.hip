// libplship.so: the one-launch small-rank step WITH the energies of its input particles.
#include "small_rank_step_launch.inc"

namespace plship {
int launch_small_rank_step_value(const SrStepP &p, hipStream_t st) { return launch_small_rank_step_any<true, false>(p, st); }
}  // namespace plship
