// libplship.so: the one-launch small-rank step whose operand ends with PRIOR rows (SrStepP.Ndata), without the energy by-product.
#include "small_rank_step_launch.inc"

namespace plship {
int launch_small_rank_step_prior(const SrStepP &p, hipStream_t st) { return launch_small_rank_step_any<false, true>(p, st); }
}  // namespace plship
