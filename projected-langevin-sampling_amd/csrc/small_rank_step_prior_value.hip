// libplship.so: the one-launch small-rank step whose operand ends with PRIOR rows (SrStepP.Ndata), WITH the energies of its input.
#include "small_rank_step_launch.inc"

namespace plship {
int launch_small_rank_step_prior_value(const SrStepP &p, hipStream_t st) { return launch_small_rank_step_any<true, true>(p, st); }
}  // namespace plship
