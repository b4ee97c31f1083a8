// libplship.so: the forward GEMM with the cost-VALUE epilogue (stand-alone energy), one instantiation per (cost, link)
// pair the reference's experiments use plus the run-time switch.
#include "common.h"
#include "cost_epilogues.h"
#include "gemm_launch.h"

namespace plship {

template <int COST, int LINK>
static int launch_cl(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                     double *partial, int64_t ldp, const double *y, const CostP &cp, hipStream_t st) {
  GemmShape g{Lf, ldlf, V, ldv, rows, j, kdim, 0, 0, 0};
  if (use_big_tiles(rows, j)) {
    EpiCostValue<128, 128, 64, 64, COST, LINK> e{partial, ldp, y, cp};
    return launch_gemm_cfg<128, 128, 64, 64>(g, e, st);
  }
  EpiCostValue<64, 64, 32, 32, COST, LINK> e{partial, ldp, y, cp};
  return launch_gemm_cfg<64, 64, 32, 32>(g, e, st);
}

int launch_cost_value_gemm(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                           double *partial, int64_t ldp, const double *y, const CostP &cp, hipStream_t st) {
  const int c = cp.cost, l = cp.link;
#define PLS_CL(C, L) \
  if (c == C && l == L) return launch_cl<C, L>(Lf, ldlf, V, ldv, rows, j, kdim, partial, ldp, y, cp, st)
  PLS_CL(PLS_COST_GAUSSIAN, PLS_LINK_IDENTITY);
  PLS_CL(PLS_COST_POISSON, PLS_LINK_SQUARE);
  PLS_CL(PLS_COST_BERNOULLI, PLS_LINK_SIGMOID);
  PLS_CL(PLS_COST_BERNOULLI, PLS_LINK_PROBIT);
  PLS_CL(PLS_COST_STUDENT_T, PLS_LINK_IDENTITY);
  PLS_CL(PLS_COST_MULTIMODAL, PLS_LINK_IDENTITY);
#undef PLS_CL
  return launch_cl<-1, -1>(Lf, ldlf, V, ldv, rows, j, kdim, partial, ldp, y, cp, st);
}

}  // namespace plship
