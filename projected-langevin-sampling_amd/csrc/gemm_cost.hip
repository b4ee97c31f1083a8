// libplship.so: the forward GEMM with the cost-derivative epilogues, one instantiation per (cost, link) pair the
// reference's experiments use plus the run-time switch (its own translation unit: 14 GEMM kernels).
#include "common.h"
#include "cost_epilogues.h"
#include "gemm_launch.h"

namespace plship {

template <int COST, int LINK>
static int launch_cl(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                     double *G, int64_t ldg, const double *y, const CostP &cp, double *vpart, int64_t ldp, hipStream_t st) {
  EpiCostDeriv<COST, LINK> e{G, ldg, y, cp, vpart, ldp};
  return launch_gemm(Lf, ldlf, V, ldv, rows, j, kdim, e, st);
}

int launch_cost_deriv_gemm(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                           double *G, int64_t ldg, const double *y, const CostP &cp, double *vpart, int64_t ldp,
                           hipStream_t st) {
  const int c = cp.cost, l = cp.link;
  if (c == PLS_COST_GAUSSIAN && l == PLS_LINK_IDENTITY) {  // direct register -> global epilogue
    EpiGaussDeriv e{G, ldg, y, 1.0 / cp.p0, vpart, ldp};
    return launch_gemm(Lf, ldlf, V, ldv, rows, j, kdim, e, st);
  }
#define PLS_CL(C, L) \
  if (c == C && l == L) return launch_cl<C, L>(Lf, ldlf, V, ldv, rows, j, kdim, G, ldg, y, cp, vpart, ldp, st)
  PLS_CL(PLS_COST_POISSON, PLS_LINK_SQUARE);
  PLS_CL(PLS_COST_BERNOULLI, PLS_LINK_SIGMOID);
  PLS_CL(PLS_COST_BERNOULLI, PLS_LINK_PROBIT);
  PLS_CL(PLS_COST_STUDENT_T, PLS_LINK_IDENTITY);
  PLS_CL(PLS_COST_MULTIMODAL, PLS_LINK_IDENTITY);
#undef PLS_CL
  return launch_cl<-1, -1>(Lf, ldlf, V, ldv, rows, j, kdim, G, ldg, y, cp, vpart, ldp, st);
}

}  // namespace plship
