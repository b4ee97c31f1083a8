// Cost functions, their derivatives and the link functions, evaluated per element on the device.
// Reference: src/projected_langevin_sampling/costs/*.py and link_functions.py (lines cited per branch).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "fmath.h"

#include "../../include/plship.h"

namespace plship {

struct CostP {
  int cost, link, mode;
  double p0, p1, p2, p3, jitter;
  double ip0;  // 1 / p0 (Gaussian: 1 / sigma2), computed on the host
  double mm_l1, mm_l2, mm_norm, mm_is2;  // multimodal: log p2, log(1 - p2), 0.5 log(2 pi s2), 1 / s2 (host)
};

__host__ inline CostP make_costp(const pls_cost_desc *d) {
  CostP c;
  c.cost = d->cost;
  c.link = d->link;
  c.mode = d->deriv_mode;
  c.p0 = d->p[0];
  c.p1 = d->p[1];
  c.p2 = d->p[2];
  c.p3 = d->p[3];
  c.jitter = d->jitter;
  c.ip0 = (d->p[0] != 0.0) ? 1.0 / d->p[0] : 0.0;
  c.mm_l1 = c.mm_l2 = c.mm_norm = c.mm_is2 = 0.0;
  if (d->cost == PLS_COST_MULTIMODAL) {
    const double s2 = d->p[0] * d->p[0];
    c.mm_l1 = std::log(d->p[2]);
    c.mm_l2 = std::log(1.0 - d->p[2]);
    c.mm_norm = 0.5 * std::log(2.0 * 3.14159265358979323846 * s2);
    c.mm_is2 = 1.0 / s2;
  }
  return c;
}

__device__ inline double clipd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

// link_functions.py:30-80.  *slope receives d link / d f as torch autograd sees it (0 outside the clip).
// (No floating-point contraction in link_eval / cost_value / cost_deriv: a kernel that evaluates value AND derivative shares
// their common subexpressions, and whether `a * b + c` becomes one fma depends on how many uses the product has -- left to the
// compiler, asking a step for the energy of its input could move the step by an ulp.)
__device__ inline double link_eval(int link, double f, double jit, double *slope) {
#pragma clang fp contract(off)
  switch (link) {
    case PLS_LINK_IDENTITY:  // :54-55
      *slope = 1.0;
      return f;
    case PLS_LINK_SQUARE:  // :79-80
      *slope = 2.0 * f;
      return f * f;
    case PLS_LINK_SIGMOID: {  // :67-70
      const double ex = fast_exp(-f);
      double raw = fast_div(1.0, 1.0 + ex);
      bool inside = (raw >= jit) && (raw <= 1.0 - jit);
      // d/df 1/(1+e^-f) = e^-f / (1+e^-f)^2, the form autograd differentiates (raw*(1-raw) cancels near raw = 1)
      *slope = inside ? ex * raw * raw : 0.0;
      return clipd(raw, jit, 1.0 - jit);
    }
    default: {  // PLS_LINK_PROBIT :39-45
      double raw = 0.5 * (1.0 + erf(f * 0.70710678118654752440));
      bool inside = (raw >= jit) && (raw <= 1.0 - jit);
      *slope = inside ? 0.39894228040143267794 * fast_exp(-0.5 * f * f) : 0.0;
      return clipd(raw, jit, 1.0 - jit);
    }
  }
}

// cost(y, f) for one (n, j) entry; summed over n by the callers.
__device__ inline double cost_value(const CostP &c, double y, double f) {
#pragma clang fp contract(off)
  double slope;
  double p = link_eval(c.link, f, c.jitter, &slope);
  switch (c.cost) {
    case PLS_COST_GAUSSIAN: {  // gaussian.py:63-73 (observation_noise is a variance here)
      double e = p - y;
      return e * e / (2.0 * c.p0);
    }
    case PLS_COST_POISSON:  // poisson.py:59-66
      return -2.0 * y * fast_log(fabs(f)) + p;
    case PLS_COST_BERNOULLI:  // bernoulli.py:57-62
      // binary labels (the usual case) need ONE logarithm; the other term of the general formula is +-0 * finite
      // (p is clipped away from 0 and 1), so these returns are bit-identical to it.  In the row-walking epilogues y
      // is uniform across the wave, so the branch really skips the second log.
      // (one select, ONE logarithm for any mix of the two labels among the lanes of a wave: in the fused small-rank kernels
      // a register holds four different data rows, and `if (y == 1) ... if (y == 0) ...` evaluated both logarithms whenever
      // the four rows did not carry the same label)
      // (with a positive jitter p and 1 - p lie in [jitter, 1 - jitter]: the logarithm without its three special cases)
      if (y == 1.0 || y == 0.0) {
        const double a = (y == 1.0) ? p : 1.0 - p;
        return (c.jitter > 0.0 && c.jitter < 0.5) ? -fast_log_unit(a) : -fast_log(a);
      }
      return -fast_log(p) * y - fast_log(1.0 - p) * (1.0 - y);
    case PLS_COST_STUDENT_T: {  // student_t.py:57-72, p0 = dof, p1 = scale
      double e = p - y;
      return 0.5 * (c.p0 + 1.0) * fast_log(1.0 + e * e / (c.p0 * c.p1 * c.p1));
    }
    default: {  // PLS_COST_MULTIMODAL multimodal.py:37-77, p0 = sigma (std), p1 = shift, p2 = bernoulli_noise
      double e1 = y - p + c.p1, e2 = y - p;
      double a1 = c.mm_l1 - 0.5 * e1 * e1 * c.mm_is2 - c.mm_norm;
      double a2 = c.mm_l2 - 0.5 * e2 * e2 * c.mm_is2 - c.mm_norm;
      double m = fmax(a1, a2);
      return -(m + fast_log(fast_exp(a1 - m) + fast_exp(a2 - m)));
    }
  }
}

// d cost / d f for one entry.
__device__ inline double cost_deriv(const CostP &c, double y, double f) {
#pragma clang fp contract(off)
  double slope;
  double p = link_eval(c.link, f, c.jitter, &slope);
  const bool ref = (c.mode == PLS_DERIV_REFERENCE);
  switch (c.cost) {
    case PLS_COST_GAUSSIAN:  // gaussian.py:86-88 closed form == chain rule for the identity link
      return (p - y) * c.ip0 * slope;  // (* 1/sigma2 instead of / sigma2: <= 1 ulp, and no fp64 division per element)
    case PLS_COST_POISSON:  // poisson.py:76-82 (square link closed form == chain rule); else autograd value
      return fast_div(-2.0 * y, f) + slope;
    case PLS_COST_BERNOULLI:
      if (ref && c.link == PLS_LINK_SIGMOID)  // bernoulli.py:64-77, uses the CLIPPED p
        return -y * (1.0 - p) + (1.0 - y) * p;
      return (fast_div(-y, p) + fast_div(1.0 - y, 1.0 - p)) * slope;
    case PLS_COST_STUDENT_T: {  // student_t.py:82-88
      double e = p - y;
      return fast_div((c.p0 + 1.0) * e, c.p0 * c.p1 * c.p1 + e * e) * slope;
    }
    default: {  // multimodal.py:79-91: always the autograd value
      double e1 = y - p + c.p1, e2 = y - p;
      double a1 = c.mm_l1 - 0.5 * e1 * e1 * c.mm_is2;
      double a2 = c.mm_l2 - 0.5 * e2 * e2 * c.mm_is2;
      double m = fmax(a1, a2);
      double w1 = fast_exp(a1 - m), w2 = fast_exp(a2 - m);
      return -fast_div(w1 * e1 + w2 * e2, w1 + w2) * c.mm_is2 * slope;
    }
  }
}

}  // namespace plship
