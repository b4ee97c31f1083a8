// libplship from plain C++ (no Python, no torch): one fused Langevin step on the orthonormal basis through the C ABI
// of include/plship.h, checked against a scalar host loop, and the device Cholesky + block-substitution solve of the
// inducing-point basis checked by its residual.  This is the boundary a non-Python host would bind.
//   build: hipcc -O2 -std=c++17 -I include examples/cabi_step.cpp -L projected-langevin-sampling_amd -lplship \
//                -Wl,-rpath,$PWD/projected-langevin-sampling_amd -o /tmp/cabi_step
// Reference semantics: projected_langevin_sampling.py:107-123 with basis/orthonormal.py:98-108,128-159 and
// costs/gaussian.py:86-88, costs/poisson.py:76-82.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "plship.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define PLS_OK_(x)                                                                 \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ != 0) {                                                                \
      std::fprintf(stderr, "libplship error %d: %s (%s:%d)\n", rc_, pls_last_error(), __FILE__, __LINE__); \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

template <class T>
static T *to_device(const std::vector<T> &h) {
  T *d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main() {
  const int64_t n = 300, mk = 20, j = 37;  // ragged on purpose
  const double eta = 1e-3, sigma2 = 0.3;
  std::mt19937_64 gen(1);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::vector<double> A(mk * n), At(n * mk), lam(mk), U(mk * j), xi(mk * j), y(n), ycount(n);
  for (int64_t m = 0; m < mk; ++m)
    for (int64_t i = 0; i < n; ++i) At[i * mk + m] = A[m * n + i] = nd(gen) / std::sqrt((double)mk);
  for (auto &v : lam) v = 0.5 + std::abs(nd(gen));
  for (auto &v : U) v = nd(gen);
  for (auto &v : xi) v = nd(gen);
  for (auto &v : y) v = nd(gen);
  for (auto &v : ycount) v = std::floor(4.0 * std::abs(nd(gen)));

  double *dA = to_device(A), *dAt = to_device(At), *dlam = to_device(lam), *dU = to_device(U), *dxi = to_device(xi);
  double *dy = to_device(y), *dyc = to_device(ycount), *dout = nullptr, *de = nullptr;
  HIP_OK(hipMalloc(&dout, mk * j * sizeof(double)));
  HIP_OK(hipMalloc(&de, j * sizeof(double)));
  if (!dA || !dAt || !dlam || !dU || !dxi || !dy || !dyc) return 2;

  pls_onb_desc basis{};
  basis.mk = mk, basis.n = n;
  basis.A = dA, basis.lda = n, basis.At = dAt, basis.ldat = mk, basis.lam = dlam;
  pls_noise_desc noise{};
  noise.kind = PLS_NOISE_INJECTED, noise.xi = dxi, noise.ldxi = j;
  const size_t ws_bytes = pls_onb_step_workspace_bytes(&basis, j, n);
  void *ws = nullptr;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  double worst = 0.0;
  for (int which = 0; which < 2; ++which) {
    pls_cost_desc cost{};
    cost.cost = which == 0 ? PLS_COST_GAUSSIAN : PLS_COST_POISSON;
    cost.link = which == 0 ? PLS_LINK_IDENTITY : PLS_LINK_SQUARE;
    cost.deriv_mode = PLS_DERIV_REFERENCE;
    cost.p[0] = sigma2;
    cost.jitter = 1e-10;
    const std::vector<double> &yy = which == 0 ? y : ycount;
    PLS_OK_(pls_onb_step(&basis, &cost, which == 0 ? dy : dyc, dU, j, j, eta, &noise, dout, j, /*out_mode=*/0,
                         /*force_generic=*/1, de, ws, ws_bytes, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> got(mk * j), e(j);
    HIP_OK(hipMemcpy(got.data(), dout, got.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(e.data(), de, e.size() * sizeof(double), hipMemcpyDeviceToHost));
    // scalar restatement: F = A^T U, G = d cost / d f, dU = -eta A G - eta U / lam + sqrt(2 eta) xi; energy of U
    double err = 0.0, scale = 0.0, eerr = 0.0;
    for (int64_t c = 0; c < j; ++c) {
      std::vector<double> G(n);
      double cost_c = 0.0;
      for (int64_t i = 0; i < n; ++i) {
        double f = 0.0;
        for (int64_t m = 0; m < mk; ++m) f += A[m * n + i] * U[m * j + c];
        if (which == 0) {
          G[i] = (f - yy[i]) / sigma2;
          cost_c += (f - yy[i]) * (f - yy[i]) / (2.0 * sigma2);
        } else {
          G[i] = -2.0 * yy[i] / f + 2.0 * f;
          cost_c += -2.0 * yy[i] * std::log(std::abs(f)) + f * f;
        }
      }
      double prior = 0.0;
      for (int64_t m = 0; m < mk; ++m) {
        double d = 0.0;
        for (int64_t i = 0; i < n; ++i) d += A[m * n + i] * G[i];
        const double want = -eta * d - eta * U[m * j + c] / lam[m] + std::sqrt(2.0 * eta) * xi[m * j + c];
        err = std::fmax(err, std::abs(got[m * j + c] - want));
        scale = std::fmax(scale, std::abs(want));
        prior += 0.5 * U[m * j + c] * U[m * j + c] / lam[m];
      }
      eerr = std::fmax(eerr, std::abs(e[c] - (cost_c + prior)) / std::abs(cost_c + prior));
    }
    std::printf("%s: step max rel err %.2e, input-energy max rel err %.2e\n", which == 0 ? "gaussian/identity" : "poisson/square",
                err / scale, eerr);
    worst = std::fmax(worst, std::fmax(err / scale, eerr));
  }
  // error behaviour across the boundary: a too-small workspace is reported, not crashed on.  (The one-launch small-rank step
  // needs no workspace at this size -- one row slab per column block --, so the slab kernels are selected for this call.)
  pls_set_option(PLS_OPT_SMALL_RANK_STEP, 0);
  pls_cost_desc cost{};
  cost.cost = PLS_COST_GAUSSIAN, cost.link = PLS_LINK_IDENTITY, cost.p[0] = sigma2, cost.jitter = 1e-10;
  const int rc = pls_onb_step(&basis, &cost, dy, dU, j, j, eta, &noise, dout, j, 0, 1, nullptr, ws, 16, st);
  if (rc == 0 || pls_last_error()[0] == '\0') {
    std::fprintf(stderr, "expected a workspace error\n");
    return 4;
  }
  std::printf("workspace error reported: %s\n", pls_last_error());
  pls_set_option(PLS_OPT_SMALL_RANK_STEP, 1);
  // the factorisation behind gpytorch.solve (inducing_point.py:89-93, :130-132) from plain C++: K = Q Q^T + m I (SPD, M = 200:
  // two 128-blocks, the second one partial), K = Lc Lc^T on the device, V = K^-1 U by block substitution, residual on the host
  {
    const int64_t m = 200, jj = 45, ld = 208;  // (even leading dimension, 16-byte aligned rows)
    std::vector<double> Q(m * m), K(m * ld, 0.0), Uh(m * jj);
    for (auto &v : Q) v = nd(gen);
    for (auto &v : Uh) v = nd(gen);
    for (int64_t a = 0; a < m; ++a)
      for (int64_t b = 0; b < m; ++b) {
        double acc = (a == b) ? (double)m : 0.0;
        for (int64_t k = 0; k < m; ++k) acc += Q[a * m + k] * Q[b * m + k];
        K[a * ld + b] = acc;
      }
    double *dK = to_device(K), *dUh = to_device(Uh), *dLc, *dLcT, *dSf, *dSb, *dV;
    int32_t *dinfo;
    HIP_OK(hipMalloc(&dLc, m * ld * sizeof(double)));
    HIP_OK(hipMalloc(&dLcT, m * ld * sizeof(double)));
    HIP_OK(hipMalloc(&dSf, m * ld * sizeof(double)));
    HIP_OK(hipMalloc(&dSb, m * ld * sizeof(double)));
    HIP_OK(hipMalloc(&dV, m * jj * sizeof(double)));
    HIP_OK(hipMalloc(&dinfo, sizeof(int32_t)));
    if (!dK || !dUh) return 2;
    PLS_OK_(pls_chol_factor(dK, ld, m, /*jitter=*/0.0, dLc, ld, dLcT, ld, dSf, ld, dSb, ld, dinfo, st));
    pls_chol_desc f{m, dLc, ld, dLcT, ld, dSf, ld, dSb, ld};
    PLS_OK_(pls_chol_solve(&f, dUh, jj, jj, dV, jj, st));
    HIP_OK(hipStreamSynchronize(st));
    int32_t info = -1;
    std::vector<double> V(m * jj);
    HIP_OK(hipMemcpy(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(V.data(), dV, V.size() * sizeof(double), hipMemcpyDeviceToHost));
    double res = 0.0, sc = 0.0;
    for (int64_t a = 0; a < m; ++a)
      for (int64_t c = 0; c < jj; ++c) {
        double acc = 0.0;
        for (int64_t b = 0; b < m; ++b) acc += K[a * ld + b] * V[b * jj + c];
        res = std::fmax(res, std::abs(acc - Uh[a * jj + c]));
        sc = std::fmax(sc, std::abs(Uh[a * jj + c]));
      }
    std::printf("cholesky: info %d, solve residual max|K V - U| / max|U| = %.2e\n", (int)info, res / sc);
    if (info != 0 || !(res / sc < 1e-11)) {
      std::fprintf(stderr, "cholesky / solve failure\n");
      return 5;
    }
    // a matrix that is not positive definite is reported through `info`, not crashed on
    K[5 * ld + 5] = -1.0;
    HIP_OK(hipMemcpy(dK, K.data(), K.size() * sizeof(double), hipMemcpyHostToDevice));
    PLS_OK_(pls_chol_factor(dK, ld, m, 0.0, dLc, ld, dLcT, ld, dSf, ld, dSb, ld, dinfo, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipMemcpy(&info, dinfo, sizeof(info), hipMemcpyDeviceToHost));
    std::printf("not positive definite: info = %d (first failing pivot, 1-based)\n", (int)info);
    if (info != 6) return 6;
  }
  if (!(worst < 1e-9)) {
    std::fprintf(stderr, "parity failure: %.3e\n", worst);
    return 1;
  }
  std::printf("cabi_step OK\n");
  return 0;
}
