// Micro-benchmark (development aid): store patterns for the RBF Gram build k(Z, X) -> 1024 x 1e5 doubles (0.82 GB).
// Same per-element work as kernel_gram_kernel (D = 8); sweeps rows per block, column segments per thread, the
// blockIdx -> tile order and the store cache policy.  Prints ms and TB/s written.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/gram_store_sweep tools/gram_store_sweep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int D = 8;

__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;
  p = fma(p, r, 2.0876756987868098e-09); p = fma(p, r, 2.5052108385441720e-08); p = fma(p, r, 2.7557319223985893e-07);
  p = fma(p, r, 2.7557319223985888e-06); p = fma(p, r, 2.4801587301587302e-05); p = fma(p, r, 1.9841269841269841e-04);
  p = fma(p, r, 1.3888888888888889e-03); p = fma(p, r, 8.3333333333333332e-03); p = fma(p, r, 4.1666666666666664e-02);
  p = fma(p, r, 1.6666666666666666e-01); p = fma(p, r, 0.5); p = fma(p, r, 1.0); p = fma(p, r, 1.0);
  const double v = ldexp(p, (int)n);
  return (x < -745.2) ? 0.0 : v;
}

// block: ROWS rows x (NSEG * 512) columns; thread: NSEG column pairs, 512 columns apart.  ORDER 0: blockIdx.x over
// column tiles, blockIdx.y over row groups (the shipped order).  ORDER 1: linear block id walks row groups fastest
// (neighbouring blocks write the same columns of neighbouring rows).  ORDER 2: as 0 but with the XCD remap (blocks b,
// b + 8, ... are consecutive tiles on one XCD).  NT: non-temporal stores.  COMPUTE 0: store a constant (pure stream).
template <int ROWS, int NSEG, int ORDER, int NT, int COMPUTE>
__global__ __launch_bounds__(256) void gram(const double *__restrict__ x1, int64_t n1, const double *__restrict__ x2,
                                            int64_t n2, double *__restrict__ out, int64_t ld, int ntx, int nty) {
  __shared__ __attribute__((aligned(16))) double a_s[ROWS][D];
  const int t = threadIdx.x;
  int bx, by;
  const int lin = blockIdx.x;
  if (ORDER == 0) { bx = lin % ntx; by = lin / ntx; }
  else if (ORDER == 1) { by = lin % nty; bx = lin / nty; }
  else { const int total = ntx * nty; const int per = (total + 7) / 8; int l2 = (lin % 8) * per + lin / 8; if (l2 >= total) l2 = lin; bx = l2 % ntx; by = l2 / ntx; }
  const int64_t row0 = (int64_t)by * ROWS;
  const int nrows = (int)((n1 - row0 < ROWS) ? (n1 - row0) : ROWS);
  for (int e = t; e < nrows * D; e += 256) a_s[e / D][e % D] = x1[(row0 + e / D) * D + e % D];
  __syncthreads();
  double b0[NSEG][D], b1[NSEG][D];
  int64_t col[NSEG];
#pragma unroll
  for (int s = 0; s < NSEG; ++s) {
    col[s] = ((int64_t)bx * NSEG + s) * 512 + 2 * t;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      b0[s][k] = col[s] < n2 ? x2[col[s] * D + k] : 0.0;
      b1[s][k] = col[s] + 1 < n2 ? x2[(col[s] + 1) * D + k] : 0.0;
    }
  }
#pragma unroll 2
  for (int r = 0; r < nrows; ++r) {
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
      double s0 = 0.0, s1 = 0.0;
      if (COMPUTE) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const double a = a_s[r][k];
          const double e0 = a - b0[s][k], e1 = a - b1[s][k];
          s0 = fma(e0, e0, s0);
          s1 = fma(e1, e1, s1);
        }
        s0 = 1.7 * exp_nonpos(-0.5 * s0);
        s1 = 1.7 * exp_nonpos(-0.5 * s1);
      } else {
        s0 = b0[s][0] + r;
        s1 = b1[s][0] + r;
      }
      if (col[s] + 1 < n2) {
        double2_t *dst = reinterpret_cast<double2_t *>(out + (row0 + r) * ld + col[s]);
        if (NT) __builtin_nontemporal_store(double2_t{s0, s1}, dst); else *dst = double2_t{s0, s1};
      }
    }
  }
}

template <int ROWS, int NSEG, int ORDER, int NT, int COMPUTE>
int run(const double *x1, const double *x2, double *out, int64_t n1, int64_t n2) {
  const int ntx = (int)((n2 + NSEG * 512 - 1) / (NSEG * 512)), nty = (int)((n1 + ROWS - 1) / ROWS);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) gram<ROWS, NSEG, ORDER, NT, COMPUTE><<<ntx * nty, 256>>>(x1, n1, x2, n2, out, n2, ntx, nty);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) gram<ROWS, NSEG, ORDER, NT, COMPUTE><<<ntx * nty, 256>>>(x1, n1, x2, n2, out, n2, ntx, nty);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("rows %3d  seg %d  order %d  nt %d  compute %d : %.4f ms  %.2f TB/s\n", ROWS, NSEG, ORDER, NT, COMPUTE, ms,
         8.0 * n1 * n2 / ms / 1e9);
  return 0;
}

int main() {
  const int64_t n1 = 1024, n2 = 100000;
  std::vector<double> h1(n1 * D), h2(n2 * D);
  for (size_t i = 0; i < h1.size(); ++i) h1[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
  for (size_t i = 0; i < h2.size(); ++i) h2[i] = (double)((i * 40503u) % 2001) / 1000.0 - 1.0;
  double *x1, *x2, *out;
  CK(hipMalloc(&x1, h1.size() * 8)); CK(hipMalloc(&x2, h2.size() * 8)); CK(hipMalloc(&out, n1 * n2 * 8));
  CK(hipMemcpy(x1, h1.data(), h1.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(x2, h2.data(), h2.size() * 8, hipMemcpyHostToDevice));
#define R(...) if (run<__VA_ARGS__>(x1, x2, out, n1, n2)) return 1
  R(64, 1, 0, 0, 1);  // shipped geometry
  R(64, 1, 0, 0, 0);
  R(64, 1, 0, 1, 1);
  R(64, 1, 1, 0, 1);
  R(64, 1, 2, 0, 1);
  R(16, 1, 0, 0, 1); R(32, 1, 0, 0, 1); R(128, 1, 0, 0, 1); R(256, 1, 0, 0, 1);
  R(64, 2, 0, 0, 1); R(64, 4, 0, 0, 1); R(32, 2, 0, 0, 1); R(128, 2, 0, 0, 1); R(32, 4, 0, 0, 1);
  R(16, 1, 1, 0, 1); R(32, 1, 1, 0, 1); R(128, 1, 1, 0, 1);
  R(32, 2, 1, 0, 1); R(64, 2, 1, 0, 1); R(64, 2, 2, 0, 1); R(64, 2, 0, 1, 1);
  R(64, 2, 0, 0, 0); R(64, 4, 0, 0, 0); R(128, 1, 0, 0, 0); R(64, 1, 0, 1, 0); R(64, 1, 1, 0, 0); R(256, 1, 0, 0, 0);
  return 0;
}
