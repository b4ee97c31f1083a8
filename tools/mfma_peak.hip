// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 with nothing else in the loop (inline asm so that the
// compiler cannot add accumulator copies).  Prints TFLOP/s and cycles per MFMA per SIMD for 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters, long long *cyc) {
  double4_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, int threads) {
  int ncu = 256, iters = 4000;
  double *out; long long *cyc;
  CK(hipMalloc(&out, sizeof(double) * ncu * blocks_per_cu * threads));
  CK(hipMalloc(&cyc, sizeof(long long) * ncu * blocks_per_cu));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  mfma_loop<NACC><<<ncu * blocks_per_cu, threads>>>(out, 10, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  mfma_loop<NACC><<<ncu * blocks_per_cu, threads>>>(out, iters, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long c; CK(hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost));
  double waves = (double)ncu * blocks_per_cu * threads / 64.0;
  double flops = waves * (double)iters * NACC * 2048.0;
  double wps = blocks_per_cu * threads / 64.0 / 4.0;
  printf("NACC=%2d waves/SIMD=%.0f: %.3f ms  %.2f TFLOP/s  memtime ticks per MFMA per SIMD = %.1f\n", NACC, wps, ms,
         flops / ms / 1e9, (double)c / (iters * (double)NACC) / wps);
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<16>(1, 256); run<16>(2, 256); run<16>(4, 256);
  run<4>(1, 256); run<4>(2, 256);
  run<1>(1, 256); run<1>(2, 256); run<2>(1, 256); run<2>(2, 256); run<2>(4, 256); run<8>(1, 256); run<8>(2, 256);
  return 0;
}
