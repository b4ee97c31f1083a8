// Micro-benchmark: do the f64 MFMA and the fp64 vector ALU of one SIMD overlap when they come from DIFFERENT waves?
// A workgroup of 512 threads = 8 waves = 2 per SIMD.  Waves 0-3 run a v_mfma_f64_16x16x4 loop; waves 4-7 run
//   mode 0: nothing (exit at once)      mode 1: a dependent-free v_fma_f64 loop     mode 2: an integer VALU loop
//   mode 3: also MFMAs (two MFMA waves per SIMD)
// Reported: time of the MFMA waves (s_memtime ticks per MFMA) and of the companion waves, per mode.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void share(double *out, int iters, int valu_iters, int mode, long long *cyc) {
  const int wave = threadIdx.x >> 6;
  long long t0 = __builtin_amdgcn_s_memtime();
  double s = 0;
  if (wave < 4 || mode == 3) {
    double4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (mode == 1) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double m = 0.999999, c = 1e-9;
#pragma unroll 1
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
  } else if (mode == 2) {
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
#pragma unroll 1
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[i]) : "v"(0x9E37u));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  const int ncu = 256, iters = 2000;
  double *out; long long *cyc;
  CK(hipMalloc(&out, sizeof(double) * ncu * 512));
  CK(hipMalloc(&cyc, sizeof(long long) * ncu * 8));
  share<<<ncu, 512>>>(out, 10, 10, 1, cyc);
  CK(hipDeviceSynchronize());
  // companion work sized to about the MFMA waves' duration when alone: iters*16 MFMAs*64 cycles = 2.05M cycles;
  // 32 VALU per companion iteration at 4 cycles -> 128 cycles per iteration -> 16000 iterations
  const int modes[] = {0, 1, 2, 3, 1, 2};
  const int vit[] = {0, 16000, 16000, 0, 4000, 4000};
  for (int k = 0; k < 6; ++k) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    share<<<ncu, 512>>>(out, iters, vit[k], modes[k], cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c[8]; CK(hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost));
    printf("mode %d (companion iters %5d): kernel %.3f ms; MFMA wave %.1f ticks/MFMA (%lld ticks); companion wave %lld ticks", modes[k], vit[k], ms,
           (double)c[0] / (iters * 16.0), c[0], c[4]);
    if (vit[k]) printf(" = %.2f ticks per VALU instr", (double)c[4] / (vit[k] * 32.0));
    printf("\n");
  }
  return 0;
}
