// libplship.so: solve with k(Z,Z) and colour the noise of a step on at most 128 inducing points in one launch (ipb_prep.h).
#include "ipb_prep.h"

#include "common.h"
#include "philox.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ipb_prep.hip: fp64 MFMA kernel written for gfx950 (MI355X); gfx942 shares the instruction"
#endif

namespace plship {

using prep_f64x4 = __attribute__((ext_vector_type(4))) double;

// The A fragments of one triangular product for this wave.  Row tile t of the output takes the tiles of contraction kt =
// 0 .. t (LOWER) or t .. kb - 1 (upper); wave w owns tiles w and 7 - w of a lower product, kb - 1 - w and kb - 8 + w of an
// upper one, so that a long and a short tile pair up either way.  Positions 0 .. n0 - 1 of the fragment list belong to the
// first tile, n0 .. n0 + n1 - 1 to the second: at most NP = 9 (kb <= 4: one tile per wave, NP = kb positions).
template <int NP>
struct PrepFrags {
  double a[NP][4];
  int t0, t1, n0, n1, k0, k1;
};

template <bool LOWER, int NP>
__device__ __forceinline__ void prep_load(const double *__restrict__ Amat, int64_t lda, int m, int kb, int wave, int lane,
                                          PrepFrags<NP> &f) {
  const int g = lane >> 4, c = lane & 15;
  f.t0 = LOWER ? wave : kb - 1 - wave;
  f.t1 = LOWER ? 7 - wave : kb - 8 + wave;
  const bool has0 = f.t0 >= 0 && f.t0 < kb, has1 = NP > 4 && f.t1 >= 0 && f.t1 < kb;
  f.n0 = has0 ? (LOWER ? f.t0 + 1 : kb - f.t0) : 0;
  f.n1 = has1 ? (LOWER ? f.t1 + 1 : kb - f.t1) : 0;
  f.k0 = LOWER ? 0 : f.t0;
  f.k1 = LOWER ? 0 : f.t1;
  if (!has0) f.t0 = -1;
  if (!has1) f.t1 = -1;
  // Buffer loads, 32-bit offsets: the descriptor ends with the last stored element, so rows k >= m come back as zeros by the
  // range check; fragments of dead positions and of columns >= m get an offset beyond it.  One add per load, no branch, and
  // all of a product's fragments are in flight before anything waits for one.
  const uint32_t bytes = (uint32_t)(((int64_t)(m - 1) * lda + m) * sizeof(double));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Amat), 0, bytes, 0x00020000);
  const int rowstep = (int)(4 * lda * sizeof(double));
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool first = p < f.n0, live = p < f.n0 + f.n1;
    const int t = first ? f.t0 : f.t1, kt = first ? f.k0 + p : f.k1 + (p - f.n0);
    const int col = 16 * t + c;
    const int off0 = (live && col < m) ? (int)((((int64_t)16 * kt + g) * lda + col) * sizeof(double)) : 0x7FFFFFF0;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      f.a[p][s] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (live && col < m) ? off0 + s * rowstep : off0, 0, 0));
  }
}

// out tiles t0 (acc0) and t1 (acc1) of the product whose A fragments are `f`, B from the [row][16] LDS image `bt`
template <int NP>
__device__ __forceinline__ void prep_mfma(const PrepFrags<NP> &f, const double *bt, int lane, prep_f64x4 &acc0, prep_f64x4 &acc1) {
  const int g = lane >> 4, c = lane & 15;
  acc0 = prep_f64x4{0.0, 0.0, 0.0, 0.0};
  acc1 = prep_f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool first = p < f.n0, live = p < f.n0 + f.n1;
    if (!live) break;  // (wave-uniform)
    const int kt = first ? f.k0 + p : f.k1 + (p - f.n0);
    double b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = bt[(16 * kt + 4 * s + g) * IPB_PREP_COLS + c];
    if (first) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[p][s], b[s], acc0, 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f.a[p][s], b[s], acc1, 0, 0, 0);
    }
  }
}

// blockIdx.y = 0: V for 16 columns;  blockIdx.y = 1 (launched when noise is drawn): E for the same columns -- the coloured noise
// does not depend on the particles, so it gets workgroups of its own instead of a place in the solve's critical path.
template <int NP>
__global__ __launch_bounds__(256) void ipb_prep_kernel(IpbPrepP p) {
  __shared__ double lds_u[NP >= 9 ? IPB_PREP_MAX_M * IPB_PREP_COLS : NP * 256];  // U, or xi
  __shared__ double lds_t[NP >= 9 ? IPB_PREP_MAX_M * IPB_PREP_COLS : NP * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, c = lane & 15;
  const int m = p.m, kb = (m + 15) >> 4, mp = kb * 16;
  const int64_t col0 = (int64_t)blockIdx.x * IPB_PREP_COLS;
  const bool col_ok = col0 + c < p.j;
  prep_f64x4 acc0, acc1;
  if (blockIdx.y == 1) {
    PrepFrags<NP> f3;
    prep_load<true, NP>(p.LcT, p.ldlct, m, kb, wave, lane, f3);  // E = Lc xi:  A[k][i] = LcT[k][i], k <= i
    // xi: rows r and r + 4 of every group of 8 share one Philox call (normal_fill_kernel's pairing, the same bits); all mp
    // rows are written: the padding rows of the image must be zeros, not what LDS held
    const uint64_t step = p.nz.live_step();
    for (int e = tid; e < kb * 8 * IPB_PREP_COLS; e += 256) {
      const int pr = e >> 4, cc = e & 15;
      const int ib = (pr >> 2) * 8 + (pr & 3);
      double z0 = 0.0, z1 = 0.0;
      if (ib < m && col0 + cc < p.j) normal_pair(p.nz.seed, step, ib, p.nz.global_column(col0 + cc), z0, z1);
      lds_u[ib * IPB_PREP_COLS + cc] = ib < m ? z0 : 0.0;
      lds_u[(ib + 4) * IPB_PREP_COLS + cc] = ib + 4 < m ? z1 : 0.0;
    }
    __syncthreads();
    prep_mfma<NP>(f3, lds_u, lane, acc0, acc1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int r0 = 16 * f3.t0 + g + 4 * r, r1 = 16 * f3.t1 + g + 4 * r;
      if (f3.t0 >= 0 && r0 < m && col_ok) p.E[(int64_t)r0 * p.lde + col0 + c] = acc0[r];
      if (f3.t1 >= 0 && r1 < m && col_ok) p.E[(int64_t)r1 * p.lde + col0 + c] = acc1[r];
    }
    return;
  }
  // the factor's fragments first: they travel while the particles are staged
  PrepFrags<NP> f1, f2;
  prep_load<true, NP>(p.LinvT, p.ldlinvt, m, kb, wave, lane, f1);  // T = Linv U:     A[k][i] = LinvT[k][i], k <= i
  prep_load<false, NP>(p.Linv, p.ldlinv, m, kb, wave, lane, f2);   // V = Linv^T T:   A[k][i] = Linv[k][i],  k >= i
  for (int e = tid; e < mp * IPB_PREP_COLS; e += 256) {            // (rows beyond m and columns beyond j as zeros)
    const int r = e >> 4, cc = e & 15;
    lds_u[e] = (r < m && col0 + cc < p.j) ? p.U[(int64_t)r * p.ldu + col0 + cc] : 0.0;
  }
  __syncthreads();
  prep_mfma<NP>(f1, lds_u, lane, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (f1.t0 >= 0) lds_t[(16 * f1.t0 + g + 4 * r) * IPB_PREP_COLS + c] = acc0[r];
    if (f1.t1 >= 0) lds_t[(16 * f1.t1 + g + 4 * r) * IPB_PREP_COLS + c] = acc1[r];
  }
  __syncthreads();
  prep_mfma<NP>(f2, lds_t, lane, acc0, acc1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int r0 = 16 * f2.t0 + g + 4 * r, r1 = 16 * f2.t1 + g + 4 * r;
    if (f2.t0 >= 0 && r0 < m && col_ok) p.V[(int64_t)r0 * p.ldv + col0 + c] = acc0[r];
    if (f2.t1 >= 0 && r1 < m && col_ok) p.V[(int64_t)r1 * p.ldv + col0 + c] = acc1[r];
  }
}

int launch_ipb_prep(const IpbPrepP &p, hipStream_t st) {
  if (p.m < 1 || p.m > IPB_PREP_MAX_M) return fail(PLS_ERR_INVALID_ARGUMENT, "ipb_prep: %d inducing points (1 .. 128)", p.m);
  if (!p.LinvT || !p.Linv || !p.U || !p.V || (p.draw && (!p.LcT || !p.E)))
    return fail(PLS_ERR_INVALID_ARGUMENT, "ipb_prep: NULL operand");
  if (p.ldlinvt < p.m || p.ldlinv < p.m || (p.draw && p.ldlct < p.m) || p.ldu < p.j || p.ldv < p.j || (p.draw && p.lde < p.j))
    return fail(PLS_ERR_INVALID_ARGUMENT, "ipb_prep: leading dimension too small");
  if (p.j <= 0) return PLS_OK;
  LaunchScope scope(PLS_TAG_IPB_PREP, st);
  const dim3 grid((unsigned)cdiv(p.j, IPB_PREP_COLS), p.draw ? 2u : 1u);
  const int kb = (p.m + 15) / 16;
  if (kb == 1)
    hipLaunchKernelGGL(ipb_prep_kernel<1>, grid, dim3(256), 0, st, p);
  else if (kb == 2)
    hipLaunchKernelGGL(ipb_prep_kernel<2>, grid, dim3(256), 0, st, p);
  else if (kb <= 4)
    hipLaunchKernelGGL(ipb_prep_kernel<4>, grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(ipb_prep_kernel<9>, grid, dim3(256), 0, st, p);
  return check_launch("ipb_prep");
}

}  // namespace plship
