// Device Cholesky factorisation of k(Z,Z) and the per-step solves  V = k(Z,Z)^-1 U  of the inducing-point basis.
//
// Reference: gpytorch.solve(lhs = k(Z,Z), input = rhs = U) at basis/inducing_point.py:89-93, :104-106, :130-132,
// :235-239 (Cholesky + cholesky_solve for the sizes its tests pin) and the per-step eigh(k(Z,Z)) of the
// noise sampler (samplers.py:27-44 via inducing_point.py:133-137), replaced by e = L xi.
//
// Factorisation (once per basis, everything stays on the device, nothing synchronises):
//   right-looking blocked Cholesky, panels of CH_PB = 64 columns:
//     chol_diag_kernel   the 64 x 64 diagonal block, one workgroup, in LDS
//     chol_panel_kernel  the rows below it: x L11^T = a, one row per thread, L11 broadcast from LDS; writes L (lower,
//                        row-major) AND L^T (upper) so that both are k-major operands of the MFMA contraction
//     trailing update    A22 -= L21 L21^T through gemm_tn_f64_kernel (gemm_tn_ex)
//   then the substitution operators: with D_b the inverse of the b-th TS_NB x TS_NB diagonal block of L,
//     forward   y_b = D_b u_b   - sum_{k<b} (D_b L_bk) y_k          Sf[k][i] = (D_b L_bk)^T, D_b^T on the diagonal
//     backward  v_b = D_b^T y_b - sum_{k>b} (D_b^T L_kb^T) v_k      Sb[k][i] = (L_kb D_b),   D_b   on the diagonal
//   i.e. block forward / backward substitution whose diagonal solve is folded into the (pre-scaled) off-diagonal blocks:
//   each block row of a solve is then ONE MFMA k-loop.  Only the 128 x 128 DIAGONAL blocks are inverted (by
//   substitution, what rocBLAS / MAGMA trsm do); k(Z,Z)^-1 is never formed.
//
// Solve (every Langevin step): tri_solve_strip_kernel.  The particle columns are independent, so a workgroup owns a
// strip of TS_SC = 32 columns for the WHOLE solve (forward then backward, block row after block row, one launch, no
// inter-workgroup traffic): its own earlier block rows of V are the R operand of the later ones.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/plship.h"
#include "chol.h"
#include "common.h"
#include "gemm_api.h"

namespace plship {

typedef double double4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

constexpr int CH_PB = 64;   // Cholesky panel width
constexpr int TS_NB = 128;  // block size of the substitution operators
constexpr int TS_SC = 32;   // particle columns per workgroup
constexpr int TS_BK = 32;   // k-depth of a strip-solve sub-step (one set of A fragments)
constexpr int TS_RK = 128;  // rows of the R tile in LDS (one barrier per tile)

// Lc = K (+ jitter on the diagonal); LcT, Sf, Sb = 0
__global__ __launch_bounds__(256) void chol_init_kernel(const double *__restrict__ K, int64_t ldk, int64_t m, double jitter,
                                                         double *__restrict__ Lc, int64_t ldl, double *__restrict__ LcT,
                                                         int64_t ldlt, double *__restrict__ Sf, int64_t ldsf,
                                                         double *__restrict__ Sb, int64_t ldsb, int *info) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && info) *info = 0;
  if (col >= m) return;
  for (int64_t row = blockIdx.y; row < m; row += gridDim.y) {
    double v = K[row * ldk + col];
    if (row == col) v += jitter;
    Lc[row * ldl + col] = v;
    LcT[row * ldlt + col] = 0.0;
    if (Sf) Sf[row * ldsf + col] = 0.0;
    if (Sb) Sb[row * ldsb + col] = 0.0;
  }
}

// Factor the nb x nb diagonal block at (k0, k0) in LDS.  info (device int): 1-based index of the first pivot that is not
// positive (the factor is then garbage and the caller escalates the jitter), 0 otherwise.
__global__ __launch_bounds__(256) void chol_diag_kernel(double *Lc, int64_t ldl, double *LcT, int64_t ldlt, int64_t k0,
                                                         int nb, int *info) {
  __shared__ double S[CH_PB][CH_PB + 1];
  __shared__ double colv[CH_PB];
  const int tid = threadIdx.x;
  for (int e = tid; e < CH_PB * CH_PB; e += 256) {
    const int r = e / CH_PB, c = e % CH_PB;
    S[r][c] = (r < nb && c < nb) ? Lc[(k0 + r) * ldl + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int c = 0; c < nb; ++c) {
    const double d = S[c][c];
    if (!(d > 0.0) && tid == 0 && info && *info == 0) *info = (int)(k0 + c) + 1;
    const double sd = sqrt(d);
    if (tid < CH_PB) colv[tid] = (tid > c) ? S[tid][c] / sd : 0.0;
    __syncthreads();
    for (int e = tid; e < CH_PB * CH_PB; e += 256) {
      const int r = e / CH_PB, k = e % CH_PB;
      if (k > c && r >= k) S[r][k] = fma(-colv[r], colv[k], S[r][k]);
    }
    if (tid < CH_PB) {
      if (tid > c) S[tid][c] = colv[tid];
      if (tid == c) S[c][c] = sd;
    }
    __syncthreads();
  }
  for (int e = tid; e < CH_PB * CH_PB; e += 256) {
    const int r = e / CH_PB, c = e % CH_PB;
    if (r < nb && c < nb) {
      const double lo = (r >= c) ? S[r][c] : 0.0;  // L[r][c]
      const double up = (c >= r) ? S[c][r] : 0.0;  // L^T[r][c] = L[c][r]
      Lc[(k0 + r) * ldl + k0 + c] = lo;
      LcT[(k0 + r) * ldlt + k0 + c] = up;
    }
  }
}

// Rows below a full 64-column panel: row r of A21 -> x with x L11^T = a (forward substitution along the row).
__global__ __launch_bounds__(64) void chol_panel_kernel(double *Lc, int64_t ldl, double *LcT, int64_t ldlt, int64_t k0,
                                                        int64_t m) {
  __shared__ double L11[CH_PB][CH_PB + 1];
  const int tid = threadIdx.x;
  for (int e = tid; e < CH_PB * CH_PB; e += 64) {
    const int r = e / CH_PB, c = e % CH_PB;
    L11[r][c] = Lc[(k0 + r) * ldl + k0 + c];
  }
  __syncthreads();
  const int64_t r = k0 + CH_PB + (int64_t)blockIdx.x * 64 + tid;
  if (r >= m) return;
  double x[CH_PB];
  const double2v *src = reinterpret_cast<const double2v *>(Lc + r * ldl + k0);  // (k0 multiple of 64, ldl even: 16-B aligned)
#pragma unroll
  for (int c = 0; c < CH_PB / 2; ++c) {
    const double2v v = src[c];
    x[2 * c] = v.x;
    x[2 * c + 1] = v.y;
  }
#pragma unroll
  for (int c = 0; c < CH_PB; ++c) {
    double s = x[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s = fma(-x[k], L11[c][k], s);
    x[c] = s / L11[c][c];
  }
  double2v *dst = reinterpret_cast<double2v *>(Lc + r * ldl + k0);
#pragma unroll
  for (int c = 0; c < CH_PB / 2; ++c) dst[c] = double2v{x[2 * c], x[2 * c + 1]};
#pragma unroll
  for (int c = 0; c < CH_PB; ++c) {
    LcT[(k0 + c) * ldlt + r] = x[c];  // L^T[k0 + c][r]: coalesced across the block's rows
    Lc[(k0 + c) * ldl + r] = 0.0;     // the mirrored entry above the diagonal (garbage of the trailing updates)
  }
}

// D = inverse of the nbk x nbk lower-triangular diagonal block at r0 (nbk <= 128): thread c owns column c of D,
// x_i = ([i == c] - sum_{k<i} L_ik x_k) / L_ii by rows; X lives in LDS ([i][c]: conflict-free), row i of L is staged
// once per i.  Writes D into the diagonal block of Sb (row-major) and D^T into that of Sf.
__global__ __launch_bounds__(TS_NB) void tri_block_inverse_kernel(const double *__restrict__ Lc, int64_t ldl, int64_t m,
                                                                   double *__restrict__ Sf, int64_t ldsf,
                                                                   double *__restrict__ Sb, int64_t ldsb) {
  extern __shared__ __attribute__((aligned(16))) double xs[];  // [TS_NB][TS_NB] then one row of L
  double *lrow = xs + TS_NB * TS_NB;
  const int c = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * TS_NB;
  const int nbk = (int)((m - r0 < TS_NB) ? (m - r0) : TS_NB);
  for (int i = 0; i < nbk; ++i) {
    if (c <= i) lrow[c] = Lc[(r0 + i) * ldl + r0 + c];
    __syncthreads();
    if (c < nbk) {
      double s = (i == c) ? 1.0 : 0.0;
      double s2 = 0.0;
      int k = c;  // x_k = 0 for k < c
      for (; k + 1 < i; k += 2) {
        s = fma(-lrow[k], xs[k * TS_NB + c], s);
        s2 = fma(-lrow[k + 1], xs[(k + 1) * TS_NB + c], s2);
      }
      if (k < i) s = fma(-lrow[k], xs[k * TS_NB + c], s);
      xs[i * TS_NB + c] = (i >= c) ? (s + s2) / lrow[i] : 0.0;
    }
    __syncthreads();
  }
  if (c < nbk) {
    for (int i = 0; i < nbk; ++i) {
      const double v = xs[i * TS_NB + c];  // D[i][c]
      Sb[(r0 + i) * ldsb + r0 + c] = v;    // Sb diagonal block = D     (coalesced over c)
      Sf[(r0 + c) * ldsf + r0 + i] = v;    // Sf diagonal block = D^T
    }
  }
}

struct StripArgs {
  const double *Sf, *Sb;
  int64_t ldsf, ldsb;
  const double *U;
  int64_t ldu;
  double *V;
  int64_t ldv;
  int64_t m, j;
};

// One block row of a substitution:  acc(128 x 32) = sum_{k in [kbeg, kend)} S[k][i0 + i] * R[k][j0 + j], where R is Rlo
// for k < ksw and Rhi for k >= ksw (forward: the rows already solved come from V, the block's own rows from U).
struct RowCtx {
  const double *S;
  int64_t lds;
  const double *Rlo;
  int64_t ldlo;
  const double *Rhi;
  int64_t ldhi;
  int64_t ksw, kbeg, kend, i0;
};

__device__ __forceinline__ RowCtx fwd_row(const StripArgs &a, int64_t b) {
  const int64_t i0 = b * TS_NB;
  return RowCtx{a.Sf, a.ldsf, a.V, a.ldv, a.U, a.ldu, i0, 0, (i0 + TS_NB < a.m) ? i0 + TS_NB : a.m, i0};
}
__device__ __forceinline__ RowCtx bwd_row(const StripArgs &a, int64_t b) {
  const int64_t i0 = b * TS_NB;
  return RowCtx{a.Sb, a.ldsb, a.V, a.ldv, a.V, a.ldv, 0, i0, a.m, i0};
}

constexpr int TS_NQ = TS_BK / 4;       // k-quads per sub-step
constexpr int TS_NSUB = TS_RK / TS_BK;  // sub-steps per R tile

// Operand registers that live across block rows: the A fragments of the next sub-step and the next R tile.
struct StripRegs {
  double afr[2][TS_NQ];
  double2v rreg[TS_NSUB];
};

// 8 waves: wave w owns rows [16 w, 16 w + 16) x the strip's 32 columns (one A fragment, two B fragments, two MFMAs per
// k-quad); two waves per SIMD, so one wave's LDS / memory waits sit under the other's MFMAs.
//   * The A operand (the substitution operator S) is NOT shared between the waves -- each owns different output rows --
//     so it never touches LDS: lane l fetches S[k0 + 4 kq + (l >> 4)][i0 + 16 w + (l & 15)] straight into the MFMA
//     operand register (16 consecutive doubles per lane group: four 128-byte segments per wave-instruction), one
//     32-deep sub-step ahead.  (Staged through LDS it cost 32 KB of ds_write per 32 rows.)
//   * The B operand (the strip's rows of R) is shared by all eight waves and goes through LDS in tiles of TS_RK = 128
//     rows (32 KB, double-buffered): ONE barrier per 128 rows of the contraction, 64 MFMAs per wave between barriers.
//   * Addressing costs no vector instruction in the k-loop (every VALU instruction is paid in matrix-pipe issue slots):
//     buffer loads with a descriptor whose base is the first row of the piece (two scalar adds), the row inside it as a
//     scalar offset, and ONE loop-invariant lane offset.  The descriptor's range ends at row kend, so rows past the end
//     of the contraction (the K tail of a matrix whose size is not a multiple of 128) read as zero without a branch.  A
//     column past the matrix edge is CLAMPED, not zeroed: column i of S only ever reaches output row i, column j of R
//     output column j, and the store drops rows >= m and columns >= j.
//   * `have_first`: the row's first R tile and A fragments are already in `regs` (the previous row fetched them under its
//     last MFMAs); `next`: the row whose first tile THIS row fetches under its last MFMAs (NULL: none -- the caller
//     passes it only when that tile does not overlap the rows this row is about to store).
// VEC (workgroup-uniform, chosen by the caller): a full strip -> one 16-byte load per R pair.
// next_kind: 0 none, 1 forward row next_b, 2 backward row next_b (its RowCtx is only built where it is used, at the last
// tile of this row: held across the k-loop the second set of pointers and strides spilled 33 scalar registers)
// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains the
// wave's global loads and stores (vmcnt(0)) -- right for handing over rows of V, wrong inside the k-loop, where the loads
// in flight are the NEXT sub-step's operand fragments.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <bool VEC>
__device__ __forceinline__ void strip_block_row(const StripArgs &a, const RowCtx c, bool have_first, int next_kind,
                                                int64_t next_b, int64_t j0, double *lds, StripRegs &regs,
                                                double4v (&acc)[2]) {
  const int64_t m = a.m, j = a.j;
  constexpr int SR = TS_SC + 16;
  constexpr int NQ = TS_NQ, NSUB = TS_NSUB;
  static_assert(TS_BK == 32 && TS_RK == 128 && (NSUB % 2) == 0, "sub-step parity = A-fragment buffer");
  double *Rs = lds;  // [2][TS_RK][SR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  const int rrow = tid >> 4, lcr = (tid & 15) * 2;  // R tile: 4 passes of 32 rows x 16 threads x 2 doubles
  acc[0] = double4v{0.0, 0.0, 0.0, 0.0};
  acc[1] = double4v{0.0, 0.0, 0.0, 0.0};
  const int64_t cj0 = j0 + lcr;
  const int64_t cr0 = (cj0 < j) ? cj0 : 0, cr1 = (cj0 + 1 < j) ? cj0 + 1 : 0;
  auto clamp = [](int64_t bytes) { return (int)(bytes < 0 ? 0 : (bytes < 0x7FFFFF00 ? bytes : 0x7FFFFF00)); };
  auto load_a = [&](const RowCtx &x, int64_t k0, int nxt) {  // A fragments of the 32-deep sub-step that starts at row k0
#if defined(__HIP_DEVICE_COMPILE__)
    const int64_t ca = (x.i0 + wave * 16 + c16 < m) ? x.i0 + wave * 16 + c16 : 0;
    const int voff_a = (int)((q * x.lds + ca) * 8);
    const int row4 = (int)(x.lds * 32);  // bytes per 4 rows of S
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x.S + k0 * x.lds), 0,
                                                                       clamp((x.kend - k0) * x.lds * 8), 0x00020000);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq)
      regs.afr[nxt][kq] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ra, voff_a, kq * row4, 0));
#else
    (void)x, (void)k0, (void)nxt;
#endif
  };
  auto load_r = [&](const RowCtx &x, int64_t k0) {  // the 128-row R tile at row k0 (never straddles ksw: both multiples of 128)
#if defined(__HIP_DEVICE_COMPILE__)
    const bool hi = k0 >= x.ksw;
    const int64_t ldr = hi ? x.ldhi : x.ldlo;
    const double *rbase = hi ? x.Rhi + k0 * x.ldhi : x.Rlo + k0 * x.ldlo;
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(rbase), 0, clamp((x.kend - k0) * ldr * 8), 0x00020000);
    const int v0 = (int)((rrow * ldr + cr0) * 8), v1 = (int)((rrow * ldr + cr1) * 8);
    const int pass = (int)(ldr * 8 * TS_BK);  // bytes per 32 rows
#pragma unroll
    for (int p = 0; p < NSUB; ++p) {
      if constexpr (VEC) {
        regs.rreg[p] = __builtin_bit_cast(double2v, __builtin_amdgcn_raw_buffer_load_b128(rr, v0, p * pass, 0));
      } else {
        regs.rreg[p].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, v0, p * pass, 0));
        regs.rreg[p].y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, v1, p * pass, 0));
      }
    }
#else
    (void)x, (void)k0, (void)cr0, (void)cr1;
#endif
  };
  auto store_r = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NSUB; ++p)
      *reinterpret_cast<double2v *>(Rs + (buf * TS_RK + p * TS_BK + rrow) * SR + lcr) = regs.rreg[p];
  };
  auto compute = [&](int rbuf, int sub, auto abuf_tag) {
    constexpr int abuf = decltype(abuf_tag)::value;
    // ALL B fragments of the sub-step are fetched in one burst (16 ds_read_b64, 32 VGPRs) and the MFMAs wait on them with
    // counted lgkmcnt; the fences keep the scheduler from sinking every read to just before its MFMA pair, and from
    // hoisting what follows (the next tile's ds_write + barrier) above these MFMAs.
    const double *r = Rs + (rbuf * TS_RK + sub * TS_BK + q) * SR + c16;
    double b[NQ][2];
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      b[kq][0] = r[kq * 4 * SR];
      b[kq][1] = r[kq * 4 * SR + 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(regs.afr[abuf][kq], b[kq][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(regs.afr[abuf][kq], b[kq][1], acc[1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  const int64_t nsup = (c.kend - c.kbeg + TS_RK - 1) / TS_RK;
  if (!have_first) {
    load_r(c, c.kbeg);
    load_a(c, c.kbeg, 0);
  }
  store_r(0);
  __syncthreads();
  // the steady state (every tile but the row's last) is its own loop: no test for the row's end, no second row context
  int64_t sup = 0;
  for (; sup + 1 < nsup; ++sup) {
    const int rbuf = (int)(sup & 1);
    const int64_t k0 = c.kbeg + sup * TS_RK;
    load_r(c, k0 + TS_RK);
    // four sub-steps, A fragments one sub-step ahead
    load_a(c, k0 + TS_BK, 1);
    compute(rbuf, 0, std::integral_constant<int, 0>{});
    load_a(c, k0 + 2 * TS_BK, 0);
    compute(rbuf, 1, std::integral_constant<int, 1>{});
    load_a(c, k0 + 3 * TS_BK, 1);
    compute(rbuf, 2, std::integral_constant<int, 0>{});
    load_a(c, k0 + TS_RK, 0);
    compute(rbuf, 3, std::integral_constant<int, 1>{});
    store_r(rbuf ^ 1);
    lds_barrier();  // (only the R tile is handed over here: the A fragments just requested keep flying across it)
  }
  {  // the row's last tile: the next block row's first tile and fragments fly under its 64 MFMAs
    const int rbuf = (int)(sup & 1);
    const int64_t k0 = c.kbeg + sup * TS_RK;
    RowCtx nx = c;
    if (next_kind) nx = next_kind == 1 ? fwd_row(a, next_b) : bwd_row(a, next_b);
    if (next_kind) load_r(nx, nx.kbeg);
    load_a(c, k0 + TS_BK, 1);  // (rows past kend read as zero)
    compute(rbuf, 0, std::integral_constant<int, 0>{});
    load_a(c, k0 + 2 * TS_BK, 0);
    compute(rbuf, 1, std::integral_constant<int, 1>{});
    load_a(c, k0 + 3 * TS_BK, 1);
    compute(rbuf, 2, std::integral_constant<int, 0>{});
    if (next_kind) load_a(nx, nx.kbeg, 0);
    compute(rbuf, 3, std::integral_constant<int, 1>{});
    __syncthreads();
  }
}

// acc -> V rows [i0, i0 + 128), columns [j0, j0 + 32): register (tb, r) of lane l is row 4 r + (l >> 4) of the wave's 16
// rows, column 16 tb + (l & 15)
__device__ __forceinline__ void strip_store(const double4v (&acc)[2], double *V, int64_t ldv, int64_t i0, int64_t j0,
                                            int64_t m, int64_t j) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = i0 + wave * 16 + 4 * r + q;
      const int64_t col = j0 + tb * 16 + c16;
      if (row < m && col < j) V[row * ldv + col] = acc[tb][r];
    }
}

// V = L^-T L^-1 U for one strip of TS_SC columns: forward block rows 0 .. nb-1 (reads U, writes V), then backward block
// rows nb-1 .. 0 in place.  The strip is private to the workgroup: its earlier stores are ordered before the later
// loads by a workgroup barrier (all waves of a workgroup share the CU's L1).  A row fetches the next row's first operands
// under its own last MFMAs whenever they cannot be rows it is about to store: forward from row 1 on (the next row starts
// at rows 0..127 of V), backward always (the next row starts one block above); not from forward row 0, and not across the
// forward -> backward turn, whose first tile is exactly the block just solved.
// fwd_only != 0 stops after the forward solve (V = L^-1 U).
template <bool VEC>
__device__ __forceinline__ void strip_solve(const StripArgs &a, int fwd_only, int64_t j0, double *lds) {
  const int64_t nb = (a.m + TS_NB - 1) / TS_NB;
  StripRegs regs;
  double4v acc[2];
  bool have = false;
  for (int64_t b = 0; b < nb; ++b) {
    const RowCtx c = fwd_row(a, b);
    const bool pre = b >= 1 && b + 1 < nb;
    strip_block_row<VEC>(a, c, have, pre ? 1 : 0, b + 1, j0, lds, regs, acc);
    strip_store(acc, a.V, a.ldv, c.i0, j0, a.m, a.j);
    have = pre;
    if (!pre) __syncthreads();  // (drains the stores: the next row's first loads read them)
  }
  if (fwd_only) return;
  for (int64_t b = nb - 1; b >= 0; --b) {
    const RowCtx c = bwd_row(a, b);
    const bool pre = b >= 1;
    strip_block_row<VEC>(a, c, have, pre ? 2 : 0, b - 1, j0, lds, regs, acc);
    strip_store(acc, a.V, a.ldv, c.i0, j0, a.m, a.j);
    have = pre;
    if (!pre) __syncthreads();
  }
}

__global__ __launch_bounds__(512) void tri_solve_strip_kernel(StripArgs a, int fwd_only) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int64_t j0 = (int64_t)blockIdx.x * TS_SC;
  // 16-byte loads of the R operand: a full strip, even leading dimensions, 16-byte aligned bases (workgroup-uniform)
  const bool vec = (j0 + TS_SC <= a.j) && (((a.ldu | a.ldv) & 1) == 0) &&
                   (((reinterpret_cast<uintptr_t>(a.U) | reinterpret_cast<uintptr_t>(a.V)) & 15) == 0);
  if (vec)
    strip_solve<true>(a, fwd_only, j0, lds);
  else
    strip_solve<false>(a, fwd_only, j0, lds);
}

static size_t strip_lds_bytes() { return (size_t)2 * TS_RK * (TS_SC + 16) * sizeof(double); }

int chol_solve_launch(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv,
                      int fwd_only, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tri_solve_strip_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)strip_lds_bytes());
    if (e != hipSuccess) return fail(PLS_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  StripArgs a{f->Sf, f->Sb, f->ldsf, f->ldsb, U, ldu, V, ldv, f->m, j};
  {
    LaunchScope scope(PLS_TAG_TRI_SOLVE, st);
    hipLaunchKernelGGL(tri_solve_strip_kernel, dim3((unsigned)cdiv(j, TS_SC)), dim3(512), strip_lds_bytes(), st, a, fwd_only);
  }
  return check_launch("tri_solve_strip");
}

}  // namespace plship

using namespace plship;

extern "C" {

int pls_chol_factor(const double *K, int64_t ldk, int64_t m, double jitter, double *Lc, int64_t ldlc, double *LcT,
                    int64_t ldlct, double *Sf, int64_t ldsf, double *Sb, int64_t ldsb, int32_t *info, void *stream) {
  PLS_REQUIRE(K && Lc && LcT && info, "chol_factor: NULL pointer");
  PLS_REQUIRE((Sf == nullptr) == (Sb == nullptr), "chol_factor: Sf and Sb go together");
  PLS_REQUIRE(m > 0 && ldk >= m && ldlc >= m && ldlct >= m, "chol_factor: bad sizes");
  PLS_REQUIRE(!Sf || (ldsf >= m && ldsb >= m), "chol_factor: bad sizes");
  PLS_REQUIRE((ldlc & 1) == 0 && (ldlct & 1) == 0 && (reinterpret_cast<uintptr_t>(Lc) & 15) == 0,
              "chol_factor: Lc must be 16-byte aligned with even leading dimensions");
  PLS_REQUIRE(jitter >= 0.0, "chol_factor: jitter must be >= 0");
  PLS_REQUIRE(K != Lc, "chol_factor: the factor is built out of place");
  hipStream_t st = S(stream);
  {
    const unsigned gy = (unsigned)(m < 1024 ? m : 1024);
    hipLaunchKernelGGL(chol_init_kernel, dim3((unsigned)cdiv(m, 256), gy), dim3(256), 0, st, K, ldk, m, jitter, Lc, ldlc, LcT,
                       ldlct, Sf, ldsf, Sb, ldsb, info);
    int rc = check_launch("chol_init");
    if (rc) return rc;
  }
  for (int64_t k0 = 0; k0 < m; k0 += CH_PB) {
    const int nb = (int)((m - k0 < CH_PB) ? (m - k0) : CH_PB);
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(256), 0, st, Lc, ldlc, LcT, ldlct, k0, nb, info);
    const int64_t rem = m - k0 - nb;
    if (rem > 0) {  // (then nb == CH_PB)
      hipLaunchKernelGGL(chol_panel_kernel, dim3((unsigned)cdiv(rem, 64)), dim3(64), 0, st, Lc, ldlc, LcT, ldlct, k0, m);
      int rc = check_launch("chol_panel");
      if (rc) return rc;
      // A22 -= L21 L21^T :  L = R = L^T[k0 : k0 + 64, k0 + 64 : m]  (64 x rem, k-major)
      const double *p = LcT + k0 * ldlct + (k0 + nb);
      rc = gemm_tn_ex(p, ldlct, p, ldlct, Lc + (k0 + nb) * ldlc + (k0 + nb), ldlc, rem, rem, nb, -1.0, 1.0, 0, st);
      if (rc) return rc;
    }
  }
  int rc = check_launch("cholesky");
  if (rc || !Sf) return rc;
  return pls_chol_build_operators(Lc, ldlc, LcT, ldlct, m, Sf, ldsf, Sb, ldsb, stream);
}

int pls_chol_build_operators(const double *Lc, int64_t ldlc, const double *LcT, int64_t ldlct, int64_t m, double *Sf,
                             int64_t ldsf, double *Sb, int64_t ldsb, void *stream) {
  PLS_REQUIRE(Lc && LcT && Sf && Sb, "chol_build_operators: NULL pointer");
  PLS_REQUIRE(m > 0 && ldlc >= m && ldlct >= m && ldsf >= m && ldsb >= m, "chol_build_operators: bad sizes");
  hipStream_t st = S(stream);
  int rc;
  static bool attr_set = false;
  const size_t inv_lds = (size_t)(TS_NB * TS_NB + TS_NB) * sizeof(double);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(tri_block_inverse_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)inv_lds);
    if (e != hipSuccess) return fail(PLS_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int64_t nbk = cdiv(m, TS_NB);
  hipLaunchKernelGGL(tri_block_inverse_kernel, dim3((unsigned)nbk), dim3(TS_NB), inv_lds, st, Lc, ldlc, m, Sf, ldsf, Sb, ldsb);
  rc = check_launch("tri_block_inverse");
  if (rc) return rc;
  for (int64_t b = 0; b < nbk; ++b) {
    const int64_t r0 = b * TS_NB, r1 = (r0 + TS_NB < m) ? r0 + TS_NB : m, w = r1 - r0;
    if (r0 > 0) {  // Sf[0 : r0, r0 : r1] = -Lc[r0 : r1, 0 : r0]^T D_b^T
      rc = gemm_tn_ex(Lc + r0 * ldlc, ldlc, Sf + r0 * ldsf + r0, ldsf, Sf + r0, ldsf, r0, w, w, -1.0, 0.0, 0, st);
      if (rc) return rc;
    }
    if (r1 < m) {  // Sb[r1 : m, r0 : r1] = -LcT[r0 : r1, r1 : m]^T D_b
      rc = gemm_tn_ex(LcT + r0 * ldlct + r1, ldlct, Sb + r0 * ldsb + r0, ldsb, Sb + r1 * ldsb + r0, ldsb, m - r1, w, w, -1.0,
                      0.0, 0, st);
      if (rc) return rc;
    }
  }
  return PLS_OK;
}

int pls_chol_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, void *stream) {
  PLS_REQUIRE(f && f->Sf && f->Sb && f->m > 0 && f->ldsf >= f->m && f->ldsb >= f->m, "chol_solve: bad factor descriptor");
  PLS_REQUIRE(U && V && j >= 0 && ldu >= j && ldv >= j, "chol_solve: bad arguments");
  PLS_REQUIRE(U != V, "chol_solve: V must not alias U");
  if (j == 0) return PLS_OK;
  return chol_solve_launch(f, U, ldu, j, V, ldv, 0, S(stream));
}

int pls_tri_multiply(const double *LcT, int64_t ldlct, int64_t m, const double *X, int64_t ldx, int64_t j, double *out,
                     int64_t ldo, void *stream) {
  PLS_REQUIRE(LcT && X && out && m > 0 && j >= 0 && ldlct >= m && ldx >= j && ldo >= j, "tri_multiply: bad arguments");
  if (j == 0) return PLS_OK;
  return gemm_tn_ex(LcT, ldlct, X, ldx, out, ldo, m, j, m, 1.0, 0.0, 1, S(stream));
}

}  // extern "C"
