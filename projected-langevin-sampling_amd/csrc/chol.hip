// Device Cholesky factorisation of k(Z,Z) and the per-step solves  V = k(Z,Z)^-1 U  of the inducing-point basis.
//
// Reference: gpytorch.solve(lhs = k(Z,Z), input = rhs = U) at basis/inducing_point.py:89-93, :104-106, :130-132,
// :235-239 (Cholesky + cholesky_solve for the sizes its tests pin) and the per-step eigh(k(Z,Z)) of the
// noise sampler (samplers.py:27-44 via inducing_point.py:133-137), replaced by e = L xi.
//
// Factorisation (once per basis, everything stays on the device, nothing synchronises):
//   right-looking blocked Cholesky, panels of CH_PB = 64 columns:
//     chol_diag_kernel   the 64 x 64 diagonal block, one wave, in registers
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

// Factor the nb x nb diagonal block at (k0, k0): ONE wave, lane r holds row r in registers, no LDS and no barrier.
// Column c: the pivot and the column below it are broadcast with v_readlane (the lane index is a compile-time constant:
// the loops are fully unrolled), the rank-1 update of the trailing columns is one fma per (row, column) with the
// broadcast value as a scalar operand.  ~8.5 k straight-line instructions = ~17 us; the round-2 first version (256
// threads, the block in LDS, two workgroup barriers per column) took 104 us per panel -- 1.7 of the 2.8 ms of a
// factorisation at M = 1024.
// info (device int): 1-based index of the first pivot that is not positive (the factor is then garbage and the caller
// escalates the jitter), 0 otherwise.
__device__ __forceinline__ double readlane_f64(double v, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
#else
  (void)lane;
  return v;
#endif
}

__global__ __launch_bounds__(64) void chol_diag_kernel(double *Lc, int64_t ldl, double *LcT, int64_t ldlt, int64_t k0,
                                                        int nb, int *info) {
  static_assert(CH_PB == 64, "one lane per row of the panel");
  const int r = threadIdx.x;
  double x[CH_PB];
  if (nb == CH_PB) {  // (k0 is a multiple of 64 and ldl even: 16-byte aligned rows)
    const double2v *src = reinterpret_cast<const double2v *>(Lc + (k0 + r) * ldl + k0);
#pragma unroll
    for (int c = 0; c < CH_PB / 2; ++c) {
      const double2v v = src[c];
      x[2 * c] = v.x;
      x[2 * c + 1] = v.y;
    }
  } else {  // the last, partial block: identity beyond nb
#pragma unroll
    for (int c = 0; c < CH_PB; ++c) x[c] = (r < nb && c < nb) ? Lc[(k0 + r) * ldl + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  int bad = 0;
#pragma unroll
  for (int c = 0; c < CH_PB; ++c) {
    const double d = readlane_f64(x[c], c);
    if (!(d > 0.0) && bad == 0) bad = c + 1;  // (wave-uniform)
    const double sd = sqrt(d), inv = 1.0 / sd;
    const double l = (r > c) ? x[c] * inv : (r == c ? sd : 0.0);
    x[c] = l;
#pragma unroll
    for (int k = c + 1; k < CH_PB; ++k) x[k] = fma(-l, readlane_f64(l, k), x[k]);
  }
  if (bad && r == 0 && info && *info == 0) *info = (int)k0 + bad;
  if (r < nb) {
#pragma unroll
    for (int c = 0; c < CH_PB; ++c)
      if (c < nb) Lc[(k0 + r) * ldl + k0 + c] = (c <= r) ? x[c] : 0.0;  // L[r][c]
  }
#pragma unroll
  for (int c = 0; c < CH_PB; ++c)
    if (c < nb && r < nb) LcT[(k0 + c) * ldlt + k0 + r] = (r >= c) ? x[c] : 0.0;  // L^T[c][r] = L[r][c]: coalesced over the lanes
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

// ---- the strip solve as ONE pipeline of tiles ------------------------------------------------------------------------
// A block row of a substitution is  acc(128 x 32) = sum_k S[k][i0 + i] R[k][j0 + j]  over 128-row tiles of the strip's
// rows R.  Every tile but one holds rows that were final long ago (the right-hand side U, or blocks of V solved at least
// one block row earlier); the exception is the block the PREVIOUS block row has just produced.  The order of the sum is
// free, so every row takes that newest block LAST, and takes it from LDS: a row that finishes writes its 128 x 32 result
// to global memory (output, and operand of the rows after the next) and into a hand-over tile in LDS in the R-tile
// layout.  The rows then chain without a bubble: while the hand-over of row s travels, row s + 1 is already contracting
// its old tiles; nobody waits for a global store to complete (the round-1/early-round-2 kernel drained its stores and
// re-read them through L1 at every block row: 5.4 us x 16 rows of a 0.41 ms solve at M = 1024).  The only true
// dependency left is the forward -> backward turn, where the first backward row needs the last forward block at once.
//
//   forward  row b:  [U block b]  [V blocks 0 .. b-2]        [hand-over: y_(b-1)]
//   backward row b:  [V block b (y_b; the hand-over itself for the first backward row)]  [V blocks b+2 ..]  [hand-over: v_(b+1)]
//
// One tile = four 32-deep sub-steps.  All operands of a sub-step are requested one sub-step ahead: the A fragments (the
// operator S, not shared between the waves: straight from global memory / L2 into the MFMA operand registers) and the B
// fragments (LDS -> registers, double-buffered: 64 VGPRs); the next tile's rows are requested at sub-step 0, written to
// LDS after sub-step 1 and handed over by the tile's only barrier after sub-step 2, so that sub-step 3 can already fetch
// the next tile's first fragments.  The tile body has no branch: a tile without a global source (hand-over, end of the
// solve) loads through a descriptor of range 0 (zeros, no traffic) into an R buffer nobody reads.
// LDS: 2 R tiles + 2 hand-over tiles of 128 x 32 doubles = 128 KB; element (r, c) of a tile sits at
// r * 32 + (c ^ ((r & 1) << 4)) -- odd rows swap their 16-column halves, which makes the B-fragment reads (4 rows x 16
// columns per wave-instruction) conflict-free without row padding.
[[maybe_unused]] constexpr int TS_NQ = TS_BK / 4;        // k-quads per sub-step
[[maybe_unused]] constexpr int TS_NSUB = TS_RK / TS_BK;  // sub-steps per R tile
constexpr int TS_TILE = TS_RK * TS_SC;  // doubles per LDS tile

struct TileDesc {  // wave-uniform
  const double *S;  // first operator row of the tile (S + k0 * lds); range 0 = no tile
  int64_t lds;
  int s_bytes;      // descriptor range of S from that row (0: every load returns zero)
  int64_t i0;       // first output row of the block row
  const double *R;  // first row of the tile in global memory, NULL = hand-over tile (or no tile)
  int64_t ldr;
  int r_bytes;
  int r_is_u;       // R points into U (its leading dimension selects the lane offsets)
  int hbuf;         // hand-over buffer (R == NULL)
  int s, t;         // block-row sequence number (0 .. 2 nb - 1) and tile index inside it
  int first, last, valid;
};

__device__ __forceinline__ int clamp_range(int64_t bytes) {
  return (int)(bytes < 0 ? 0 : (bytes < 0x7FFFFF00 ? bytes : 0x7FFFFF00));
}

struct RowState {  // wave-uniform; what the tiles of one block row share (computed once per row, not once per tile:
  const double *S;   // the eight waves of a workgroup run this bookkeeping at the same moment on the CU's one scalar unit)
  int64_t lds, kend, i0;
  int nt, b, fwd, s, valid;
};

__device__ __forceinline__ RowState make_row(const StripArgs &a, int nb, int nrows, int s) {
  RowState r;
  r.s = s;
  r.valid = s < nrows;
  r.fwd = s < nb;
  r.b = r.fwd ? s : 2 * nb - 1 - s;
  if (!r.valid) r.b = 0;  // (past the last row: a well-formed descriptor of range 0, never a negative row offset)
  r.i0 = (int64_t)r.b * TS_NB;
  r.nt = r.fwd ? r.b + 1 : nb - r.b;
  r.S = r.fwd ? a.Sf : a.Sb;
  r.lds = r.fwd ? a.ldsf : a.ldsb;
  const int64_t kf = (r.i0 + TS_NB < a.m) ? r.i0 + TS_NB : a.m;
  r.kend = r.valid ? (r.fwd ? kf : a.m) : 0;
  if (!r.valid) r.nt = 1;
  return r;
}

// tile t of block row r:
//   forward  row b:  t = 0 the block's rows of U;  1 .. nt-2 blocks 0 .. b-2 of V;  nt-1 (>= 1) the hand-over y_(b-1)
//   backward row b:  t = 0 block b of V (the hand-over itself in the first backward row);  1 .. nt-2 blocks b+2 ..;  nt-1 the
//                    hand-over v_(b+1)
__device__ __forceinline__ TileDesc make_tile(const StripArgs &a, const RowState &r, int t) {
  TileDesc d;
  d.s = r.s;
  d.t = t;
  d.valid = r.valid;
  d.first = (t == 0);
  d.last = (t == r.nt - 1);
  d.i0 = r.i0;
  d.lds = r.lds;
  d.hbuf = (r.s + 1) & 1;
  const bool handover = r.fwd ? (t >= 1 && t == r.nt - 1) : (t == r.nt - 1);
  const int kb = (t == 0) ? r.b : (r.fwd ? (handover ? r.b - 1 : t - 1) : (handover ? r.b + 1 : r.b + 1 + t));
  const int64_t k0 = (int64_t)kb * TS_NB;
  d.r_is_u = r.fwd && t == 0;
  d.ldr = d.r_is_u ? a.ldu : a.ldv;
  d.S = r.S + k0 * r.lds;
  d.s_bytes = r.valid ? clamp_range((r.kend - k0) * r.lds * 8) : 0;
  const bool global_rows = r.valid && !handover;
  d.R = global_rows ? (d.r_is_u ? a.U : a.V) + k0 * d.ldr : nullptr;
  d.r_bytes = global_rows ? clamp_range((r.kend - k0) * d.ldr * 8) : 0;
  return d;
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains the
// wave's global loads and stores (vmcnt(0)): used once per block row, at its first tile, where it makes the previous
// row's stores to V visible to the whole workgroup long before anyone loads them.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// V = L^-T L^-1 U for one strip of TS_SC columns (fwd_only: V = L^-1 U).  8 waves; wave w owns rows [16 w, 16 w + 16) of
// every block row x the strip's 32 columns (one A fragment, two B fragments, two MFMAs per k-quad).
// VEC (workgroup-uniform, chosen by the caller): a full strip -> one 16-byte load per R pair.
template <bool VEC>
__device__ __forceinline__ void strip_solve(const StripArgs &a, int fwd_only, int64_t j0, double *lds) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NQ = TS_NQ, NSUB = TS_NSUB;
  static_assert(TS_BK == 32 && TS_RK == 128 && NSUB == 4 && TS_SC == 32, "tile geometry is wired into the pipeline below");
  const int64_t m = a.m, j = a.j;
  const int nb = (int)((m + TS_NB - 1) / TS_NB);
  const int nrows = fwd_only ? nb : 2 * nb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c16 = lane & 15;
  double *const Rs = lds;                 // [2][TS_TILE]
  double *const Hs = lds + 2 * TS_TILE;   // [2][TS_TILE]
  // R tile staging: 4 passes of 32 rows x 16 threads x 2 doubles
  const int rrow = tid >> 4, lcr = (tid & 15) * 2;
  const int64_t cj0 = j0 + lcr;
  const int64_t cr0 = (cj0 < j) ? cj0 : 0, cr1 = (cj0 + 1 < j) ? cj0 + 1 : 0;  // (a column past the edge is clamped, its results dropped)
  const int v0u = (int)((rrow * a.ldu + cr0) * 8), v1u = (int)((rrow * a.ldu + cr1) * 8);
  const int v0v = (int)((rrow * a.ldv + cr0) * 8), v1v = (int)((rrow * a.ldv + cr1) * 8);
  const int st_off = rrow * TS_SC + (lcr ^ ((rrow & 1) << 4));  // + p * 32 rows: the row parity does not change
  // B fragments: rows 4 kq + q of a sub-step, columns c16 and 16 + c16 (swapped in odd rows)
  const int bo0 = q * TS_SC + ((q & 1) << 4) + c16, bo1 = q * TS_SC + (16 ^ ((q & 1) << 4)) + c16;

  double afr[NSUB][NQ];  // A fragments of sub-step s of the tile in flight (requested one sub-step ahead)
  double bq[2][NQ][2];
  double2v rreg[NSUB];
  double4v acc[2];
  acc[0] = double4v{0.0, 0.0, 0.0, 0.0};
  acc[1] = double4v{0.0, 0.0, 0.0, 0.0};

  auto a_voff = [&](const TileDesc &d) {  // lane offset of the A fragments of a block row: S[.. + q][i0 + 16 w + c16]
    const int64_t ca = (d.i0 + wave * 16 + c16 < m) ? d.i0 + wave * 16 + c16 : 0;
    return (int)((q * d.lds + ca) * 8);
  };
  // ---- operand requests, one PAIR of k-quads at a time (the sub-step interleaves them with its MFMAs) ----
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  const int dma_col = (2 * (lane & 15)) ^ (((lane >> 4) & 1) << 4);
  const int dma_u = (int)(((lane >> 4) * a.ldu + j0 + dma_col) * 8), dma_v = (int)(((lane >> 4) * a.ldv + j0 + dma_col) * 8);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // A fragments of k-quads 2 p, 2 p + 1 of sub-step `sub` of tile d -> afr[sub]
  auto load_a_pair = [&](const TileDesc &d, int voff, auto sub_tag, int p) {
    constexpr int sub = decltype(sub_tag)::value;
    const __amdgpu_buffer_rsrc_t ra =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(d.S), 0, d.s_bytes, 0x00020000);
    const int row4 = (int)(d.lds * 32);  // bytes per 4 rows of S
    const int sub_off = sub * TS_BK * (int)(d.lds * 8);
#pragma unroll
    for (int kq = 2 * p; kq < 2 * p + 2; ++kq)
      afr[sub][kq] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ra, voff, sub_off + kq * row4, 0));
  };
  // B fragments of the same pair of k-quads of sub-step `sub` of an LDS tile -> bq[buf]
  auto read_b_pair = [&](const double *tile, int sub, auto buf_tag, int p) {
    constexpr int buf = decltype(buf_tag)::value;
    const double *r0 = tile + sub * TS_BK * TS_SC + bo0, *r1 = tile + sub * TS_BK * TS_SC + bo1;
#pragma unroll
    for (int kq = 2 * p; kq < 2 * p + 2; ++kq) {
      bq[buf][kq][0] = r0[kq * 4 * TS_SC];
      bq[buf][kq][1] = r1[kq * 4 * TS_SC];
    }
  };
  // Quarter p of the next tile's rows.  Full strips (VEC) go global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ...
  // lds`): one wave-instruction deposits 1 KiB = 4 rows of the tile, lane l the 16 bytes at position l -- so the lane
  // picks the global columns that BELONG at that position (the swizzle lives in its loop-invariant offset); no staging
  // registers, no ds_write.  Completion needs no wait of its own: loads complete in order, and the A fragments of
  // sub-step 2 are requested after these, so the MFMAs of sub-step 2 (which precede the tile's barrier) cannot start
  // before the rows have landed.  Ragged strips go through registers (store_r after sub-step 1).
  auto load_r_part = [&](const TileDesc &d, double *tile, int p) {
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(d.R ? d.R : a.V), 0, d.r_bytes, 0x00020000);
    if constexpr (VEC) {
      const int dv = d.r_is_u ? dma_u : dma_v;
      const int grp = (int)(d.ldr * 32);  // bytes per 4 rows
      const int g = wave_u + 8 * p;       // 4-row group of the tile
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(tile + g * 4 * TS_SC), 16, dv, g * grp, 0, 0);
    } else {
      const int v0 = d.r_is_u ? v0u : v0v, v1 = d.r_is_u ? v1u : v1v;
      const int pass = (int)(d.ldr * 8 * TS_BK);  // bytes per 32 rows
      rreg[p].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, v0, p * pass, 0));
      rreg[p].y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rr, v1, p * pass, 0));
    }
  };
  auto store_r = [&](double *tile) {
    if constexpr (!VEC) {
#pragma unroll
      for (int p = 0; p < NSUB; ++p) *reinterpret_cast<double2v *>(tile + p * TS_BK * TS_SC + st_off) = rreg[p];
    }
  };
  auto mfma_pair = [&](auto sub_tag, int p) {  // sub-step `sub`: A fragments afr[sub], B fragments bq[sub & 1]
    constexpr int sub = decltype(sub_tag)::value, buf = sub & 1;
#pragma unroll
    for (int kq = 2 * p; kq < 2 * p + 2; ++kq) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[sub][kq], bq[buf][kq][0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[sub][kq], bq[buf][kq][1], acc[1], 0, 0, 0);
    }
  };
  // One sub-step: the 16 MFMAs of sub-step SUB of the tile in flight, and the operand requests for what comes a
  // sub-step later, INTERLEAVED pair by pair (2 A loads, 1 DMA at most, 4 B reads, then 4 MFMAs).  Issued as one burst
  // in front of the MFMAs (all eight waves at once, right after the same barrier) the 64 + 8 memory instructions of a
  // sub-step queue up behind each other at the CU's one address unit while the waves that issued them cannot reach their
  // MFMAs (measured: burst 0.405 / 4.80 ms, pairs 0.389 / 4.57 ms at M = 1024 / 4096; one k-quad at a time was twice as
  // slow); `sched_barrier` pins the interleaving.
  auto substep = [&](auto sub_tag, const TileDesc &da, int voffa, auto asub_tag, const double *tileb, int bsub, auto bbuf_tag,
                     const TileDesc *dr, double *rtile) {
#pragma unroll
    for (int p = 0; p < NQ / 2; ++p) {
      load_a_pair(da, voffa, asub_tag, p);
      if (dr) load_r_part(*dr, rtile, p);
      read_b_pair(tileb, bsub, bbuf_tag, p);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pair(sub_tag, p);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using S3 = std::integral_constant<int, 3>;

  // ---- prime: the first tile of the solve (forward row 0: rows 0 .. 127 of U) ----
  RowState row = make_row(a, nb, nrows, 0);
  TileDesc cur = make_tile(a, row, 0);
  int voff_cur = a_voff(cur);
  // every tile starts from finite content: rows past the end of a contraction are never written by a range-checked
  // load, and what they hold is multiplied by operator rows that read as zero
  for (int e = tid; e < 4 * TS_TILE / 2; e += 512) reinterpret_cast<double2v *>(lds)[e] = double2v{0.0, 0.0};
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NQ / 2; ++p) {
    load_r_part(cur, Rs, p);
    load_a_pair(cur, voff_cur, S0{}, p);
  }
  store_r(Rs);
  __syncthreads();
  const double *curbuf = Rs;
#pragma unroll
  for (int p = 0; p < NQ / 2; ++p) read_b_pair(curbuf, 0, B0{}, p);
  int gcount = 1;  // global tiles staged so far: the next one goes to Rs[gcount & 1]

  for (;;) {
    RowState nrow = row;
    if (cur.last) nrow = make_row(a, nb, nrows, cur.s + 1);
    const TileDesc nxt = make_tile(a, nrow, cur.last ? 0 : cur.t + 1);
    const int voff_nxt = cur.last ? a_voff(nxt) : voff_cur;
    double *const nxt_rs = Rs + (gcount & 1) * TS_TILE;
    const double *nxtbuf = nxt.R ? nxt_rs : Hs + nxt.hbuf * TS_TILE;
    // sub-step 0: also requests the next tile's rows (range 0 if it has none)
    substep(S0{}, cur, voff_cur, S1{}, curbuf, 1, B1{}, &nxt, nxt_rs);
    substep(S1{}, cur, voff_cur, S2{}, curbuf, 2, B0{}, nullptr, nullptr);
    store_r(nxt_rs);
    substep(S2{}, cur, voff_cur, S3{}, curbuf, 3, B1{}, nullptr, nullptr);
    if (cur.first)
      __syncthreads();  // once per block row: also completes this wave's stores of the previous row's block
    else
      lds_barrier();
    // sub-step 3 fetches the next tile's first operands (a hand-over tile that this row is about to write is read again below)
    substep(S3{}, nxt, voff_nxt, S0{}, nxtbuf, 0, B0{}, nullptr, nullptr);
    if (cur.last) {  // the block row is complete: result -> global V and the hand-over tile
      double *const hand = Hs + (cur.s & 1) * TS_TILE;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lrow = wave * 16 + 4 * r + q;
          const int64_t row = cur.i0 + lrow, col = j0 + tb * 16 + c16;
          const bool rin = row < m;
          const double v = rin ? acc[tb][r] : 0.0;  // (rows past the matrix must be ZERO in the hand-over: 0 * garbage)
          hand[lrow * TS_SC + ((tb * 16) ^ ((q & 1) << 4)) + c16] = v;
          if (rin && col < j) a.V[row * a.ldv + col] = v;
        }
      acc[0] = double4v{0.0, 0.0, 0.0, 0.0};
      acc[1] = double4v{0.0, 0.0, 0.0, 0.0};
      if (nxt.valid && nxt.R == nullptr) {  // the turn: the next row starts with the block just written
        lds_barrier();
#pragma unroll
        for (int p = 0; p < NQ / 2; ++p) read_b_pair(nxtbuf, 0, B0{}, p);
      }
    }
    if (!nxt.valid) break;
    if (nxt.R) ++gcount;
    cur = nxt;
    row = nrow;
    voff_cur = voff_nxt;
    curbuf = nxtbuf;
  }
#else
  (void)a, (void)fwd_only, (void)j0, (void)lds;
#endif
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

// out (cols x rows) = in^T, 32 x 32 tiles through LDS (both sides coalesced)
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in, int64_t ldi, double *__restrict__ out,
                                                        int64_t ldo, int64_t rows, int64_t cols) {
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int k = ty; k < 32; k += 8)
    if (r0 + k < rows && c0 + tx < cols) t[k][tx] = in[(r0 + k) * ldi + c0 + tx];
  __syncthreads();
  for (int k = ty; k < 32; k += 8)
    if (c0 + k < cols && r0 + tx < rows) out[(c0 + k) * ldo + r0 + tx] = t[tx][k];
}

// out = alpha * in + diag * I  (m x m); in may be NULL (out = diag * I)
__global__ __launch_bounds__(256) void scale_add_diag_kernel(const double *__restrict__ in, int64_t ldi, double alpha, double diag,
                                                             double *__restrict__ out, int64_t ldo, int64_t m) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= m) return;
  for (int64_t row = blockIdx.y; row < m; row += gridDim.y) {
    const double v = in ? alpha * in[row * ldi + col] : 0.0;
    out[row * ldo + col] = (row == col) ? v + diag : v;
  }
}

int launch_transpose(const double *in, int64_t ldi, double *out, int64_t ldo, int64_t rows, int64_t cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return PLS_OK;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(256), 0, st, in, ldi, out,
                     ldo, rows, cols);
  return check_launch("transpose");
}

int launch_scale_add_diag(const double *in, int64_t ldi, double alpha, double diag, double *out, int64_t ldo, int64_t m,
                          hipStream_t st) {
  if (m <= 0) return PLS_OK;
  const unsigned gy = (unsigned)(m < 1024 ? m : 1024);
  hipLaunchKernelGGL(scale_add_diag_kernel, dim3((unsigned)cdiv(m, 256), gy), dim3(256), 0, st, in, ldi, alpha, diag, out, ldo, m);
  return check_launch("scale_add_diag");
}

static size_t strip_lds_bytes() { return (size_t)4 * TS_TILE * sizeof(double); }

int chol_solve_launch(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv,
                      int fwd_only, hipStream_t st) {
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(tri_solve_strip_kernel), strip_lds_bytes(), lds_ready)) return rc;
  StripArgs a{f->Sf, f->Sb, f->ldsf, f->ldsb, U, ldu, V, ldv, f->m, j};
  {
    LaunchScope scope(PLS_TAG_TRI_SOLVE, st);
    hipLaunchKernelGGL(tri_solve_strip_kernel, dim3((unsigned)cdiv(j, TS_SC)), dim3(512), strip_lds_bytes(), st, a, fwd_only);
  }
  return check_launch("tri_solve_strip");
}

// Y = Lc^-1 U.  With the inverse factor in the descriptor (pls_chol_build_inverse) and solve mode 1 this is ONE
// triangular product on the MFMA contraction -- any number of columns fills the chip (a narrow J-shard gives the strip
// kernel cdiv(j, 32) workgroups for 256 CUs); otherwise block forward substitution (strip kernel).
int chol_forward_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *Y, int64_t ldy, hipStream_t st) {
  if (f->LinvT && solve_mode() != 0)
    return gemm_tn_ex(f->LinvT, f->ldlinvt, U, ldu, Y, ldy, f->m, j, f->m, 1.0, 0.0, 1, st, f->tri_scratch, f->tri_scratch_bytes);
  if (!f->Sf || !f->Sb) return fail(PLS_ERR_INVALID_ARGUMENT, "forward solve: the factor has neither substitution operators nor its inverse");
  return chol_solve_launch(f, U, ldu, j, Y, ldy, 1, st);
}

// V = Lc^-T Lc^-1 U; tmp (m x j, leading dimension j) is only needed by the two-product form
int chol_full_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, double *tmp,
                    hipStream_t st) {
  if (tmp && f->Linv && f->LinvT && solve_mode() != 0) {
    int rc = gemm_tn_ex(f->LinvT, f->ldlinvt, U, ldu, tmp, j, f->m, j, f->m, 1.0, 0.0, 1, st, f->tri_scratch,
                        f->tri_scratch_bytes);  // y = Linv u: k <= row
    if (rc) return rc;
    return gemm_tn_ex(f->Linv, f->ldlinv, tmp, j, V, ldv, f->m, j, f->m, 1.0, 0.0, 2, st, f->tri_scratch,
                      f->tri_scratch_bytes);  // v = Linv^T y: k >= row
  }
  if (!f->Sf || !f->Sb) return fail(PLS_ERR_INVALID_ARGUMENT, "solve: the factor has neither substitution operators nor its inverse");
  return chol_solve_launch(f, U, ldu, j, V, ldv, 0, st);
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
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(64), 0, st, Lc, ldlc, LcT, ldlct, k0, nb, info);
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
  const size_t inv_lds = (size_t)(TS_NB * TS_NB + TS_NB) * sizeof(double);
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc2 = ensure_dynamic_lds(reinterpret_cast<const void *>(tri_block_inverse_kernel), inv_lds, lds_ready)) return rc2;
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

int pls_chol_build_inverse(const pls_chol_desc *f, double *Linv, int64_t ldlinv, double *LinvT, int64_t ldlinvt, void *stream) {
  PLS_REQUIRE(f && f->Sf && f->Sb && f->m > 0 && f->ldsf >= f->m && f->ldsb >= f->m, "chol_build_inverse: bad factor descriptor");
  PLS_REQUIRE(Linv && LinvT && Linv != LinvT && ldlinv >= f->m && ldlinvt >= f->m, "chol_build_inverse: bad arguments");
  PLS_REQUIRE(((ldlinv | ldlinvt) & 1) == 0 && ((reinterpret_cast<uintptr_t>(Linv) | reinterpret_cast<uintptr_t>(LinvT)) & 15) == 0,
              "chol_build_inverse: outputs must be 16-byte aligned with even leading dimensions");
  hipStream_t st = S(stream);
  // the identity (in LinvT) as the right-hand side of a forward substitution: column c of Lc^-1 solves Lc x = e_c
  int rc = launch_scale_add_diag(nullptr, 0, 0.0, 1.0, LinvT, ldlinvt, f->m, st);
  if (rc) return rc;
  rc = chol_solve_launch(f, LinvT, ldlinvt, f->m, Linv, ldlinv, 1, st);
  if (rc) return rc;
  return launch_transpose(Linv, ldlinv, LinvT, ldlinvt, f->m, f->m, st);
}

int pls_chol_forward_solve(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *Y, int64_t ldy, void *stream) {
  PLS_REQUIRE(f && f->m > 0, "chol_forward_solve: bad factor descriptor");
  PLS_REQUIRE(U && Y && j >= 0 && ldu >= j && ldy >= j && U != Y, "chol_forward_solve: bad arguments");
  if (j == 0) return PLS_OK;
  return chol_forward_solve(f, U, ldu, j, Y, ldy, S(stream));
}

size_t pls_chol_solve_workspace_bytes(int64_t m, int64_t j) { return (m > 0 && j > 0) ? (size_t)m * j * sizeof(double) : 0; }

int pls_chol_solve_ws(const pls_chol_desc *f, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv, void *workspace,
                      size_t workspace_bytes, void *stream) {
  PLS_REQUIRE(f && f->m > 0, "chol_solve_ws: bad factor descriptor");
  PLS_REQUIRE(U && V && j >= 0 && ldu >= j && ldv >= j && U != V, "chol_solve_ws: bad arguments");
  if (j == 0) return PLS_OK;
  double *tmp = (workspace && workspace_bytes >= pls_chol_solve_workspace_bytes(f->m, j)) ? static_cast<double *>(workspace) : nullptr;
  return chol_full_solve(f, U, ldu, j, V, ldv, tmp, S(stream));
}

int pls_tri_multiply(const double *LcT, int64_t ldlct, int64_t m, const double *X, int64_t ldx, int64_t j, double *out,
                     int64_t ldo, void *stream) {
  PLS_REQUIRE(LcT && X && out && m > 0 && j >= 0 && ldlct >= m && ldx >= j && ldo >= j, "tri_multiply: bad arguments");
  if (j == 0) return PLS_OK;
  return gemm_tn_ex(LcT, ldlct, X, ldx, out, ldo, m, j, m, 1.0, 0.0, 1, S(stream));
}

}  // extern "C"
