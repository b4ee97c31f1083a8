// fp64 MFMA contraction  C(I x J) = epilogue( sum_k L[k][i] * R[k][j] ),  L (K x I), R (K x J) row-major.
//
// Every dense product on the Langevin path is this one shape ("TN": both operands are stored k-major,
// i.e. each k-row is contiguous along the output dimension), see DESIGN.md "data layout":
//   F = A^T U            L = A  (Mk x N),  R = U (Mk x J)      reference: basis/orthonormal.py:106-108
//   D = A G              L = At (N x Mk),  R = G (N x J)       reference: basis/orthonormal.py:152-155
//   B U (Gaussian path)  L = B  (Mk x Mk), R = U               the O(J M^2) contraction of README.md:9
//
// gfx950 mapping
//   * v_mfma_f64_16x16x4_f64: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15] (one f64 each);
//     the 16x16 result sits as 4 f64 per lane: C[i = (l>>4) + 4*reg][j = l&15].
//   * a 64-lane wave owns a (16*TI) x (16*TJ) block of C; per 4-deep k-quad it reads TI + TJ operand
//     registers from LDS (ds_read_b64, 16 consecutive doubles per 16 lanes) and issues TI*TJ MFMAs.
//   * LDS tiles are [BK][BI + 16] / [BK][BJ + 16] doubles: the +16 pad (128 B) shifts consecutive k-rows by half a
//     256-B bank row, so the two 16-lane groups a ds_read_b64 services together never collide.
//   * global -> LDS: each k-row of a tile is BI*8 bytes, loaded as 16-B vectors (coalesced 1 KiB per wave-row),
//     staged through registers one k-step ahead of the MFMAs (double-buffered LDS, one barrier per k-step).
//   * blockIdx -> tile: XCD-aware remap (blocks b and b+8 share an XCD and its L2) followed by a grouped raster
//     (8 i-tiles x all j) so that the ~64 blocks resident on one XCD share 8 L panels and 8 R panels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plship {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

struct GemmShape {
  const double *L;
  int64_t ldl;
  const double *R;
  int64_t ldr;
  int64_t I, J, K;
  int nti, ntj;
};

// blockIdx.x -> (tile_i, tile_j)
__device__ inline void gemm_tile_coords(int bid, int nti, int ntj, int &ti, int &tj) {
  const int nwg = nti * ntj;
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int id = base + (bid >> 3);
  const int GI = 8;
  const int group = GI * ntj;
  const int g = id / group;
  const int first_i = g * GI;
  const int gi = (nti - first_i < GI) ? (nti - first_i) : GI;
  const int in_g = id - g * group;
  ti = first_i + in_g % gi;
  tj = in_g / gi;
}

// Accumulator fragment owner: element (ti, tj, r) of lane l is C[i0 + 16*ti + 4*r + (l>>4)][j0 + 16*tj + (l&15)].
template <int TI, int TJ>
struct AccFrag {
  double4_t v[TI][TJ];
};

template <int BI, int BJ, int WI, int WJ, int BK, class Epilogue>
__global__ __launch_bounds__((BI / WI) * (BJ / WJ) * 64) void gemm_tn_f64_kernel(GemmShape g, Epilogue epi) {
  constexpr int NW = (BI / WI) * (BJ / WJ);
  constexpr int NT = NW * 64;
  constexpr int TI = WI / 16, TJ = WJ / 16;
  constexpr int PAD = 16;
  constexpr int SL = BI + PAD, SR = BJ + PAD;
  constexpr int LROWS = NT / (BI / 2);  // k-rows of the L tile covered by one pass of all threads
  constexpr int RROWS = NT / (BJ / 2);
  constexpr int LPASS = BK / LROWS, RPASS = BK / RROWS;
  static_assert(BK % LROWS == 0 && BK % RROWS == 0 && BK % 4 == 0, "tile/thread mismatch");

  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *Ls = lds;                 // [2][BK][SL]
  double *Rs = lds + 2 * BK * SL;   // [2][BK][SR]

  int tile_i, tile_j;
  gemm_tile_coords(blockIdx.x, g.nti, g.ntj, tile_i, tile_j);
  const int64_t i0 = (int64_t)tile_i * BI, j0 = (int64_t)tile_j * BJ;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = (wave / (BJ / WJ)) * WI, wj = (wave % (BJ / WJ)) * WJ;
  const int q = lane >> 4, c16 = lane & 15;

  // global-load coordinates of this thread
  const int lcol = (tid % (BI / 2)) * 2, lrow = tid / (BI / 2);
  const int rcol = (tid % (BJ / 2)) * 2, rrow = tid / (BJ / 2);
  const bool lvec = ((g.ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.L) & 15) == 0);
  const bool rvec = ((g.ldr & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
  const double *Lp = g.L + i0 + lcol;
  const double *Rp = g.R + j0 + rcol;
  const int64_t lrem = g.I - i0 - lcol;  // valid columns from this thread's first column
  const int64_t rrem = g.J - j0 - rcol;

  double2_t lreg[LPASS], rreg[RPASS];

  auto load_global = [&](int64_t k0) {
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int64_t k = k0 + lrow + p * LROWS;
      double2_t v = {0.0, 0.0};
      if (k < g.K) {
        const double *src = Lp + k * g.ldl;
        if (lvec && lrem >= 2) {
          v = *reinterpret_cast<const double2_t *>(src);
        } else {
          if (lrem >= 1) v.x = src[0];
          if (lrem >= 2) v.y = src[1];
        }
      }
      lreg[p] = v;
    }
#pragma unroll
    for (int p = 0; p < RPASS; ++p) {
      const int64_t k = k0 + rrow + p * RROWS;
      double2_t v = {0.0, 0.0};
      if (k < g.K) {
        const double *src = Rp + k * g.ldr;
        if (rvec && rrem >= 2) {
          v = *reinterpret_cast<const double2_t *>(src);
        } else {
          if (rrem >= 1) v.x = src[0];
          if (rrem >= 2) v.y = src[1];
        }
      }
      rreg[p] = v;
    }
  };
  auto store_lds = [&](int buf) {
    double *l = Ls + buf * BK * SL;
    double *r = Rs + buf * BK * SR;
#pragma unroll
    for (int p = 0; p < LPASS; ++p)
      *reinterpret_cast<double2_t *>(l + (lrow + p * LROWS) * SL + lcol) = lreg[p];
#pragma unroll
    for (int p = 0; p < RPASS; ++p)
      *reinterpret_cast<double2_t *>(r + (rrow + p * RROWS) * SR + rcol) = rreg[p];
  };

  AccFrag<TI, TJ> acc;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

  const int64_t nk = (g.K + BK - 1) / BK;
  load_global(0);
  store_lds(0);
  __syncthreads();

  for (int64_t kt = 0; kt < nk; ++kt) {
    const int buf = (int)(kt & 1);
    if (kt + 1 < nk) load_global((kt + 1) * BK);
    const double *l = Ls + buf * BK * SL + q * SL + wi + c16;
    const double *r = Rs + buf * BK * SR + q * SR + wj + c16;
#pragma unroll
    for (int kq = 0; kq < BK / 4; ++kq) {
      double a[TI], b[TJ];
#pragma unroll
      for (int t = 0; t < TI; ++t) a[t] = l[kq * 4 * SL + t * 16];
#pragma unroll
      for (int t = 0; t < TJ; ++t) b[t] = r[kq * 4 * SR + t * 16];
#pragma unroll
      for (int ta = 0; ta < TI; ++ta)
#pragma unroll
        for (int tb = 0; tb < TJ; ++tb)
          acc.v[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc.v[ta][tb], 0, 0, 0);
    }
    if (kt + 1 < nk) store_lds(buf ^ 1);
    __syncthreads();
  }

  epi.template apply<TI, TJ>(acc, i0 + wi, j0 + wj, lane, wave, g.I, g.J, tile_i, lds);
}

// ---- epilogues ------------------------------------------------------------------------------------------------
// apply(acc, iw, jw, lane, wave, I, J, tile_i, lds): iw/jw = global coordinates of the wave's block of C.

#define PLS_FOR_EACH_ACC(BODY)                                         \
  _Pragma("unroll") for (int ta = 0; ta < TI; ++ta)                    \
      _Pragma("unroll") for (int tb = 0; tb < TJ; ++tb)                \
          _Pragma("unroll") for (int r = 0; r < 4; ++r) {              \
    const int64_t i = iw + ta * 16 + r * 4 + (lane >> 4);             \
    const int64_t j = jw + tb * 16 + (lane & 15);                     \
    const double v = acc.v[ta][tb][r];                                 \
    if (i < I && j < J) { BODY }                                       \
  }

struct EpiStore {  // C = alpha * acc + beta * C
  double *C;
  int64_t ldc;
  double alpha, beta;
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int, int64_t I, int64_t J, int,
                        double *) const {
    if (beta == 0.0) {
      PLS_FOR_EACH_ACC(C[i * ldc + j] = alpha * v;)
    } else {
      PLS_FOR_EACH_ACC(C[i * ldc + j] = alpha * v + beta * C[i * ldc + j];)
    }
  }
};

}  // namespace plship
