// C (I x J) = alpha * L^T R + beta * C for an output whose row count I is NOT a multiple of the 128-row tile:
// the back-projection D = A G of a basis with 129 .. 255, 257 .. 383, ... functions (reference:
// basis/orthonormal.py:152-155; every eigenvalue threshold leaves such a rank, orthonormal.py:51-60).
//
// gemm_tn_f64_kernel<128, 128> would contract 256 rows for 129; cutting the rank into 128 + 64 + 32 + 16-row launches
// (round 2) reads G once per piece and runs the pieces on the slow small configurations.  Here the I axis is cut into
// cdiv(I, 128) tiles of EQUAL height, 16 * cdiv(cdiv(I, 16), tiles) rows, and inside a tile the 16-row blocks of the
// 128-column LDS image that the tile height covers are shared out so that all four waves issue the SAME number of
// MFMAs: an even count 2 n as n blocks per wave row (block b -> wave row b & 1) against 64 columns each, an odd count r as
// all r blocks in every wave against 32 columns each (a split by rows would leave one wave row a block short, and the
// k-step takes as long as its slowest wave: measured, 208 rows = 7 + 6 blocks cost what 256 rows do).  The block count
// is a compile-time constant of the k-loop instantiation a launch-uniform switch selects, so the MFMA count follows
// tiles * cdiv(cdiv(I, 16), tiles), the loads (LDS-DMA, the unchanged 128-wide k-rows) and the barriers are those of the
// full tile, and G is read once.  EVERY tile contracts the same number of blocks, also a last tile that holds fewer rows
// (its surplus blocks multiply columns of the next tile or column 0 and are dropped by the store): tiles of different
// duration measured SLOWER than all tiles at the larger count (176 rows as 6 + 5 blocks 4.78 ms, 192 rows as 6 + 6
// 4.52 ms; 240 as 8 + 7 5.90 ms, 256 5.62 ms, N = 1e5, J = 8192) -- the tiles of one column of G drift apart and stop
// meeting in L2.
//
// Restrictions (the launcher falls back to gemm_tn_f64_kernel otherwise): both operands 16-byte aligned with even
// leading dimensions, 128 <= ldc < kDirectMaxLd (the epilogue is the direct one: registers -> global in the MFMA layout
// through a buffer descriptor whose range ends with the tile's last row, so rows of the next tile or beyond I are
// dropped by the address check, not by a predicate; columns >= J by a lane offset outside every range).
#pragma once
#include "gemm_tn_f64.h"

namespace plship {

struct RowsStore {  // EpiStore's fields; slab `split` of C for split-K
  double *C0;
  int64_t ldc;
  double alpha, beta;
  int64_t slab;
};

// Registers -> global in the MFMA layout: block (ta, tb) of the wave sits 16 * RS * ta rows and 16 tb columns from the
// wave's corner; rows >= iend fall outside the descriptor and are dropped by the address check.
// A lane whose column is >= J gets an offset beyond any descriptor range (its loads read 0, its stores are dropped).
template <int NA, int NBJ, int RS, int TI, int TJ>
__device__ __forceinline__ void rows_store(const AccFrag<TI, TJ> &acc, const RowsStore &e, int64_t rbase, int64_t cbase,
                                           int64_t iend, int64_t J, int split) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x & 63;
  if (rbase >= iend) return;  // (wave-uniform: a wave row whose first block already lies below the tile's last row)
  double *corner = e.C0 + (int64_t)split * e.slab + rbase * e.ldc + cbase;
  // (iend - 1 - rbase) * ldc + 63 < (iend - rbase) * ldc since ldc >= 128: every row below iend is inside, none beyond
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(corner, 0, (int)((iend - rbase) * e.ldc * 8), 0x00020000);
  const int voff = (int)(((int64_t)(lane >> 4) * e.ldc + (lane & 15)) * 8);
  int voffs[NBJ];
#pragma unroll
  for (int tb = 0; tb < NBJ; ++tb) voffs[tb] = (cbase + 16 * tb + (lane & 15) < J) ? voff : (int)0x80000000;
  const int ld4 = (int)(e.ldc * 32);  // bytes per 4 rows
  const bool plain = e.beta == 0.0 && e.alpha == 1.0, nobeta = e.beta == 0.0;
#pragma unroll
  for (int ta = 0; ta < NA; ++ta)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int tb = 0; tb < NBJ; ++tb) {
        const int soff = (4 * RS * ta + q) * ld4 + tb * 128;
        double v = acc.v[ta][tb][q];
        if (!plain) {
          v *= e.alpha;
          if (!nobeta) v += e.beta * __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voffs[tb], soff, 0));
        }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rs, voffs[tb], soff, 0);
      }
#else
  (void)acc, (void)e, (void)rbase, (void)cbase, (void)iend, (void)J, (void)split;
#endif
}

// An even number 2 NB of row blocks: wave row r (waves 2r, 2r + 1) contracts blocks r, r + 2, ... against 64 columns
template <int NB, bool EDGE>
__device__ __forceinline__ void rows_even(const GemmShape &g, const RowsStore &e, int64_t i0, int64_t j0, int64_t iend, int wave,
                                          int split, double *lds) {
  AccFrag<4, 4> acc;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int r = wave >> 1, wc = wave & 1;
  gemm_tn_mainloop<128, 128, 64, 64, 16, true, EDGE, true, NB, 32>(g, i0, j0, lds, acc, 16 * r);
  rows_store<NB, 4, 2>(acc, e, i0 + 16 * r, j0 + 64 * wc, iend, g.J, split);
}

// An odd number RT of row blocks: every wave contracts all of them against its own 32 columns -- the same MFMA count in
// all four waves (2 RT per k-quad), where a split by rows would leave one wave row a block short at every barrier
template <int RT, bool EDGE>
__device__ __forceinline__ void rows_odd(const GemmShape &g, const RowsStore &e, int64_t i0, int64_t j0, int64_t iend, int wave,
                                         int split, double *lds) {
  AccFrag<8, 2> acc;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  gemm_tn_mainloop<128, 128, 128, 32, 16, true, EDGE, true, RT, 16>(g, i0, j0, lds, acc, 0);
  rows_store<RT, 2, 1>(acc, e, i0, j0 + 32 * wave, iend, g.J, split);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_f64_rows_kernel(GemmShape g, int tile_rows, RowsStore e) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  int tile_i, tile_j;
  gemm_tile_coords(blockIdx.x, g.nti, g.ntj, tile_i, tile_j);
  const int64_t i0 = (int64_t)tile_i * tile_rows, j0 = (int64_t)tile_j * 128;
  const int split = blockIdx.y;
  if (gridDim.y > 1) {  // split-K slab: this block's k-range
    const int64_t k0 = (int64_t)split * g.kchunk;
    g.L += k0 * g.ldl;
    g.R += k0 * g.ldr;
    g.K = (g.K - k0 < g.kchunk) ? g.K - k0 : g.kchunk;
  }
  const int64_t iend = (i0 + tile_rows < g.I) ? i0 + tile_rows : g.I;
  const int rt = tile_rows >> 4;  // 16-row blocks every tile contracts (1 .. 8); the last tile may hold fewer rows
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // the 128-wide k-rows of L or R overhang the matrix: those lanes fetch column 0 of their row instead
  const bool edge = i0 + 128 > g.I || j0 + 128 > g.J;
#define PLS_ROWS_CASE(N, FN, ARG)                                  \
  case N:                                                          \
    if (edge)                                                      \
      FN<ARG, true>(g, e, i0, j0, iend, wave, split, lds);         \
    else                                                           \
      FN<ARG, false>(g, e, i0, j0, iend, wave, split, lds);        \
    break;
  switch (rt) {
    PLS_ROWS_CASE(8, rows_even, 4)
    PLS_ROWS_CASE(7, rows_odd, 7)
    PLS_ROWS_CASE(6, rows_even, 3)
    PLS_ROWS_CASE(5, rows_odd, 5)
    PLS_ROWS_CASE(4, rows_even, 2)
    PLS_ROWS_CASE(3, rows_odd, 3)
    PLS_ROWS_CASE(2, rows_even, 1)
    default:
      PLS_ROWS_CASE(1, rows_odd, 1)
  }
#undef PLS_ROWS_CASE
}

}  // namespace plship
