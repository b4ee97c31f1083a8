// Epilogues of the forward GEMM that turn the F tile into d cost / d f (and optionally the cost value), and the entry
// point of their translation unit (gemm_cost.hip).
#pragma once
#include "cost_device.h"
#include "gemm_tn_f64.h"

namespace plship {

// G = d cost / d f (acc = F tile); rows of this launch are rows [row0, row0 + I) of y.
// vpart (optional): the cost VALUE of the same F as a by-product -- vpart[wave row][j] = sum over the 16*TI rows of the
// wave's block of cost(y_i, F_ij) (wave row = iw / (16 TI); fixed order, no cross-wave traffic): the energy of the
// step's INPUT particles without a third pass over A (projected_langevin_sampling.py:125-138 recomputes F for it).
// COST / LINK >= 0: compile-time cost and link (the per-element code shrinks to the one formula: measured -2.4 % on
// the forward GEMM of a Bernoulli/sigmoid step at M_k = 2048, -5 % with the energy by-product; -1: run-time switch).
template <int COST, int LINK>
struct EpiCostDeriv {
  static constexpr int kTag = PLS_TAG_GEMM_COST_DERIV;
  static constexpr bool kDirect = false;
  double *G;
  int64_t ldg;
  const double *y;
  CostP cp;
  double *vpart;
  int64_t ldp;
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int, int, double *lds) const {
    const double yl = load_row_constants(y, iw, lane, I);
    CostP cp = this->cp;
    if constexpr (COST >= 0) {
      cp.cost = COST;
      cp.link = LINK;
    }
    if (!vpart) {
      epilogue_row_pairs<TI, TJ, 1>(acc, iw, jw, lane, wave, I, J, lds, yl, 0.0,
                                 [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &rc) {
                                   G[i * ldg + j] = cost_deriv(cp, rc.k0_lo, v0);
                                   if (hi) G[(i + 4) * ldg + j] = cost_deriv(cp, rc.k0_hi, v1);
                                 });
      return;
    }
    double s = 0.0;
    epilogue_row_pairs<TI, TJ, 1>(acc, iw, jw, lane, wave, I, J, lds, yl, 0.0,
                               [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &rc) {
                                 G[i * ldg + j] = cost_deriv(cp, rc.k0_lo, v0);
                                 s += cost_value(cp, rc.k0_lo, v0);
                                 if (hi) {
                                   G[(i + 4) * ldg + j] = cost_deriv(cp, rc.k0_hi, v1);
                                   s += cost_value(cp, rc.k0_hi, v1);
                                 }
                               });
    constexpr int WJ = TJ * 16;
    if (WJ == 32) s += __shfl_xor(s, 32);  // two lane halves share the 32 columns
    if (iw < I && lane < WJ && jw + lane < J) vpart[(iw / (16 * TI)) * ldp + jw + lane] = s;
  }
};

// Gaussian cost with the identity link (gaussian.py:86-88): G = (acc - y_i) / sigma2, evaluated as
// fma(acc, 1/sigma2, -y_i/sigma2) in every tile shape, so that the result does not depend on the launch geometry.
// Interior tiles take the direct path.  vpart as in EpiCostDeriv: cost = (acc - y)^2 / (2 sigma2) = G^2 * sigma2 / 2.
struct EpiGaussDeriv {
  static constexpr int kTag = PLS_TAG_GEMM_COST_DERIV;
  static constexpr bool kDirect = true;
  double *G;
  int64_t ldg;
  const double *y;
  double inv_noise;
  double *vpart;
  int64_t ldp;
  __device__ int64_t direct_ld() const { return ldg; }
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int, int, double *lds) const {
    const double yl = load_row_constants(y, iw, lane, I);
    double s = 0.0;
    epilogue_row_pairs<TI, TJ, 1>(acc, iw, jw, lane, wave, I, J, lds, yl, 0.0,
                               [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &rc) {
                                 const double g0 = fma(v0, inv_noise, -inv_noise * rc.k0_lo);
                                 G[i * ldg + j] = g0;
                                 s = fma(g0, g0, s);
                                 if (hi) {
                                   const double g1 = fma(v1, inv_noise, -inv_noise * rc.k0_hi);
                                   G[(i + 4) * ldg + j] = g1;
                                   s = fma(g1, g1, s);
                                 }
                               });
    if (vpart) {
      constexpr int WJ = TJ * 16;
      if (WJ == 32) s += __shfl_xor(s, 32);
      if (iw < I && lane < WJ && jw + lane < J) vpart[(iw / (16 * TI)) * ldp + jw + lane] = s * (0.5 / inv_noise);
    }
  }
  template <int TI, int TJ>
  static constexpr bool direct_tile() { return true; }
  template <int TI, int TJ>
  __device__ void apply_direct(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int, double *) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // y of the 16 row groups this lane's registers belong to: rows iw + 4 s + (lane >> 4), s = 0..4 TI - 1
    const __amdgpu_buffer_rsrc_t ys =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(y + iw), 0, 0x7FFFFFF0, 0x00020000);
    const int yoff = (lane >> 4) * 8;
    double yv[4 * TI];  // -y_i / sigma2: one fma per element, v / sigma2 - y_i / sigma2 (abs. error <= ulp(y / sigma2))
#pragma unroll
    for (int s = 0; s < 4 * TI; ++s)
      yv[s] = -inv_noise * __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ys, yoff, s * 32, 0));
    if (!vpart) {
      epilogue_direct<TI, TJ>(acc, G + iw * ldg + jw, ldg, lane,
                              [&](double v, int slot, int, __amdgpu_buffer_rsrc_t, int, int) { return fma(v, inv_noise, yv[slot]); });
      return;
    }
    double sq[TJ];  // per 16-column block: this lane's 4*TI rows of G^2
#pragma unroll
    for (int tb = 0; tb < TJ; ++tb) sq[tb] = 0.0;
    epilogue_direct<TI, TJ>(acc, G + iw * ldg + jw, ldg, lane, [&](double v, int slot, int tb, __amdgpu_buffer_rsrc_t, int, int) {
      const double g = fma(v, inv_noise, yv[slot]);
      sq[tb] = fma(g, g, sq[tb]);
      return g;
    });
    const double half_s2 = 0.5 / inv_noise;
#pragma unroll
    for (int tb = 0; tb < TJ; ++tb) {
      double t = sq[tb];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);  // the four lane groups hold rows (lane >> 4) + 4 r of the same column
      if (lane < 16) vpart[(iw / (16 * TI)) * ldp + jw + tb * 16 + lane] = t * half_s2;
    }
#else
    (void)acc, (void)iw, (void)jw, (void)lane;
#endif
  }
};

// partial[tile_i][j] = sum over the tile's rows of cost(y_i, acc_ij); deterministic order.
template <int BI, int BJ, int WI, int WJ, int COST = -1, int LINK = -1>
struct EpiCostValue {
  static constexpr int kTag = PLS_TAG_GEMM_COST_VALUE;
  static constexpr bool kDirect = false;
  double *partial;
  int64_t ldp;
  const double *y;
  CostP cp;
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int tile_i, int, double *lds) const {
    double s = 0.0;  // this lane's column, summed over the rows it is handed (fixed order)
    CostP cp = this->cp;
    if constexpr (COST >= 0) {
      cp.cost = COST;
      cp.link = LINK;
    }
    const double yl = load_row_constants(y, iw, lane, I);
    epilogue_row_pairs<TI, TJ, 1>(acc, iw, jw, lane, wave, I, J, lds, yl, 0.0,
                               [&](int64_t, int64_t, double v0, bool hi, double v1, const RowConsts &rc) {
                                 s += cost_value(cp, rc.k0_lo, v0);
                                 if (hi) s += cost_value(cp, rc.k0_hi, v1);
                               });
    if (WJ == 32) s += __shfl_xor(s, 32);  // two lane halves share the 32 columns
    constexpr int NWJ = BJ / WJ, NWI = BI / WI;
    const int wrow = wave / NWJ, wcol = wave % NWJ;
    double *red = lds;  // [NWI][BJ]; overlaps the waves' slabs: wait until every wave has left its row loops
    __syncthreads();
    if (lane < WJ) red[wrow * BJ + wcol * WJ + lane] = s;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BJ) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < NWI; ++w) tot += red[w * BJ + t];
      const int64_t j = (jw - wcol * WJ) + t;
      if (j < J) partial[(int64_t)tile_i * ldp + j] = tot;
    }
  }
};

// Forward GEMM + cost-VALUE epilogue: partial[tile row][j] = sum over the tile's rows of cost(y_i, F_ij); the number of
// partial rows written is cdiv(rows, 128) with the big tiles, cdiv(rows, 64) otherwise (cost_value_partial_rows in
// plship.hip).  Dispatches on (cost, link); defined in gemm_cost_value.hip.
int launch_cost_value_gemm(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                           double *partial, int64_t ldp, const double *y, const CostP &cp, hipStream_t st);

// Forward GEMM + cost-derivative epilogue for rows [0, rows) of Lf / y:  G = cost'(Lf^T V); vpart (optional) receives the
// per-wave-row cost partial sums.  Dispatches on (cost, link); defined in gemm_cost.hip.
int launch_cost_deriv_gemm(const double *Lf, int64_t ldlf, const double *V, int64_t ldv, int64_t rows, int64_t j, int64_t kdim,
                           double *G, int64_t ldg, const double *y, const CostP &cp, double *vpart, int64_t ldp,
                           hipStream_t st);

}  // namespace plship
