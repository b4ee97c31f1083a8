// Fused drift kernel for projection ranks 129 .. 256: F -> d cost / d f -> back-projection in ONE pass, like small_rank.h,
// with the basis functions split over wave PAIRS.
//
// Why: between 129 and ~250 functions the two-GEMM step is bound by the N x J matrix G it writes once and reads twice
// (main 128-row pass + remainder pass of the back-projection: 19.7 GB at N = 1e5, J = 8192 against 8.5e11 flop; 0.71-0.81
// of peak, profiles/r02_step_sweep_ranks.txt), and small_rank.h stops at 128 because a wave keeps its K x 16 particle block
// AND its K x 16 drift block in registers.  Here a workgroup still owns 64 particle columns and a slab of data rows, but has
// 8 waves: waves w and w + 4 share the 16 columns `w & 3`; wave half h = w >> 2 keeps rows [0, 16 KB0) (h = 0) or
// [16 KB0, 16 (KB0 + KB1)) (h = 1) of the particle block and of the drift block -- the register budget of small_rank.h at
// rank <= 128.  Per 16-row tile of Lb (N x K row-major, streamed through LDS, three buffers):
//   * each wave contracts ITS half of the rank: a partial 16 x 16 block of F (4 KBh MFMAs, one accumulator chain);
//   * the partials are exchanged through LDS (4 doubles per lane), F = F_0 + F_1 in that order in both waves, and both
//     apply the cost derivative to the same four accumulator registers (redundant VALU, no second exchange);
//   * those registers are the B operands of the back-projection of the wave's half, D_h += Lb_tile[:, half]^T G
//     (4 KBh MFMAs into KBh accumulator blocks), whose A operands are the same LDS tile read the other way round.
// The exchange is software-pipelined: iteration t contracts F of tile t + 1, publishes it, THEN back-projects tile t, and
// only then crosses the iteration's single barrier -- the partner's partial sum has been in LDS for 4 KBh MFMAs by the
// time it is read.  The loop body is branch-free (tiles past the slab are read through zero-range descriptors and masked
// in the cost), so every accumulator is updated in place (tools/mfma_srcc_lint.py).
// Reference: projected_langevin_sampling.py:107-123 + basis/orthonormal.py:106-108, :128-159, costs/{*}.py.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cost_device.h"
#include "small_rank.h"

namespace plship {

constexpr int SR2_ROWS = 16;  // rows of Lb per LDS tile (one MFMA block)

template <int KBT>
constexpr int sr2_stride() { return 16 * KBT + 2; }
template <int KBT>
constexpr size_t sr2_lds_bytes() {
  return (size_t)(3 * SR2_ROWS * sr2_stride<KBT>() + 3 * SR2_ROWS + 2 * 8 * 256) * sizeof(double);
}

// MODE: SR_MODE_DRIFT or SR_MODE_DRIFT_VALUE (small_rank.h).  SmallRankP as there; K in (16 (KB0 + KB1 - 1), 16 (KB0 + KB1)].
template <int KB0, int KB1, int MODE, int COST, int LINK>
__global__ __launch_bounds__(512, 2) void small_rank2_kernel(SmallRankP p) {
  static_assert(KB0 <= 8 && KB1 <= KB0 && KB1 >= 1, "halves");
  static_assert(MODE == SR_MODE_DRIFT || MODE == SR_MODE_DRIFT_VALUE, "drift kernels only");
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int KBT = KB0 + KB1, KP = 16 * KBT, STR = sr2_stride<KBT>(), TR = SR2_ROWS;
  constexpr int PAIRS = TR * (KP / 2);         // 16-byte pairs per tile
  constexpr int NLOAD = (PAIRS + 511) / 512;   // per thread (the last one partly idle unless KP is a multiple of 64)
  constexpr int TILE = TR * STR;
  extern __shared__ __attribute__((aligned(16))) double sr2_lds[];
  double *const tiles = sr2_lds;
  double *const ys = sr2_lds + 3 * TILE;
  double *const exch = ys + 3 * TR;  // [parity][wave][register][lane]

  CostP cp = p.cp;
  if constexpr (COST >= 0) {
    cp.cost = COST;
    cp.link = LINK;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  const int cg = wave & 3, half = wave >> 2;
  const int64_t jcol = (int64_t)blockIdx.x * 64 + cg * 16 + c;
  const bool jin = jcol < p.J;
  const int split = blockIdx.y;
  const int64_t nbeg = (int64_t)split * p.rows_per_split;
  const int64_t nend = (nbeg + p.rows_per_split < p.N) ? nbeg + p.rows_per_split : p.N;
  const int ntile = (int)((nend - nbeg + TR - 1) / TR);  // >= 1 (the launcher sends no empty slab)

  // ---- staging of one tile: global -> registers -> LDS, pair e = tid + 512 i -> row e / (KP/2), columns 2 (e % (KP/2)) ----
  sr_double2_t stage[NLOAD];
  double ystage = 0.0;
  int voff[NLOAD], soff[NLOAD];
  bool halfpair[NLOAD], live[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int e = tid + 512 * i;
    const int row = e / (KP / 2), m = 2 * (e % (KP / 2));
    live[i] = e < PAIRS;
    voff[i] = (live[i] && m < p.K) ? (int)(((int64_t)row * p.ldlb + m) * 8) : 0x7FFFFF00;
    halfpair[i] = (m + 1 >= p.K);
    soff[i] = row * STR + m;
  }
  const int yoff = (tid < TR) ? tid * 8 : 0x7FFFFF00;
  const int64_t row_bytes = p.ldlb * 8;
  auto load_tile = [&](int t) {  // rows past the slab (and whole tiles past it) read as zero: no branch
    const int64_t n0 = nbeg + (int64_t)t * TR;
    const int64_t left = nend - n0;
    const int64_t lb = left > 0 ? left * row_bytes : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double *>(p.Lb + (left > 0 ? n0 : nbeg) * p.ldlb), 0, (int)(lb < 0x7FFFFF00 ? lb : 0x7FFFFF00), 0x00020000);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i)
      stage[i] = __builtin_bit_cast(sr_double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
    const int64_t yb = left > 0 ? left * 8 : 0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.y + (left > 0 ? n0 : nbeg)), 0,
                                                                        (int)(yb < 0x7FFFFF00 ? yb : 0x7FFFFF00), 0x00020000);
    ystage = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ry, yoff, 0, 0));
  };
  auto store_tile = [&](int buf) {
    double *T = tiles + buf * TILE;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      sr_double2_t v = stage[i];
      v.y = halfpair[i] ? 0.0 : v.y;  // (the pair that straddles an odd K: its second element is row padding)
      if (live[i]) *reinterpret_cast<sr_double2_t *>(T + soff[i]) = v;
    }
    if (tid < TR) ys[buf * TR + tid] = ystage;
  };

  // ---- one half of the rank ----
  auto run = [&](auto kb_tag, int kbase) {
    constexpr int KB = decltype(kb_tag)::value, NQ = 4 * KB;
    double ufrag[NQ];
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {
      const int m = kbase + 4 * kq + q;
      ufrag[kq] = (jin && m < p.K) ? p.V[(int64_t)m * p.ldv + jcol] : 0.0;
    }
    sr_double4_t dacc[KB];
#pragma unroll
    for (int ta = 0; ta < KB; ++ta) dacc[ta] = sr_double4_t{0.0, 0.0, 0.0, 0.0};
    double vsum = 0.0;
    const int o1 = c * STR + kbase + q;  // first contraction: row c, column kbase + 4 kq + q
    const int o2 = q * STR + kbase + c;  // second contraction: row q + 4 r, column kbase + 16 ta + c

    // partial F of the tile in buffer `buf`: one accumulator chain, A fragments fetched a pair of k-quads ahead (the first
    // pair by first_prefetch, right behind the previous barrier); in front of its LAST pair of MFMAs it also fetches the
    // first A fragments of the back-projection of the tile in `bufd`, so that neither contraction starts with an exposed
    // LDS round trip (both waves of a SIMD run this code in lockstep: nobody covers a stall)
    double a_pre[2];
    auto first_prefetch = [&](int buf) {
      const double *T = tiles + buf * TILE + o1;
      a_pre[0] = T[0];
      a_pre[1] = T[4];
    };
    double an[2][KB];
    auto first = [&](int buf, int bufd) {
      const double *T = tiles + buf * TILE + o1;
      const double *T2 = tiles + bufd * TILE + o2;
      sr_double4_t f{0.0, 0.0, 0.0, 0.0};
      double a[2][2];
      a[0][0] = a_pre[0];
      a[0][1] = a_pre[1];
#pragma unroll
      for (int g = 0; g < NQ / 2; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NQ / 2) {
          a[nxt][0] = T[8 * (g + 1)];
          a[nxt][1] = T[8 * (g + 1) + 4];
        } else {
#pragma unroll
          for (int ta = 0; ta < KB; ++ta) an[0][ta] = T2[16 * ta];
        }
        __builtin_amdgcn_sched_barrier(0);
        f = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][0], ufrag[2 * g], f, 0, 0, 0);
        f = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][1], ufrag[2 * g + 1], f, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return f;
    };
    auto publish = [&](int parity, const sr_double4_t &f) {
      double *e = exch + ((parity * 8 + wave) * 4) * 64 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r * 64] = f[r];
    };
    // F = F_0 + F_1, then the cost derivative (and value) per element; register r <-> tile row q + 4 r.  The partner's
    // partial sum and the tile's targets are requested first, then -- behind the same barrier -- the first A fragments of
    // the NEXT partial F (tile in `bufnext`): the per-element code runs while they travel.
    auto finish = [&](int parity, int t, int buf, int bufnext, const sr_double4_t &mine) {
      const double *e = exch + ((parity * 8 + (wave ^ 4)) * 4) * 64 + lane;
      const double *Y = ys + buf * TR;
      const int64_t n0 = nbeg + (int64_t)t * TR;
      double other[4], yr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        other[r] = e[r * 64];
        yr[r] = Y[q + 4 * r];
      }
      first_prefetch(bufnext);
      __builtin_amdgcn_sched_barrier(0);
      sr_double4_t g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double f = half ? other[r] + mine[r] : mine[r] + other[r];  // half 0's partial sum first
        const bool valid = n0 + q + 4 * r < nend;
        if (MODE == SR_MODE_DRIFT_VALUE) {
          const double cval = cost_value(cp, yr[r], f);
          vsum += valid ? cval : 0.0;
        }
        const double gval = cost_deriv(cp, yr[r], f);
        g[r] = valid ? gval : 0.0;
      }
      return g;
    };
    auto second = [&](int buf, const sr_double4_t &g) {  // (its first A fragments, an[0], were fetched by first())
      const double *T = tiles + buf * TILE + o2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r + 1 < 4) {
#pragma unroll
          for (int ta = 0; ta < KB; ++ta) an[(r + 1) & 1][ta] = T[4 * (r + 1) * STR + 16 * ta];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < KB; ++ta) dacc[ta] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[r & 1][ta], g[r], dacc[ta], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // prologue: tiles 0 and 1 staged; G of tile 0
    load_tile(0);
    store_tile(0);
    load_tile(1);
    store_tile(1);
    __syncthreads();
    first_prefetch(0);
    sr_double4_t fcur = first(0, 0);  // (the back-projection fragments it fetches are fetched again by the loop's first())
    publish(0, fcur);
    __syncthreads();
    sr_double4_t g = finish(0, 0, 0, 1, fcur);
    int b0 = 0, b1 = 1, b2 = 2;  // buffers of tiles t, t + 1, t + 2
    for (int t = 0; t < ntile; ++t) {
      load_tile(t + 2);                       // (zeros past the slab)
      const sr_double4_t fn = first(b1, b0);  // partial F of tile t + 1 (+ the first fragments of tile t's back-projection)
      publish((t + 1) & 1, fn);
      second(b0, g);                          // back-projection of tile t
      store_tile(b2);
      __syncthreads();  // partials of tile t + 1 and the rows of tile t + 2 are visible; tile t's buffer is free
      g = finish((t + 1) & 1, t + 1, b1, b2, fn);
      const int tmp = b0;
      b0 = b1;
      b1 = b2;
      b2 = tmp;
    }

    double *D = p.out + (int64_t)split * p.slab_stride;
    if (jin) {
#pragma unroll
      for (int ta = 0; ta < KB; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = kbase + 16 * ta + q + 4 * r;
          if (m < p.K) D[(int64_t)m * p.ldo + jcol] = dacc[ta][r];
        }
    }
    if (MODE == SR_MODE_DRIFT_VALUE) {  // (both halves hold the same sums: half 0 reports)
      vsum += __shfl_xor(vsum, 16);
      vsum += __shfl_xor(vsum, 32);
      if (half == 0 && jin && q == 0) p.vout[(int64_t)split * p.ldvo + jcol] = vsum;
    }
  };
  if (half == 0)
    run(std::integral_constant<int, KB0>{}, 0);
  else
    run(std::integral_constant<int, KB1>{}, 16 * KB0);
#else
  (void)p;
#endif
}

}  // namespace plship
