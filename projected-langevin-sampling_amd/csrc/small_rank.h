// Fused drift / cost kernels for SMALL projection ranks (K <= 128 basis functions).
//
// With few basis functions the two-GEMM step  F = Lb V  ->  G = cost'(F)  ->  D = Lb^T G  is bound by writing and
// re-reading the N x J matrix G (6.5 GB at N = 5e4, J = 16384: >= 2.6 ms of pure HBM time against 3.7 ms of MFMA
// work), and its GEMMs have k-loops of a handful of steps.  Here G never leaves the registers:
//
//   * a workgroup owns 64 particle columns (4 waves x 16 columns) and a slab of data rows; it streams the rows of
//     Lb (N x K, row-major: `At` of the orthonormal basis, `Kxz` of the inducing-point basis) through LDS in tiles
//     of 32 rows, double-buffered;
//   * the wave's 16 columns of V (K x 16) stay in registers as MFMA B-operands for the whole kernel;
//   * per 16-row block: K/4 MFMAs give the 16x16 block of F in the accumulator layout (register r of lane l = row
//     (l >> 4) + 4 r, column l & 15); the cost derivative is applied to the 4 accumulator registers in place; those
//     registers ARE the B-operands of the second contraction (k index (l >> 4) <-> row (l >> 4) + 4 r for MFMA
//     number r), whose A-operands come from the same LDS tile read the other way round: no transpose, no shuffle;
//   * D (K x 16 per wave) accumulates in registers over the whole slab and is written once; the slabs are summed in
//     a fixed order by the update kernel (deterministic, no atomics).
//
// MODE_VALUE keeps only the first contraction and sums cost(y, F) over the rows instead (energy potential).
// Reference: projected_langevin_sampling.py:107-123 + basis/orthonormal.py:106-108,128-159 (the same three calls the
// two-GEMM path fuses), costs/{*}.py for the per-element functions.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cost_device.h"

namespace plship {

typedef double sr_double4_t __attribute__((ext_vector_type(4)));
typedef double sr_double2_t __attribute__((ext_vector_type(2)));

struct SmallRankP {
  const double *Lb;  // N x K row-major
  int64_t ldlb;
  const double *V;  // K x J
  int64_t ldv;
  const double *y;
  int64_t N, J;
  int K;
  int64_t rows_per_split;  // multiple of 32
  double *out;             // MODE_DRIFT: D slabs [split][K][ldo];  MODE_VALUE: partial sums [split][ldo]
  int64_t ldo, slab_stride;
  CostP cp;
  double *vout;  // SR_MODE_DRIFT_VALUE: cost partial sums [split][ldvo] next to the drift slabs
  int64_t ldvo;
};

// DRIFT: D = Lb^T cost'(Lb V);  VALUE: sum_rows cost(Lb V);  DRIFT_VALUE: both from the same F (the step that also
// reports the energy of its input particles)
constexpr int SR_MODE_DRIFT = 0, SR_MODE_VALUE = 1, SR_MODE_DRIFT_VALUE = 2;
constexpr int SR_ROWS = 32;  // rows of Lb per LDS tile (two 16-row MFMA blocks)

template <int KB>
constexpr int sr_stride() { return 16 * KB + 2; }  // doubles per LDS row: +2 keeps both read patterns conflict-free

template <int KB>
constexpr size_t sr_lds_bytes() { return (size_t)2 * (SR_ROWS * sr_stride<KB>() + SR_ROWS) * sizeof(double); }

// COST / LINK >= 0: the cost and link are compile-time constants (the per-element code shrinks to the one formula and
// its constants: the generic version keeps ~100 registers of polynomial coefficients alive and spills at KB >= 5).
template <int KB, int MODE, int COST, int LINK>
__global__ __launch_bounds__(256, 2) void small_rank_kernel(SmallRankP p) {
  constexpr int KP = 16 * KB;      // padded rank
  constexpr int NQ = 4 * KB;       // k-quads of the first contraction
  constexpr int STR = sr_stride<KB>();
  constexpr int PAIRS = SR_ROWS * (KP / 2);  // 16-byte pairs per tile
  constexpr int NLOAD = PAIRS / 256;         // = KB... per thread
  static_assert(PAIRS % 256 == 0, "tile/thread mismatch");
  extern __shared__ __attribute__((aligned(16))) double sr_lds[];
  auto tile_of = [&](int b) { return sr_lds + b * (SR_ROWS * STR); };
  auto ys_of = [&](int b) { return sr_lds + 2 * SR_ROWS * STR + b * SR_ROWS; };

  CostP cp = p.cp;
  if constexpr (COST >= 0) {
    cp.cost = COST;
    cp.link = LINK;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c = lane & 15;
  const int nq_live = (p.K + 3) >> 2;  // k-quads of the first contraction that hold data
  const int64_t jcol = (int64_t)blockIdx.x * 64 + wave * 16 + c;
  const bool jin = jcol < p.J;
  const int split = blockIdx.y;
  const int64_t nbeg = (int64_t)split * p.rows_per_split;
  const int64_t nend = (nbeg + p.rows_per_split < p.N) ? nbeg + p.rows_per_split : p.N;

  // the wave's particle columns as B-operands: ufrag[kq] = V[4 kq + q][jcol]
  double ufrag[NQ];
#pragma unroll
  for (int kq = 0; kq < NQ; ++kq) {
    const int m = 4 * kq + q;
    ufrag[kq] = (jin && m < p.K) ? p.V[(int64_t)m * p.ldv + jcol] : 0.0;
  }
  sr_double4_t dacc[KB];
#pragma unroll
  for (int ta = 0; ta < KB; ++ta) dacc[ta] = sr_double4_t{0.0, 0.0, 0.0, 0.0};
  double vsum = 0.0;  // MODE_VALUE: this lane's rows of its column

  // global -> register staging of one tile: pair index e = tid + 256 * i -> row e / (KP/2), columns 2 * (e % (KP/2)).
  // Addressing is loop-invariant: a buffer descriptor rebuilt per tile from scalars (base = first row of the tile,
  // num_records = the bytes left in the slab, so rows past the slab read as zero) plus one 32-bit lane offset per
  // pair; pairs entirely past column K get an out-of-range offset (read as zero), the pair straddling an odd K has
  // its second element cleared (it is row padding: possibly not finite, and 0 * NaN would poison F).
  sr_double2_t stage[NLOAD];
  double ystage = 0.0;
  int voff[NLOAD];
  bool half[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int e = tid + 256 * i;
    const int row = e / (KP / 2), m = 2 * (e % (KP / 2));
    voff[i] = (m < p.K) ? (int)(((int64_t)row * p.ldlb + m) * 8) : 0x7FFFFF00;
    half[i] = (m + 1 >= p.K);
  }
  const int yoff = (tid < SR_ROWS) ? tid * 8 : 0x7FFFFF00;
  const int64_t row_bytes = p.ldlb * 8;
  // part i of the next tile's loads: i < NLOAD one 16-byte pair per thread, i == NLOAD the tile's targets.  The parts are
  // issued one per MFMA group of the second contraction, not as one burst in front of it: eight waves per CU reach this
  // point together, and a wave queued behind the CU's address unit cannot issue its MFMAs (5 % on the strip solve; here,
  // with 7 loads per 94 MFMAs, it measures as nothing -- 4.788 vs 4.784 ms at configs[2] -- but costs nothing either and
  // frees 16 registers).
  auto load_tile_part = [&](int64_t n0, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int64_t left = nend - n0;  // > 0
    if (i < NLOAD) {
      const int64_t lb = left * row_bytes;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<double *>(p.Lb + n0 * p.ldlb), 0, (int)(lb < 0x7FFFFF00 ? lb : 0x7FFFFF00), 0x00020000);
      stage[i] = __builtin_bit_cast(sr_double2_t, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0));
    } else {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<double *>(p.y + n0), 0, (int)(left * 8 < 0x7FFFFF00 ? left * 8 : 0x7FFFFF00), 0x00020000);
      ystage = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ry, yoff, 0, 0));
    }
#else
    (void)n0, (void)i, (void)yoff, (void)row_bytes;
#endif
  };
  auto load_tile = [&](int64_t n0) {
#pragma unroll
    for (int i = 0; i <= NLOAD; ++i) load_tile_part(n0, i);
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int e = tid + 256 * i;
      const int row = e / (KP / 2), m = 2 * (e % (KP / 2));
      // (the padding fix-up lives HERE, behind the second contraction: next to the load it made the wave wait for the
      // tile it had just requested before it could issue that contraction's MFMAs)
      sr_double2_t v = stage[i];
      v.y = half[i] ? 0.0 : v.y;
      *reinterpret_cast<sr_double2_t *>(tile_of(buf) + row * STR + m) = v;
    }
    if (tid < SR_ROWS) ys_of(buf)[tid] = ystage;
  };

  // one tile: F blocks -> cost -> (next tile's loads) -> second contraction.  LAST: rows past the slab end exist
  // (zeros in the tile); their cost derivative is forced to zero (it may be NaN: Poisson has -2 y / f).
  auto process_tile = [&](int buf, int64_t n0, bool more, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    const double *T = tile_of(buf);
    const double *Y = ys_of(buf);
    // first contraction: both 16-row blocks interleaved (two independent accumulator chains)
    sr_double4_t f0{0.0, 0.0, 0.0, 0.0}, f1{0.0, 0.0, 0.0, 0.0};
    // the tile's targets for this lane's 8 rows, fetched ahead of the first contraction (KB <= 6: registers allow it);
    // read next to their use, each pair cost a full LDS round trip in front of the per-element code
    double yv[2][4];
    if constexpr (KB <= 6) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[b][r] = Y[16 * b + q + 4 * r];
    }
    {
      const double *a0p = T + c * STR + q, *a1p = T + (16 + c) * STR + q;
      // only the last three k-quads of the padded rank can be empty (rank 89: 23 quads of 24); a wave-uniform skip
      // (KB >= 7 is at the register limit: no extra control flow there)
      auto live = [&](int kq) { return !(KB <= 6 && kq >= NQ - 3 && kq >= nq_live); };
      if constexpr (KB <= 6) {
        // Software-pipelined by pairs of k-quads: the A fragments of pair g + 1 are fetched BEFORE the four MFMAs of
        // pair g are issued, and sched_barrier keeps them there.  Left to itself the scheduler sinks every ds_read to
        // just in front of its MFMA (ds_read, s_waitcnt lgkmcnt(0), four MFMAs, ds_read, ...): the LDS latency is then
        // exposed once per 256 MFMA-cycles, and the co-resident workgroup only partly covers it (81 % pipe utilisation).
        double a0[2][2], a1[2][2];
        a0[0][0] = a0p[0];
        a0[0][1] = a0p[4];
        a1[0][0] = a1p[0];
        a1[0][1] = a1p[4];
#pragma unroll
        for (int g = 0; g < NQ / 2; ++g) {
          const int cur = g & 1, nxt = cur ^ 1;
          if (g + 1 < NQ / 2) {
            a0[nxt][0] = a0p[8 * (g + 1)];
            a0[nxt][1] = a0p[8 * (g + 1) + 4];
            a1[nxt][0] = a1p[8 * (g + 1)];
            a1[nxt][1] = a1p[8 * (g + 1) + 4];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int kq = 2 * g + h;
            if (!live(kq)) continue;
            f0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[cur][h], ufrag[kq], f0, 0, 0, 0);
            f1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[cur][h], ufrag[kq], f1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int kq = 0; kq < NQ; ++kq) {
          const double a0 = a0p[4 * kq], a1 = a1p[4 * kq];
          f0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, ufrag[kq], f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, ufrag[kq], f1, 0, 0, 0);
        }
      }
    }
    // per-element cost on the accumulator registers: register r <-> tile row q + 4 r (+16 for the second block)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      sr_double4_t &f = b ? f1 : f0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + q + 4 * r;
        const double yr = (KB <= 6) ? yv[b][r] : Y[row];
        const bool valid = !LAST || (n0 + row < nend);
        if (MODE != SR_MODE_DRIFT) {
          const double cval = cost_value(cp, yr, f[r]);
          vsum += valid ? cval : 0.0;
        }
        if (MODE != SR_MODE_VALUE) {
          const double gval = cost_deriv(cp, yr, f[r]);
          f[r] = valid ? gval : 0.0;
        }
      }
    }
    // the next tile's global loads fly during the second contraction (their staging registers are not live before)
    if (more && (MODE == SR_MODE_VALUE || NLOAD + 1 > 8)) load_tile(n0 + SR_ROWS);
    if (MODE != SR_MODE_VALUE) {
      // second contraction: D[16 ta + c][jcol] += sum_rows Lb[row][16 ta + c] * G[row][jcol]; the A-operands of row
      // group r + 1 are fetched before the MFMAs of group r are issued (and no earlier: registers)
      double an[2][KB];
      {
        const double *ap = T + q * STR + c;
#pragma unroll
        for (int ta = 0; ta < KB; ++ta) an[0][ta] = ap[16 * ta];
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int b = g >> 2, r = g & 3;
        const sr_double4_t &gq = b ? f1 : f0;
        if (g + 1 < 8) {
          const int b1 = (g + 1) >> 2, r1 = (g + 1) & 3;
          const double *ap = T + (16 * b1 + q + 4 * r1) * STR + c;
#pragma unroll
          for (int ta = 0; ta < KB; ++ta) an[(g + 1) & 1][ta] = ap[16 * ta];
        }
        if (NLOAD + 1 <= 8 && more && g <= NLOAD) load_tile_part(n0 + SR_ROWS, g);  // one part per group (see load_tile_part)
        // (without this fence the scheduler rotates the loop: group g's reads end up right in front of group g's MFMAs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < KB; ++ta) dacc[ta] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[g & 1][ta], gq[r], dacc[ta], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  };

  if (nbeg < nend) {
    load_tile(nbeg);
    store_tile(0);
  }
  __syncthreads();
  int buf = 0;
  int64_t n0 = nbeg;
  for (; n0 + SR_ROWS < nend; n0 += SR_ROWS, buf ^= 1) process_tile(buf, n0, true, std::false_type{});
  if (n0 < nend) process_tile(buf, n0, false, std::true_type{});

  if (MODE != SR_MODE_VALUE) {
    double *D = p.out + (int64_t)split * p.slab_stride;
    if (jin) {
#pragma unroll
      for (int ta = 0; ta < KB; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * ta + q + 4 * r;
          if (m < p.K) D[(int64_t)m * p.ldo + jcol] = dacc[ta][r];
        }
    }
  }
  if (MODE != SR_MODE_DRIFT) {
    vsum += __shfl_xor(vsum, 16);
    vsum += __shfl_xor(vsum, 32);
    double *vo = (MODE == SR_MODE_VALUE) ? p.out + (int64_t)split * p.slab_stride : p.vout + (int64_t)split * p.ldvo;
    if (jin && q == 0) vo[jcol] = vsum;
  }
}

// number of row slabs: enough workgroups for two per CU twice over; up to 32 slabs of >= 128 rows, up to 128 slabs of
// >= 512 rows (every slab costs a K x J block of workspace, its write and one more read by the update kernel:
// N = 2e4, K = 32, J = 256 is 0.046 ms with 32 slabs and 0.080 ms with 128, N = 1e5, K = 128, J = 256 0.43 and 0.27 ms)
// (narrow particle sets, J of a few hundred, otherwise leave most CUs idle: N = 4096, K = 128, J = 512 ran on 64
// workgroups, 75 us; the slabs cost S*K*J*8 bytes of workspace and one extra read by the update kernel)
static inline int64_t small_rank_splits(int64_t J, int64_t N, int64_t *rows_per_split) {
  const int64_t jt = (J + 63) / 64;
  int64_t s = (1024 + jt - 1) / jt;
  if (s > 128) s = 128;
  if (s > 32 && N / s < 512) s = (N / 512 > 32) ? N / 512 : 32;  // beyond 32 slabs only while each keeps >= 512 rows
  while (s > 1 && N / s < 128) --s;
  if (s < 1) s = 1;
  int64_t rows = ((N + s - 1) / s + SR_ROWS - 1) / SR_ROWS * SR_ROWS;
  if (rows < SR_ROWS) rows = SR_ROWS;
  s = (N + rows - 1) / rows;
  if (s < 1) s = 1;
  *rows_per_split = rows;
  return s;
}

}  // namespace plship
