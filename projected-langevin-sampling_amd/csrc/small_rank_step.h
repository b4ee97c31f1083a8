// ONE launch per Langevin step for small projection ranks (K <= 128 basis functions) in the launch-bound regime: the sizes
// the reference's own experiments and its profiler protocol run at (N = 100 .. 4096, M = 10 .. 128, J = 50 .. 1000:
// experiments/profiler/config.yaml:1-22, experiments/curves/*/config.yaml).  There a step is a few microseconds of
// arithmetic, and the three to four launches of the general route (small_rank.h drift -> column reduce -> update -> mean)
// cost more than the arithmetic.  This kernel does the whole iteration of experiments/trainers.py:149-158 for a cost
// without the Gaussian algebra:
//
//   F = Lb V  ->  G = cost'(y, F)  [, sum_rows cost(y, F)]  ->  D = Lb^T G  ->  fixed-order sum of D over the row slabs
//   ->  out = [U +] -eta D - eta U / lambda + sqrt(2 eta) xi   [-> energies of the INPUT particles -> their 256-column sums]
//
// Layout (differs from small_rank.h, which is built for N x J in the tens of millions):
//   * a workgroup owns 16 particle columns and a slab of data rows; its four waves take DIFFERENT 16-row blocks of the slab
//     (wave w: rows 64 t + 16 w ...), so the partial drifts of the four waves meet inside the workgroup (LDS, fixed order)
//     and a slab leaves K x 16 doubles, not K x 64: with ~256 workgroups on the chip the cross-workgroup reduction is
//     ns <= 8 slabs of 16 KB instead of 32 slabs of 64 KB, small enough for the workgroup that arrives LAST at a column
//     block to finish it alone (nobody waits: no co-residency assumption, no spinning);
//   * every wave streams its own rows global -> LDS by LDS-DMA (one buffer instruction per row, no staging registers),
//     double-buffered in a wave-private region: the main loop has NO workgroup barrier;
//   * per 16-row block: K/4 MFMAs give F (16 x 16) in the accumulator layout, the cost derivative is applied in place, and
//     the four accumulator registers are the B-operands of the second contraction, whose A-operands are the same LDS rows
//     read the other way round (as in small_rank.h);
//   * finishing: the last workgroup of a column block adds the slabs in ascending order (every byte write-through / read
//     past the L1: the visibility rules of gemm_tn_f64_kg.h), applies prior drift, noise and step, and -- when energies are
//     asked for -- adds cost partial sums + prior energy, stores the energies and bumps the counter of its 256-column chunk;
//     the last column block of a chunk adds the chunk's energies in the library's fixed order (chunk256_sum) and stores the
//     sum, possibly straight into pinned host memory: the training loop polls it, one launch per iteration.
// Counters are zero on entry and left zero.  Reference: projected_langevin_sampling.py:107-138,
// basis/orthonormal.py:98-159, costs/{*}.py.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "cost_device.h"
#include "philox.h"
#include "step_params.h"

// The hand-over between workgroups below (write-through `sc1` stores, a drained store queue, one relaxed agent-scope atomic,
// `sc1` loads past the L1) is written against the cache hierarchy of gfx942 / gfx950 (per-XCD L2s that are not coherent with
// each other, write-through vector L1s): another target needs its own protocol, not a silent recompile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the inter-workgroup hand-over of libplship is written for gfx942 / gfx950"
#endif
namespace plship {

typedef double srs_double4_t __attribute__((ext_vector_type(4)));

struct SrStepP {
  const double *Lb;  // N x K row-major (At of the orthonormal basis), 16-byte aligned, even leading dimension
  int64_t ldlb;
  const double *U;  // K x J: the coordinates the forward map contracts -- the particles of the orthonormal basis, V = k(Z,Z)^-1 U
                    // of the inducing-point basis -- and the operand of the prior term
  int64_t ldu;
  const double *Uadd;  // K x J: the particles the update is added to (add_u = 1); NULL: U itself
  int64_t lduadd;
  const double *y;
  const double *lam;  // (K) eigenvalues: prior drift U / lambda, prior energy U^2 / (2 lambda); NULL: the constant below
  double pconst;      // lam == NULL: prior drift pconst * U, prior energy pconst * U^2 / 2 (inducing-point basis: M)
  int64_t N, J;
  int K;
  int64_t rows_per_split;  // multiple of 64
  int nsplit;
  CostP cp;
  double *out;  // K x J: dU (add_u = 0) or U + dU (add_u = 1)
  int64_t ldo;
  int add_u;
  EtaP etap;
  NoiseP nz;
  unsigned *cb_sync;     // [cdiv(J, 16)] arrival counters of the column blocks (nsplit > 1)
  unsigned *chunk_sync;  // [cdiv(J, 256)] arrival counters of the 256-column chunks (esums != NULL)
  double *slab;          // [cdiv(J, 16)][nsplit][K][16] partial drifts (nsplit > 1)
  double *vslab;         // [cdiv(J, 16)][nsplit][16] partial cost sums (nsplit > 1, VALUE)
  double *e;             // (J) energies of the input particles (VALUE)
  double *esums;         // [cdiv(J, 256)] chunk sums of e, may be pinned host memory; NULL: not wanted
#ifdef PLS_SRS_PROBE
  int debug_stop;        // probe builds: 1 = return after the main loop, 2 = after the in-workgroup sum, 3 = after the arrival, 4 = after the slab loads
#endif
  int64_t Ndata;         // rows [0, Ndata) of Lb are data rows (cost cp against y); rows [Ndata, N) are PRIOR rows: their "cost"
                         // is f^2 / 2 with derivative f -- a prior drift R^T R u and a prior energy |R u|^2 / 2 for the block R of
                         // Lb they hold (the inducing-point basis in whitened coordinates: R^T R = M (Lc^T Lc)^-1).  Ndata = N: none.
                         // y holds Ndata entries
  double *sums16;        // [cdiv(J, 16)] sums of e over the 16 columns of each column block (ascending), may be pinned host
                         // memory; NULL: not wanted.  Costs nothing (the finishing workgroup of a column block holds them); the
                         // chunk sums cost a second hand-over between workgroups
};

constexpr int SRS_TILE = 16;       // rows of a wave's tile (one MFMA row block)
constexpr int SRS_WG_ROWS = 64;    // rows the four waves of a workgroup take per round
constexpr int SRS_TILE_OPS = 17;   // vector-memory instructions per tile: 16 rows + the tile's targets
template <int KB>
constexpr int srs_stride() { return 16 * KB + 2; }  // doubles per LDS row (small_rank.h: both read patterns conflict-free)
// tiles a wave keeps in LDS (one in use, the others in flight): what 160 KB of LDS hold for four waves, and at most five
// (s_waitcnt counts up to 63 outstanding instructions = three tiles in flight BEHIND the one being waited for)
template <int KB>
constexpr int srs_nbuf() { return KB <= 3 ? 5 : KB == 4 ? 4 : KB <= 6 ? 3 : 2; }
template <int KB>
constexpr int srs_tile_doubles() { return SRS_TILE * srs_stride<KB>() + SRS_TILE; }  // rows + targets
template <int KB>
constexpr int srs_wave_doubles() { return srs_nbuf<KB>() * srs_tile_doubles<KB>(); }
// wave regions (tiles, later the waves' partial drifts) + cost partial sums [4][16] + prior partial sums [16][16] + one word
template <int KB>
constexpr size_t srs_lds_bytes() { return (size_t)(4 * srs_wave_doubles<KB>() + 64 + 256 + 8) * sizeof(double); }

// The plan: slabs per column block.  A round of tiles (64 rows of a workgroup) takes ~0.28 us per 16 functions when the matrix
// pipe bounds it and never less than ~1.1 us (narrow ranks: seventeen LDS-DMA instructions per tile and the latencies between the
// two contractions, which nothing overlaps at one wave per SIMD); handing a column block over between workgroups costs three
// memory round trips past the caches (~8 us with the finishing loads).  So: ONE slab while the whole column block is at most
// ~6 us of tiles (the cost function adds up to 1 us per round to them); otherwise as many slabs as put one workgroup on every CU, each at least two rounds long.
// (profiles/r05_sr_step_probe.txt, r05_ipb_small_probe.txt; tools/sr_step_sweep.py)
static inline int64_t small_rank_step_splits(int64_t J, int64_t N, int K, int64_t *rows_per_split) {
  const int64_t ncb = (J + 15) / 16;
  const int64_t kb = (K + 15) / 16;
  const int64_t rounds = (N + SRS_WG_ROWS - 1) / SRS_WG_ROWS;
  const double round_us = 0.28 * (double)kb > 1.1 ? 0.28 * (double)kb : 1.1;
  int64_t s = 1;
  if ((double)rounds * round_us > 6.0) {
    s = 256 / ncb;  // (rounded DOWN: 315 workgroups on 256 CUs take two passes, 252 one)
    if (s > rounds / 2) s = rounds / 2;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
  }
  int64_t rows = ((N + s - 1) / s + SRS_WG_ROWS - 1) / SRS_WG_ROWS * SRS_WG_ROWS;
  if (rows < SRS_WG_ROWS) rows = SRS_WG_ROWS;
  if (s > 1) {
    // a few rows past a multiple of the round (N = 1000 data rows + 32 prior rows: 129 rows per slab) must not cost every slab a
    // whole round: rather one short slab more, while the finishing workgroup still fetches all slabs in one batch of loads and
    // the grid stays within one workgroup per CU
    int64_t down = (N / s) / SRS_WG_ROWS * SRS_WG_ROWS;
    if (down < 2 * SRS_WG_ROWS) down = 2 * SRS_WG_ROWS;
    const int64_t sd = (N + down - 1) / down;
    const int64_t batch = kb <= 2 ? 16 : kb <= 4 ? 12 : 8;  // (slabs per batch of finishing loads, SB in the kernel)
    if (down < rows && sd <= batch && ncb * sd <= 256) rows = down;
  }
  s = (N + rows - 1) / rows;
  if (s < 1) s = 1;
  *rows_per_split = rows;
  return s;
}

static inline size_t small_rank_step_sync_words(int64_t J) { return (size_t)((J + 15) / 16 + (J + 255) / 256); }
static inline size_t small_rank_step_slab_bytes(int64_t J, int64_t N, int K) {
  int64_t rows;
  const int64_t ns = small_rank_step_splits(J, N, K, &rows);
  if (ns <= 1) return 0;
  return (size_t)((J + 15) / 16) * ns * ((size_t)K + 1) * 16 * sizeof(double);
}

// dU of one element (langevin_update_kernel's formula), evaluated without contraction: the same bits with and without the
// energy by-product, whatever else the instantiation computes next to it
__device__ __forceinline__ double srs_langevin_delta(double eta, double sq2eta, double drift, double ps, double u, double z) {
#pragma clang fp contract(off)
  return -eta * drift - eta * ps * u + sq2eta * z;
}

template <int KB, int COST, int LINK, bool VALUE, bool PRIOR = false>
// (PRIOR: the last rows of Lb may be prior rows, SrStepP.Ndata -- instantiations of their own, so that the kernels without them
// stay the instruction sequence they were tuned as: a select per element in the cost loop cost the Poisson step 1.2 us)
// (launch bounds of TWO workgroups per CU although the LDS image admits one: the bound caps the kernel at 256 registers, all of
// them vector registers.  Allowed 512, the register allocator keeps the loop-carried drift accumulators of the wide ranks in
// vector registers and copies all of them to accumulation registers and back around every tile: ~120 copies per 64 MFMAs.)
__global__ __launch_bounds__(256, 2) void small_rank_step_kernel(SrStepP p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NQ = 4 * KB;   // k-quads of the first contraction
  constexpr int STR = srs_stride<KB>();
  constexpr int NBUF = srs_nbuf<KB>();
  constexpr int TD = srs_tile_doubles<KB>();
  constexpr int WD = srs_wave_doubles<KB>();
  constexpr int NPT = (KB + 1) / 2;  // row pairs {i, i + 4} per thread in the finishing phase (8 KB pairs over 16 thread rows)
  static_assert(WD >= 256 * KB && (NBUF - 2) * SRS_TILE_OPS <= 63, "wave region / s_waitcnt range");
  extern __shared__ __attribute__((aligned(16))) double srs_lds[];
  typedef __attribute__((address_space(3))) void *lds_ptr_t;

  CostP cp = p.cp;
  if constexpr (COST >= 0) {
    cp.cost = COST;
    cp.link = LINK;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, c = lane & 15;
  const int cb = blockIdx.x, split = blockIdx.y;
  const int64_t jcol = (int64_t)cb * 16 + c;
  const bool jin = jcol < p.J;
  const int64_t nbeg = (int64_t)split * p.rows_per_split;
  const int64_t nend = (nbeg + p.rows_per_split < p.N) ? nbeg + p.rows_per_split : p.N;
  const int nq_live = (p.K + 3) >> 2;
  const bool tail_dead = ((nq_live - 1) * 4 + q) >= p.K;  // this lane's k index of the last live quad is past the rank

  double *wreg = srs_lds + wave * WD;   // this wave's tiles; later its partial drift [16 KB][16]
  double *vred = srs_lds + 4 * WD;      // [4][16] cost partial sums of the waves
  double *pred = vred + 64;             // [16][16] prior-energy partial sums of the finishing thread rows
  unsigned *word = reinterpret_cast<unsigned *>(pred + 256);

  // the wave's particle columns as B-operands: ufrag[kq] = U[4 kq + q][jcol]
  double ufrag[NQ];
#pragma unroll
  for (int kq = 0; kq < NQ; ++kq) {
    const int m = 4 * kq + q;
    ufrag[kq] = (jin && m < p.K) ? p.U[(int64_t)m * p.ldu + jcol] : 0.0;
  }
  srs_double4_t dacc[KB];
#pragma unroll
  for (int ta = 0; ta < KB; ++ta) dacc[ta] = srs_double4_t{0.0, 0.0, 0.0, 0.0};
  double vsum = 0.0;

  // One tile = 16 rows of Lb for this wave, copied global -> LDS by one LDS-DMA instruction per row (lanes 0 .. cdiv(K, 2) - 1
  // carry 16 bytes each -- never a byte past column K rounded up to even, which an even leading dimension always holds --; the
  // row pad of the LDS image stays), and the 16 targets behind them by one more.  Rows past the end of the matrix re-read its
  // last row (finite data; their derivative is forced to zero below).  LDS columns from K (rounded up to even) to 16 KB are
  // never written and hold whatever the LDS held: harmless -- in the first contraction they are masked or skipped, in the
  // second they only reach output rows >= K, which nobody stores.  Every operand of the DMA instructions is made wave-uniform
  // explicitly (64-bit products are vector instructions: without readfirstlane each copy becomes a waterfall loop).
  const int row_bytes = __builtin_amdgcn_readfirstlane((int)(p.ldlb * 8));
  const int voff = lane * 16;
  const int nl = (p.K + 1) >> 1;
  const int w_rows0 = __builtin_amdgcn_readfirstlane((int)(nend - nbeg - SRS_TILE * wave));  // rows from the wave's first tile to the slab end
  const int ntiles = w_rows0 > 0 ? (w_rows0 + SRS_WG_ROWS - 1) / SRS_WG_ROWS : 0;               // tiles of this wave (uniform)
  auto tile_row0 = [&](int t) { return nbeg + SRS_TILE * wave + (int64_t)SRS_WG_ROWS * t; };
  auto issue_tile = [&](int t, int buf) {
    const int64_t n0w = tile_row0(t);
    const int64_t rows_mem = p.N - n0w;  // >= 1
    const int64_t bytes64 = ((rows_mem - 1) * p.ldlb + p.K) * 8;
    const int bytes = __builtin_amdgcn_readfirstlane((int)(bytes64 < 0x7FFFFFF0 ? bytes64 : 0x7FFFFFF0));
    const uint64_t base = reinterpret_cast<uint64_t>(p.Lb + n0w * p.ldlb);
    const uint64_t ubase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(ubase), 0, bytes, 0x00020000);
    const int last = __builtin_amdgcn_readfirstlane((int)(rows_mem < SRS_TILE ? rows_mem - 1 : SRS_TILE - 1));
    double *T = wreg + buf * TD;
    if (lane < nl) {
      if (last == SRS_TILE - 1) {  // (all but the matrix's last rows: the row offsets are loop-invariant scalars)
#pragma unroll
        for (int r = 0; r < SRS_TILE; ++r)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(T + r * STR), 16, voff, r * row_bytes, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < SRS_TILE; ++r) {
          const int rr = r < last ? r : last;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(T + r * STR), 16, voff, rr * row_bytes, 0, 0);
        }
      }
    }
    // the tile's targets: 16 doubles behind the rows (lanes 0 .. 7; past the end of y the lanes read nothing: those rows
    // are not valid and their cost is discarded by a select)
    const uint64_t ybase = reinterpret_cast<uint64_t>(p.y + n0w);
    const uint64_t uy = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ybase >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ybase);
    int ybytes;
    if constexpr (PRIOR) {  // the targets end with the DATA rows: a tile of prior rows must not read behind y (<= 0 rows: nothing)
      const int64_t rows_y = p.Ndata - n0w;
      ybytes = __builtin_amdgcn_readfirstlane((int)(rows_y <= 0 ? 0 : rows_y * 8 < 0x7FFFFFF0 ? rows_y * 8 : 0x7FFFFFF0));
    } else {
      ybytes = __builtin_amdgcn_readfirstlane((int)(rows_mem * 8 < 0x7FFFFFF0 ? rows_mem * 8 : 0x7FFFFFF0));
    }
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(uy), 0, ybytes, 0x00020000);
    if (lane < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(T + SRS_TILE * STR), 16, voff, 0, 0, 0);
  };
  // tile `t` has landed once at most `behind` younger tiles are still in flight (vector-memory operations complete in order)
  auto wait_tile = [&](int behind) {
    if (behind <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SRS_TILE_OPS) : "memory");
    else if (behind == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SRS_TILE_OPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * SRS_TILE_OPS) : "memory");
  };

  auto process_tile = [&](int buf, int64_t n0w) {
    const double *T = wreg + buf * TD;
    const double *Y = T + SRS_TILE * STR;
    double yv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) yv[r] = Y[q + 4 * r];
    srs_double4_t f{0.0, 0.0, 0.0, 0.0};
    {
      // first contraction, software-pipelined by pairs of k-quads (small_rank.h: left to itself the scheduler sinks every
      // ds_read to just in front of its MFMA)
      const double *ap = T + c * STR + q;
      auto live = [&](int kq) { return !(kq >= NQ - 3 && kq >= nq_live); };  // only the last three quads can be empty
      auto mask = [&](int kq, double a) { return (kq >= NQ - 4 && kq == nq_live - 1 && tail_dead) ? 0.0 : a; };
      double a[2][2];
      a[0][0] = ap[0];
      a[0][1] = ap[4];
#pragma unroll
      for (int g = 0; g < NQ / 2; ++g) {
        const int cur = g & 1, nxt = cur ^ 1;
        if (g + 1 < NQ / 2) {
          a[nxt][0] = ap[8 * (g + 1)];
          a[nxt][1] = ap[8 * (g + 1) + 4];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kq = 2 * g + h;
          if (!live(kq)) continue;
          f = __builtin_amdgcn_mfma_f64_16x16x4f64(mask(kq, a[cur][h]), ufrag[kq], f, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the first A-operands of the second contraction travel while the cost is evaluated
    double an[2][KB];
    {
      const double *ap = T + q * STR + c;
#pragma unroll
      for (int ta = 0; ta < KB; ++ta) an[0][ta] = ap[16 * ta];
    }
    // per-element cost on the accumulator registers: register r <-> tile row q + 4 r
    bool data_tile = true;
    if constexpr (PRIOR) data_tile = n0w + SRS_TILE <= p.Ndata;  // (wave-uniform)
    if (data_tile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = n0w + q + 4 * r < nend;
        if constexpr (VALUE) {
          const double cval = cost_value(cp, yv[r], f[r]);
          vsum += valid ? cval : 0.0;
        }
        const double gval = cost_deriv(cp, yv[r], f[r]);
        f[r] = valid ? gval : 0.0;
      }
    } else {  // a tile with prior rows (at most cdiv(K, 16) + 1 tiles of the last slab): f^2 / 2 and f for those
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = n0w + q + 4 * r < nend;
        const bool prow = n0w + q + 4 * r >= p.Ndata;
        if constexpr (VALUE) {
          const double cval = prow ? 0.5 * (f[r] * f[r]) : cost_value(cp, yv[r], f[r]);
          vsum += valid ? cval : 0.0;
        }
        const double gval = prow ? f[r] : cost_deriv(cp, yv[r], f[r]);
        f[r] = valid ? gval : 0.0;
      }
    }
    // second contraction: D[16 ta + c][jcol] += sum_rows Lb[row][16 ta + c] * G[row][jcol]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r + 1 < 4) {
        const double *ap = T + (q + 4 * (r + 1)) * STR + c;
#pragma unroll
        for (int ta = 0; ta < KB; ++ta) an[(r + 1) & 1][ta] = ap[16 * ta];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ta = 0; ta < KB; ++ta) dacc[ta] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[r & 1][ta], f[r], dacc[ta], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // main loop of the wave: no workgroup barrier.  NBUF - 1 tiles are requested ahead; tile t is awaited by count (the younger
  // requests stay in flight), and the request for tile t + NBUF - 1 goes into the buffer tile t - 1 has just left.
  {
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
      if (t < ntiles) issue_tile(t, t);
    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
      const int ahead = ntiles - 1 - t;  // tiles after t
      wait_tile(ahead < NBUF - 2 ? ahead : NBUF - 2);
      if (t + NBUF - 1 < ntiles) issue_tile(t + NBUF - 1, buf == 0 ? NBUF - 1 : buf - 1);
#ifdef PLS_SRS_NO_COMPUTE
      if (p.K < 0)
#endif
      process_tile(buf, tile_row0(t));
      buf = (buf + 1 == NBUF) ? 0 : buf + 1;
    }
  }

#ifdef PLS_SRS_PROBE
  if (p.debug_stop == 1) return;
#endif
  // ---- the four waves' partial drifts meet in LDS (fixed order w0 + w1 + w2 + w3) --------------------------------------
  // (a wave writes into its OWN region: its tile reads are complete, nobody else ever touched the region)
#pragma unroll
  for (int ta = 0; ta < KB; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) wreg[(16 * ta + q + 4 * r) * 16 + c] = dacc[ta][r];  // [16 KB rows][16]
  if constexpr (VALUE) {
    vsum += __shfl_xor(vsum, 16);
    vsum += __shfl_xor(vsum, 32);
    if (q == 0) vred[wave * 16 + c] = vsum;
  }
  __syncthreads();
  // finishing layout: thread (tp = tid >> 4, tc = tid & 15) owns column tc and the row pairs pr = tp + 16 e,
  // rows ib = 8 (pr >> 2) + (pr & 3) and ib + 4: the two rows of one Philox call (philox.h)
  const int tp = tid >> 4, tc = tid & 15;
  const int64_t fcol = (int64_t)cb * 16 + tc;
  const bool fin_col = fcol < p.J;
  double d[NPT][2];
#pragma unroll
  for (int e = 0; e < NPT; ++e) {
    const int pr = tp + 16 * e;
    const int ib = 8 * (pr >> 2) + (pr & 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = ib + 4 * h;
      double s = 0.0;
      if (pr < 8 * KB) {
#pragma unroll
        for (int w = 0; w < 4; ++w) s += srs_lds[w * WD + i * 16 + tc];
      }
      d[e][h] = s;
    }
  }
  double vcol = 0.0;  // threads 0 .. 15: the workgroup's cost partial sum of column tid
  if constexpr (VALUE) {
    if (tid < 16) vcol = ((vred[tid] + vred[16 + tid]) + vred[32 + tid]) + vred[48 + tid];
  }

#ifdef PLS_SRS_PROBE
  if (p.debug_stop == 2) return;
#endif
  // What the update needs besides the drift -- step size, noise, the particles and 1 / lambda of this thread's rows -- does not
  // depend on the other slabs: with several slabs it is fetched / drawn while the arrival counter's answer travels.
  double fz[NPT][2], fu[NPT][2], fps[NPT][2], fb[NPT][2];  // noise, prior operand, prior weight, base of the new state
  double eta = 0.0, sq2eta = 0.0;
  auto prefetch_update_operands = [&]() {
    if (fin_col) {
      eta = p.etap.at(fcol);
      sq2eta = sqrt(2.0 * eta);
    }
    const int64_t jg = p.nz.global_column(fcol);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      const int pr = tp + 16 * e;
      const int ib = 8 * (pr >> 2) + (pr & 3);
      const bool on = fin_col && pr < 8 * KB && ib < p.K;
      double z0 = 0.0, z1 = 0.0;
      if (on) {
        if (p.nz.kind == PLS_NOISE_PHILOX) {
          normal_pair(p.nz.seed, p.nz.live_step(), ib, jg, z0, z1);
        } else if (p.nz.kind == PLS_NOISE_INJECTED) {
          z0 = p.nz.xi[ib * p.nz.ldxi + fcol];
          if (ib + 4 < p.K) z1 = p.nz.xi[(ib + 4) * p.nz.ldxi + fcol];
        }
      }
      fz[e][0] = z0;
      fz[e][1] = z1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = ib + 4 * h;
        const bool in = on && i < p.K;
        fu[e][h] = in ? p.U[(int64_t)i * p.ldu + fcol] : 0.0;
        fps[e][h] = in ? (p.lam ? 1.0 / p.lam[i] : p.pconst) : 0.0;
        if (p.Uadd) fb[e][h] = (in && p.add_u) ? p.Uadd[(int64_t)i * p.lduadd + fcol] : 0.0;
      }
    }
  };

  // ---- several slabs: leave this one write-through, the last arriver of the column block adds them in ascending order ----
  if (p.nsplit > 1) {
    double *mine = p.slab + ((int64_t)cb * p.nsplit + split) * ((int64_t)p.K * 16);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      const int pr = tp + 16 * e;
      const int ib = 8 * (pr >> 2) + (pr & 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = ib + 4 * h;
        if (pr < 8 * KB && i < p.K) __hip_atomic_store(mine + i * 16 + tc, d[e][h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if constexpr (VALUE) {
      if (tid < 16)
        __hip_atomic_store(p.vslab + ((int64_t)cb * p.nsplit + split) * 16 + tid, vcol, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have been performed
    __syncthreads();
    unsigned ticket = 0;
    if (tid == 0) ticket = __hip_atomic_fetch_add(p.cb_sync + cb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    prefetch_update_operands();  // (under the atomic's round trip)
    if (tid == 0) word[0] = ticket;
    __syncthreads();
    if (word[0] + 1 != (unsigned)p.nsplit) return;  // (workgroup-uniform) somebody else finishes this column block
    if (tid == 0) __hip_atomic_store(p.cb_sync + cb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef PLS_SRS_PROBE
    if (p.debug_stop == 3) return;
#endif
    const double *all = p.slab + (int64_t)cb * p.nsplit * ((int64_t)p.K * 16);
    // every element of SB slabs requested before the first addition (a thread owns 2 NPT elements: up to 64 loads in flight,
    // one memory round trip per SB slabs), then added slab by slab in ascending order
    constexpr int SB = NPT >= 3 ? 8 : NPT == 2 ? 12 : 16;
    int eoff[NPT][2];
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      const int pr = tp + 16 * e;
      const int ib = 8 * (pr >> 2) + (pr & 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = ib + 4 * h;
        eoff[e][h] = (pr < 8 * KB && i < p.K) ? i * 16 + tc : tc;  // (elements past the rank re-read row 0: never used)
        d[e][h] = 0.0;
      }
    }
    const int64_t sstride = (int64_t)p.K * 16;
    for (int s0 = 0; s0 < p.nsplit; s0 += SB) {
      double t[SB][NPT][2];
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        const int sk = s0 + k < p.nsplit ? s0 + k : p.nsplit - 1;  // (slabs past the last re-read it: dropped below)
#pragma unroll
        for (int e = 0; e < NPT; ++e)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            t[k][e][h] = __hip_atomic_load(all + sk * sstride + eoff[e][h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < SB; ++k)
        if (s0 + k < p.nsplit) {
#pragma unroll
          for (int e = 0; e < NPT; ++e)
#pragma unroll
            for (int h = 0; h < 2; ++h) d[e][h] += t[k][e][h];
        }
    }
    if constexpr (VALUE) {
      if (tid < 16) {
        const double *vs = p.vslab + (int64_t)cb * p.nsplit * 16 + tid;
        double s = 0.0;
        for (int s0 = 0; s0 < p.nsplit; s0 += 8) {  // eight loads in flight, added in ascending slab order
          double t[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            t[k] = __hip_atomic_load(vs + (s0 + k < p.nsplit ? s0 + k : p.nsplit - 1) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (s0 + k < p.nsplit) s += t[k];
        }
        vcol = s;
      }
    }
  } else {
    prefetch_update_operands();
  }
#ifdef PLS_SRS_PROBE
  if (p.debug_stop == 4) return;
#endif

  // ---- prior drift + noise + step (langevin_update_kernel's formula), prior energy of the rows this thread walks ----------
  double prior = 0.0;
  if (fin_col) {
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      const int pr = tp + 16 * e;
      const int ib = 8 * (pr >> 2) + (pr & 3);
      if (pr >= 8 * KB || ib >= p.K) continue;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = ib + 4 * h;
        if (i < p.K) {
          const double ps = fps[e][h], u = fu[e][h];
          const double dd = srs_langevin_delta(eta, sq2eta, d[e][h], ps, u, fz[e][h]);
          p.out[(int64_t)i * p.ldo + fcol] = p.add_u ? (p.Uadd ? fb[e][h] : u) + dd : dd;
          if constexpr (VALUE) prior += u * u * ps;
        }
      }
    }
  }
  if constexpr (VALUE) {
    // e[col] = cost sum + 1/2 sum_m u_m^2 / lambda_m (orthonormal.py:120-125): the sixteen thread rows in ascending order
    pred[tp * 16 + tc] = prior;
    __syncthreads();
    double ev = 0.0;
    if (tid < 16 && fin_col) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += pred[k * 16 + tid];
      ev = vcol + 0.5 * s;
      if (p.esums)
        __hip_atomic_store(p.e + fcol, ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read by the chunk's last column block
      else
        p.e[fcol] = ev;
    }
    if (p.sums16) {  // (kernel-uniform) the column block's own sum, columns in ascending order
      if (tid < 16) {  // (lanes 0 .. 15 of wave 0 hold the sixteen energies; columns past J hold zero)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += __shfl(ev, k);
        if (tid == 0) p.sums16[cb] = s;
      }
    }
    if (!p.esums) return;  // (kernel-uniform)
    // the 256-column chunk sums of the energies (chunk256_sum's order: the values pls_chunk_sums computes, bit for bit)
    const int64_t chunk = ((int64_t)cb * 16) >> 8, c0 = chunk << 8, c1 = (c0 + 256 < p.J) ? c0 + 256 : p.J;
    const unsigned expect = (unsigned)((c1 - c0 + 15) / 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) word[0] = __hip_atomic_fetch_add(p.chunk_sync + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (word[0] + 1 != expect) return;  // (workgroup-uniform)
    if (tid == 0) __hip_atomic_store(p.chunk_sync + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t col = c0 + tid;
    double v = (col < c1) ? __hip_atomic_load(p.e + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    double *ws = vred;  // (free again)
    __syncthreads();
    if ((tid & 63) == 0) ws[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) p.esums[chunk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
  }
#else
  (void)p;
#endif
}

}  // namespace plship
