// fp64 MFMA contraction for FEW output tiles: 64 x 64 tiles whose k range is split over KG wave groups INSIDE the
// workgroup (no slabs in global memory, the epilogue is applied once to the fixed-order sum).
//
// Why: the particle columns are independent (reference: basis/orthonormal.py:151-158, inducing_point.py:143-149), so an
// 8-GPU run hands each rank J / 8 columns.  At M_k = 1024, J = 1024 the 128 x 128 configuration of gemm_tn_f64.h has 64
// tiles for 256 CUs and the 64 x 64 one puts ONE wave on each SIMD with register-staged operands (60 % MFMA-pipe
// utilisation, profiles/r02_pmc_summary.json).  Here a 64 x 64 tile is owned by 4 KG waves: group g contracts the k-steps
// g, g + KG, g + 2 KG, ... (16 rows each), every wave a 32 x 32 block of its group's partial sum, so a 256-tile problem
// runs two waves per SIMD on every CU and a 512-tile problem four.
//
// gfx950 mapping
//   * global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... offen lds`): one wave-instruction deposits 1 KiB = TWO k-rows of a
//     64-wide operand tile.  The LDS image is linear in (row, position); the lane picks the global columns that belong at
//     its position, and odd rows hold their 16-column halves swapped (position p of an odd row = column p ^ 16): the four
//     16-lane groups of an MFMA operand fetch (rows 4 kq + q, 16 consecutive doubles each) then fall on disjoint banks
//     without any row padding.  The swizzle lives in a loop-invariant lane offset; the k-loop has no VALU but the MFMAs.
//   * per super-step (16 KG rows) a wave issues 4 DMA instructions, 16 ds_read_b64 and 16 MFMAs; the rotated pipeline of
//     gemm_tn_f64.h (DMA of the next step first, the last k-quad's MFMAs after the barrier) carries over.
//   * the K tail (fewer than 16 KG rows) goes through registers with a zero fill, like the big configuration's.
//   * epilogue: the groups exchange the halves of their 32 x 32 blocks through LDS and every wave finishes a 16 x 32
//     block of the SUM (group order 0, 1, ...: deterministic), i.e. the workgroup looks like 4 x 2 waves of 16 x 32 to the
//     epilogues of gemm_tn_f64.h / plship.hip, which run unchanged (Langevin update with in-register Philox, energy
//     partials, plain store).
//   * tri: 1 = L[k][i] == 0 for k > i (tile contracts k < i0 + 64), 2 = L[k][i] == 0 for k < i (tile contracts k >= i0).
#pragma once
#include "gemm_tn_f64.h"

namespace plship {

template <int KG>
struct KgGeom {
  static constexpr int NW = 4 * KG, NT = 64 * NW, SROWS = 16 * KG;
  static constexpr int TILE = SROWS * 64;  // doubles per operand tile of one super-step
  static constexpr int BUF = 2 * TILE;     // L tile + R tile
  static constexpr int EPI = NW * epi_lds_doubles_per_wave<32>();
  static constexpr int EXCH = (KG > 1) ? NW * 8 * 64 : 0;
  static constexpr int M1 = (2 * BUF > EPI) ? 2 * BUF : EPI;
  static constexpr int LDS_DOUBLES = (M1 > EXCH) ? M1 : EXCH;
};

template <int KG, class Epilogue>
__global__ __launch_bounds__(256 * KG, 4) void gemm_tn_f64_kg_kernel(GemmShape g, Epilogue epi) {
  static_assert(KG == 1 || KG == 2, "epilogue hand-over is written for one or two k-groups");
#if defined(__HIP_DEVICE_COMPILE__)
  using G = KgGeom<KG>;
  constexpr int NW = G::NW, NT = G::NT, SROWS = G::SROWS, TILE = G::TILE, BUF = G::BUF;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  typedef __attribute__((address_space(3))) void *lds_ptr_t;

  int tile_i, tile_j;
  gemm_tile_coords(blockIdx.x, g.nti, g.ntj, tile_i, tile_j);
  const int64_t i0 = (int64_t)tile_i * 64, j0 = (int64_t)tile_j * 64;
  const int split = blockIdx.y;
  int64_t kbeg = 0, kend = g.K;
  if (gridDim.y > 1) {
    kbeg = (int64_t)split * g.kchunk;
    kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
  }
  if (g.tri == 1 && i0 + 64 < kend) kend = i0 + 64;
  if (g.tri == 2 && i0 > kbeg) kbeg = i0;
  const int64_t klen = kend > kbeg ? kend - kbeg : 0;
  const int nS = (int)(klen / SROWS);           // super-steps copied by DMA
  const int rem = (int)(klen - (int64_t)nS * SROWS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w = wave & 3, wr = w >> 1, wc = w & 1;
  const int q = lane >> 4, c16 = lane & 15;

  // operand fragments: row 16 grp + 4 kq + q of the super-step, columns 32 wr + 16 t + c16 (halves swapped in odd rows)
  const int fl0 = (grp * 16 + q) * 64 + wr * 32 + ((q & 1) << 4) + c16, fl1 = fl0 ^ 16;
  const int fr0 = TILE + (grp * 16 + q) * 64 + wc * 32 + ((q & 1) << 4) + c16, fr1 = fr0 ^ 16;

  // DMA: this wave copies row pairs `wave` and `wave + NW` of both operand tiles
  const int rp = lane >> 5, pos = (lane & 31) * 2, colx = pos ^ (rp << 4);
  const int voffl = (int)(((int64_t)rp * g.ldl + i0 + colx) * 8), voffr = (int)(((int64_t)rp * g.ldr + j0 + colx) * 8);
  const int pairl = (int)(g.ldl * 16), pairr = (int)(g.ldr * 16);  // bytes per row pair
  const char *lnext = reinterpret_cast<const char *>(g.L + kbeg * g.ldl);
  const char *rnext = reinterpret_cast<const char *>(g.R + kbeg * g.ldr);
  const char *const lend = reinterpret_cast<const char *>(g.L + (g.K - 1) * g.ldl + g.I);  // (g.K >= 1 whenever a load is issued)
  const char *const rend = reinterpret_cast<const char *>(g.R + (g.K - 1) * g.ldr + g.J);
  const int64_t lstep = (int64_t)SROWS * g.ldl * 8, rstep = (int64_t)SROWS * g.ldr * 8;
  // descriptor range = bytes from the step's first row to the end of the operand (overhanging lanes of the last rows must
  // not touch memory behind the matrix), clamped to 31 bits; scalar arithmetic only
  auto range_of = [](uint64_t bytes) { return (uint32_t)(bytes >> 31) ? 0x7FFFFFF0 : (int)(uint32_t)bytes; };
  auto dma_load = [&](int buf) {  // copies the NEXT full super-step into buffer `buf` (wave-uniform)
    const __amdgpu_buffer_rsrc_t lr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(lnext), 0, range_of((uint64_t)(lend - lnext)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rnext), 0, range_of((uint64_t)(rend - rnext)), 0x00020000);
    double *b = lds + buf * BUF;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int pr = wave + p * NW;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (lds_ptr_t)(b + pr * 128), 16, voffl, pr * pairl, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(b + TILE + pr * 128), 16, voffr, pr * pairr, 0, 0);
    }
    lnext += lstep;
    rnext += rstep;
  };

  AccFrag<2, 2> acc;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

  // K tail (fewer than SROWS rows): through registers with a zero fill, requested NOW so that the whole k-loop hides the
  // latency; thread t stages pairs t and t + NT of each tile.  It is contracted after the loop, in code of its own.
  double2_t tl[2], tr[2];
  if (rem > 0) {
    const char *lt = lnext + (int64_t)nS * lstep, *rt = rnext + (int64_t)nS * rstep;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int p = tid + s * NT, row = p >> 5, cp = (p & 31) * 2;
      const bool kin = row < rem;
      const bool lin = kin && (i0 + cp < g.I), rin = kin && (j0 + cp < g.J);
      const double2_t lv = *reinterpret_cast<const double2_t *>(lt + ((int64_t)(lin ? row : 0) * g.ldl + (lin ? i0 + cp : 0)) * 8);
      const double2_t rv = *reinterpret_cast<const double2_t *>(rt + ((int64_t)(rin ? row : 0) * g.ldr + (rin ? j0 + cp : 0)) * 8);
      tl[s] = lin ? lv : double2_t{0.0, 0.0};
      tr[s] = rin ? rv : double2_t{0.0, 0.0};
    }
  }

  auto read_frag = [&](int buf, int kq, double (&a)[2], double (&b)[2]) {
    const double *p = lds + buf * BUF + kq * 256;
    a[0] = p[fl0];
    a[1] = p[fl1];
    b[0] = p[fr0];
    b[1] = p[fr1];
  };
  auto mfma_block = [&](const double (&a)[2], const double (&b)[2]) {
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
        acc.v[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc.v[ta][tb], 0, 0, 0);
  };

  // The pipelined loop contracts the DMA steps in PAIRS (buffer 0, buffer 1: the LDS addresses are immediates) and has ONE
  // body: the accumulators must stay in place.  An MFMA whose result register differs from its C operand leaves the C
  // registers dead, the register allocator parks a copy or a fragment load there, and on gfx950 the fp64 MFMA is still
  // streaming C in: the sum is corrupted (tools/mfma_srcc_lint.py finds such sites in the ISA; ROCm 7.2's hazard recogniser
  // does not pad them).  Copies appear where code paths with different register assignments meet -- peeled or
  // alternative loop bodies -- so there are none: an odd last step and the K tail are contracted after the loop by
  // straight-line code, the odd step still inside the pipeline (its rows are requested and its first fragments fetched by
  // the last pair).
  if (nS > 0) {
    double fa[2], fb[2], ga[2], gb[2];
    dma_load(0);
    __syncthreads();
    read_frag(0, 0, fa, fb);
    auto three_quads = [&](int buf) {  // quads 0 .. 2 of the step in `buf` (compile-time after inlining)
      read_frag(buf, 1, ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(buf, 2, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(buf, 3, ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(fa, fb);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto turn = [&](int nextbuf) {  // barrier, the next step's first fragments, then the last quad's MFMAs over both
      __syncthreads();  // vmcnt(0): the next step's rows have landed; barrier: every wave is done reading this buffer
      __builtin_amdgcn_sched_barrier(0);
      read_frag(nextbuf, 0, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(ga, gb);
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s + 1 < nS; s += 2) {
      dma_load(1);
      three_quads(0);
      turn(1);
      if (s + 2 < nS) dma_load(0);  // (scalar state only)
      three_quads(1);
      turn(0);
    }
    if (nS & 1) {  // the odd last step sits in buffer 0, its quad 0 in fa / fb
      three_quads(0);
      mfma_block(ga, gb);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (rem > 0) {
    // buffer nS & 1 is the one no step is using (every read of it precedes the loop's last barrier)
    __syncthreads();
    double *b = lds + (nS & 1) * BUF;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int p = tid + s * NT, row = p >> 5, cp = (p & 31) * 2;
      const int o = row * 64 + (cp ^ ((row & 1) << 4));
      *reinterpret_cast<double2_t *>(b + o) = tl[s];
      *reinterpret_cast<double2_t *>(b + TILE + o) = tr[s];
    }
    __syncthreads();
    double ta[4][2], tb[4][2];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const double *p = b + kq * 256;
      ta[kq][0] = p[fl0];
      ta[kq][1] = p[fl1];
      tb[kq][0] = p[fr0];
      tb[kq][1] = p[fr1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) mfma_block(ta[kq], tb[kq]);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- hand-over between the k-groups, then the epilogue on the sum ----
  const bool edge = (i0 + 64 > g.I) || (j0 + 64 > g.J);
  if constexpr (KG == 1) {
    const int64_t iw = i0 + wr * 32, jw = j0 + wc * 32;
    if constexpr (Epilogue::kDirect) {
      if constexpr (Epilogue::template direct_tile<2, 2>()) {
        if (!edge && epi.direct_ld() < kDirectMaxLd) {
          epi.template apply_direct<2, 2>(acc, iw, jw, lane, split, lds + wave * 128);
          return;
        }
      }
    }
    epi.template apply<2, 2>(acc, iw, jw, lane, wave, g.I, g.J, tile_i, split, lds);
  } else {
    // wave (grp, w) keeps rows 16 grp .. + 16 of its 32 x 32 block (acc.v[grp]) and hands the other half to its partner
    double *ex = lds + (grp * 4 + w) * 512;
    __syncthreads();  // (every wave has left the operand tiles; the last step's barrier precedes its last fragment reads)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) ex[(tb * 4 + r) * 64 + lane] = grp ? acc.v[0][tb][r] : acc.v[1][tb][r];
    __syncthreads();
    const double *ox = lds + ((grp ^ 1) * 4 + w) * 512;
    AccFrag<1, 2> fin;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double mine = grp ? acc.v[1][tb][r] : acc.v[0][tb][r];
        const double other = ox[(tb * 4 + r) * 64 + lane];
        fin.v[0][tb][r] = grp ? other + mine : mine + other;  // group 0's partial sum first
      }
    __syncthreads();  // the epilogue slabs overlap the exchange area
    const int v = (2 * wr + grp) * 2 + wc;  // wave index in the 4 x 2 arrangement of 16 x 32 blocks
    const int64_t iw = i0 + (2 * wr + grp) * 16, jw = j0 + wc * 32;
    if constexpr (Epilogue::kDirect) {
      if constexpr (Epilogue::template direct_tile<1, 2>()) {
        if (!edge && epi.direct_ld() < kDirectMaxLd) {
          epi.template apply_direct<1, 2>(fin, iw, jw, lane, split, lds + v * 128);
          return;
        }
      }
    }
    epi.template apply<1, 2>(fin, iw, jw, lane, v, g.I, g.J, tile_i, split, lds);
  }
#else
  (void)g, (void)epi;
#endif
}

}  // namespace plship
