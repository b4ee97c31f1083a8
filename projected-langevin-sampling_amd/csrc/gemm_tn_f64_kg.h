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

struct KgNoHook {
  __device__ __forceinline__ void operator()() const {}
};

// Rows [kbeg, kend) of both operands contracted into the wave's 32 x 32 block `acc` of its k-group's partial sum of the
// tile at (i0, j0).  acc is accumulated in place (the caller zeroes it); every wave of the workgroup must call this.
// hook(): called once by every thread right behind the FIRST workgroup barrier of the contraction, in front of which every
// wave has drained its vector-memory operations (s_waitcnt vmcnt(0): the first operand rows have landed -- and whatever the
// wave stored before the call has been performed).  The balanced triangular kernel signals a published partial sum from
// there, so that the store drain rides on the wait the first k-step needs anyway.
// pre(): called once by every thread right after the first operand rows have been REQUESTED (or at once if the k range holds
// no full super-step): work placed there -- it may hold workgroup barriers -- runs while those rows travel.
template <int KG, class Hook = KgNoHook, class Pre = KgNoHook>
__device__ __forceinline__ void kg_contract(const GemmShape &g, int64_t i0, int64_t j0, int64_t kbeg, int64_t kend, double *lds,
                                            AccFrag<2, 2> &acc, Hook hook = Hook{}, Pre pre = Pre{}) {
#if defined(__HIP_DEVICE_COMPILE__)
  using G = KgGeom<KG>;
  constexpr int NW = G::NW, NT = G::NT, SROWS = G::SROWS, TILE = G::TILE, BUF = G::BUF;
  typedef __attribute__((address_space(3))) void *lds_ptr_t;
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
  bool hooked = false;
  if (nS == 0) pre();
  if (nS > 0) {
    double fa[2], fb[2], ga[2], gb[2];
    dma_load(0);
    pre();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (what the barrier needs anyway: the rows have landed)
    __syncthreads();
    PLS_STAMP_AT(1);
    hook();
    hooked = true;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!hooked) hook();
    hooked = true;
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
  if constexpr (!std::is_same<Hook, KgNoHook>::value) {
    if (!hooked) {  // an empty k range
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      hook();
    }
  }
#else
  (void)g, (void)i0, (void)j0, (void)kbeg, (void)kend, (void)lds, (void)acc, (void)hook;
#endif
}

// Two k-groups: wave (grp, w) keeps rows 16 grp .. + 16 of its 32 x 32 block (acc.v[grp]) and hands the other half to its
// partner through LDS; afterwards every wave holds a 16 x 32 block `fin` of the SUM (group 0's partial sum first:
// deterministic) and the workgroup looks like 4 x 2 waves of 16 x 32 to the epilogues.  Ends with a barrier (the epilogue
// slabs overlap the exchange area).
// word / value (optional): thread 0 leaves `value` in the LDS word `word` between the first two barriers, so that every
// thread can read it behind the last one (the balanced triangular kernel passes on a flag it loaded before the call).
__device__ __forceinline__ void kg_handover(const AccFrag<2, 2> &acc, double *lds, AccFrag<1, 2> &fin, unsigned *word = nullptr,
                                            unsigned value = 0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 2, w = wave & 3;
  double *ex = lds + (grp * 4 + w) * 512;
  __syncthreads();  // (every wave has left the operand tiles; the last step's barrier precedes its last fragment reads)
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) ex[(tb * 4 + r) * 64 + lane] = grp ? acc.v[0][tb][r] : acc.v[1][tb][r];
  if (word && threadIdx.x == 0) word[0] = value;
  __syncthreads();
  const double *ox = lds + ((grp ^ 1) * 4 + w) * 512;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double mine = grp ? acc.v[1][tb][r] : acc.v[0][tb][r];
      const double other = ox[(tb * 4 + r) * 64 + lane];
      fin.v[0][tb][r] = grp ? other + mine : mine + other;  // group 0's partial sum first
    }
  __syncthreads();
}

// The epilogue of a two-group tile on the waves' 16 x 32 blocks of the sum.
// pre-drawn (optional): the noise pairs, the companion values and the two per-row constants of the wave's block, drawn /
// requested before the k-loop (Epilogue::pregen; 8 elements per lane).  Plain arrays and scalars, not a struct: passed
// around as one object they stayed in scratch memory.
template <class Epilogue>
__device__ __forceinline__ void kg_finish2(const GemmShape &g, const Epilogue &epi, const AccFrag<1, 2> &fin, int64_t i0,
                                           int64_t j0, int tile_i, int split, double *lds, const double (*pz)[8] = nullptr,
                                           const double (*px)[8] = nullptr, double pcl = 0.0, double pil = 0.0) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 2, w = wave & 3, wr = w >> 1, wc = w & 1;
  const bool edge = (i0 + 64 > g.I) || (j0 + 64 > g.J);
  const int v = (2 * wr + grp) * 2 + wc;  // wave index in the 4 x 2 arrangement of 16 x 32 blocks
  const int64_t iw = i0 + (2 * wr + grp) * 16, jw = j0 + wc * 32;
  if constexpr (epi_has_pregen<Epilogue>::value) {
    if (pz) {
      epi.template apply_pregen<1, 2>(fin, iw, jw, lane, v, g.I, g.J, tile_i, split, lds, *pz, *px, pcl, pil);
      return;
    }
  }
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

template <int KG, class Epilogue>
__global__ __launch_bounds__(256 * KG, 4) void gemm_tn_f64_kg_kernel(GemmShape g, Epilogue epi) {
  static_assert(KG == 1 || KG == 2, "epilogue hand-over is written for one or two k-groups");
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) double lds[];

  if constexpr (epi_has_prev<Epilogue>::value) {
    // workgroups BEHIND the tiles (launch_gemm_kg appends Epilogue::prev_chunks() of them): each finishes one 256-column chunk
    // of the PREVIOUS launch's energies -- sixteen loads per thread, two barriers, two stores -- on whatever CU has room,
    // beside the tiles' k-loops.  (As a prologue of the tiles of row 0 the same work made those four workgroups, and with
    // them the launch, 2.5 us late: a launch lasts as long as its slowest workgroup.)
    if ((int)blockIdx.x >= g.nti * g.ntj) {
      epi.prev_chunk((int)blockIdx.x - g.nti * g.ntj);
      return;
    }
  }
  PLS_STAMP_AT(0);
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

  AccFrag<2, 2> acc;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  // noise drawn in front of the k-loop (two k-groups; launch-uniform switch)
  constexpr bool kPregen = KG == 2 && epi_has_pregen<Epilogue>::value;
  [[maybe_unused]] double pz[8], px[8], pcl = 0.0, pil = 0.0;
  [[maybe_unused]] bool pregen = false;
  if constexpr (kPregen) pregen = epi.pregen_on();
  if constexpr (kPregen) {
    // the epilogue's own operands (particles, per-row constants) are REQUESTED here and land during the k-loop, and the
    // Philox / Box-Muller code of the wave's 16 x 32 block of the output (8 elements per lane) runs while the first operand
    // rows travel: the vector ALU has nothing else to do there, and behind the k-loop a memory round trip and ~560
    // instructions per wave are serial time of a launch that lasts 40 us
    kg_contract<KG>(g, i0, j0, kbeg, kend, lds, acc, KgNoHook{}, [&]() {
      if (pregen) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int grp = wave >> 2, w = wave & 3, wr = w >> 1, wc = w & 1;
        epi.pregen(i0 + (2 * wr + grp) * 16, j0 + wc * 32, (int)(threadIdx.x & 63), g.I, g.J, pz, px, pcl, pil);
      }
    });
  } else {
    kg_contract<KG>(g, i0, j0, kbeg, kend, lds, acc);
  }

  PLS_STAMP_AT(2);
  // ---- hand-over between the k-groups, then the epilogue on the sum ----
  if constexpr (KG == 1) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wr = (wave & 3) >> 1, wc = wave & 1;
    const bool edge = (i0 + 64 > g.I) || (j0 + 64 > g.J);
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
    AccFrag<1, 2> fin;
    kg_handover(acc, lds, fin);
    if constexpr (kPregen) {
      if (pregen) {
        kg_finish2(g, epi, fin, i0, j0, tile_i, split, lds, &pz, &px, pcl, pil);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PLS_STAMP_AT(3);
        return;
      }
    }
    kg_finish2(g, epi, fin, i0, j0, tile_i, split, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PLS_STAMP_AT(3);
  }
#else
  (void)g, (void)epi;
#endif
}

// ---- balanced triangular products --------------------------------------------------------------------------------
// A triangular k-major operand (tri = 1: L[k][i] == 0 for k > i; tri = 2: for k < i) gives tile row t a k range of
// (t + 1) * 64 (resp. K - 64 t) rows: with one 64 x 64 tile per workgroup and every workgroup resident at once (M = 1024,
// J = 1024: 256 tiles on 256 CUs), the launch lasts as long as its heaviest tile -- the FULL contraction, for half the
// flops (34 us at 0.40 of peak against 34 us for the full product).  No assignment of whole tiles to workgroups balances
// that (the heaviest tile alone is twice the average load), so the tile rows are paired (t, nti - 1 - t: together nti + 1
// row blocks whatever t is) and every pair is contracted by TWO workgroups with equal loads:
//     A: the first half of the pair's rows -- all of them rows of the heavy tile -- into a partial sum;
//     B: the rest of the heavy tile into a second partial sum, then the light tile, complete, with its epilogue.
// The heavy tile is finished by whichever of A and B arrives second: the first one leaves its 64 x 64 partial sum (32 KB, in
// the register layout: coalesced 512-byte rows) in a scratch slot and bumps the tile's flag; the second one finds the flag
// set, adds the other's partial sum to its own registers and runs the epilogue.  B reaches that point first by construction
// (its share of the heavy tile is the smaller one), so A looks at the flag BEFORE writing anything and normally skips its
// own write.  Nobody spins or waits, so the scheme needs no co-residency; the sum of two partial sums does not depend on
// who adds them (a + b == b + a), so the result is deterministic; the finisher clears the flag, so the scratch arrives zero
// at the next launch (captured graphs replay the same arguments).  Stream-K with a fixed two-way split, in short.
//   scratch: GemmShape::tri_flags (one word per pair and column tile, zero on entry, zero on exit) and tri_part (two 4096-
//   double slots per pair and column tile); kg_tri_scratch_bytes() below.  blockIdx -> (virtual row 2 pair + role, column
//   tile) goes through the usual XCD remap, which keeps A and B of a tile on ONE XCD (their ids are neighbours), so the
//   partial sum normally travels through that XCD's L2.
constexpr int kKgTriFlagBytes = 16384;  // 4096 flag words in front of the partial slots

__host__ __device__ inline size_t kg_tri_scratch_bytes(int64_t I, int64_t J) {
  const int64_t pairs = ((I + 63) / 64 + 1) / 2, ntj = (J + 63) / 64;
  return (size_t)kKgTriFlagBytes + (size_t)pairs * ntj * 2 * 4096 * sizeof(double);
}

// The partial sums of a heavy tile that two workgroups share.
//
// Visibility (per-XCD L2s are not coherent with each other, a CU's L1 is never refreshed by another CU's stores): an
// agent-scope release / acquire pair -- __threadfence() -- writes back and invalidates whole caches (buffer_wbl2 /
// buffer_inv), and with 256 workgroups doing that in the middle of their k-loops the launch took 99 us instead of 35 (the
// operand panels of every other workgroup on the XCD went with it).  Instead every byte of a partial sum is stored
// write-through (sc1, 16 bytes per lane: a wave instruction writes eight whole 128-byte lines), every storing wave drains
// its stores (s_waitcnt vmcnt(0)), the workgroup's barrier collects the waves, and ONE lane bumps the flag with an
// agent-scope atomic; the reader learns of the partial sum from the value its own atomic add returned (or from an sc1 load
// of the flag) and loads every byte of it with sc1 loads, which are served past the L1: no cache-wide operation anywhere.
// The hand-over between workgroups below (write-through `sc1` stores, a drained store queue, one relaxed agent-scope atomic,
// `sc1` loads past the L1) is written against the cache hierarchy of gfx942 / gfx950 (per-XCD L2s that are not coherent with
// each other, write-through vector L1s): another target needs its own protocol, not a silent recompile.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the inter-workgroup hand-over of libplship is written for gfx942 / gfx950"
#endif
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int kKgSc1 = 16;  // aux bit of the raw buffer instructions: sc1

// slot layout: [register pair 0 .. 3][thread 0 .. 511] of 16 bytes
__device__ __forceinline__ void kg_tri_publish(const AccFrag<1, 2> &fin, double *mine) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 4096 * 8, 0x00020000);
  const int voff = threadIdx.x * 16;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double2_t v{fin.v[0][tb][2 * h], fin.v[0][tb][2 * h + 1]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, voff, (tb * 2 + h) * 8192, kKgSc1);
    }
#else
  (void)fin, (void)mine;
#endif
}

// fin (+)= the partial sum in `slot` (every byte by an sc1 load)
template <bool ADD>
__device__ __forceinline__ void kg_tri_fetch(AccFrag<1, 2> &fin, const double *slot) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(slot), 0, 4096 * 8, 0x00020000);
  const int voff = threadIdx.x * 16;
  double2_t o[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) o[p] = __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rt, voff, p * 8192, kKgSc1));
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (ADD) {
        fin.v[0][tb][2 * h] += o[tb * 2 + h][0];
        fin.v[0][tb][2 * h + 1] += o[tb * 2 + h][1];
      } else {
        fin.v[0][tb][2 * h] = o[tb * 2 + h][0];
        fin.v[0][tb][2 * h + 1] = o[tb * 2 + h][1];
      }
    }
#else
  (void)fin, (void)slot;
#endif
}

// drain this wave's stores, collect the workgroup, bump the flag (one lane); returns the flag's previous value to every thread
__device__ __forceinline__ unsigned kg_tri_signal_and_wait(unsigned *flag, unsigned *word) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have been performed
  __syncthreads();
  if (threadIdx.x == 0) word[0] = __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned seen = word[0];
  __syncthreads();
  return seen;
#else
  (void)flag, (void)word;
  return 0;
#endif
}

template <class Epilogue>
__global__ __launch_bounds__(512, 4) void gemm_tn_f64_kg_tri_kernel(GemmShape g, Epilogue epi) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ unsigned word[1];  // thread 0's view of the tile's flag for the whole workgroup
  const int tid = threadIdx.x;
  const int pairs = (g.nti + 1) >> 1;
  int vrow, tile_j;
  gemm_tile_coords(blockIdx.x, 2 * pairs, g.ntj, vrow, tile_j);
  const int pair = vrow >> 1, role = vrow & 1;
  const int64_t j0 = (int64_t)tile_j * 64;
  // the pair's tile rows and their k ranges; "heavy" has the longer range
  const int t_lo = pair, t_hi = g.nti - 1 - pair;
  const int heavy = (g.tri == 1) ? t_hi : t_lo, light = (g.tri == 1) ? t_lo : t_hi;
  auto k_range = [&](int t, int64_t &kb, int64_t &ke) {
    kb = 0;
    ke = g.K;
    const int64_t i0 = (int64_t)t * 64;
    if (g.tri == 1 && i0 + 64 < ke) ke = i0 + 64;
    if (g.tri == 2) kb = i0 < g.K ? i0 : g.K;
  };
  int64_t hb, he, lb, le;
  k_range(heavy, hb, he);
  k_range(light, lb, le);
  const bool have_light = light != heavy;
  const int64_t len_h = he - hb, len_l = have_light ? le - lb : 0;
  int64_t a_len = ((len_h + len_l + 1) / 2 + 31) / 32 * 32;  // A's share: half of the pair's rows, whole super-steps
  if (a_len > len_h) a_len = len_h;
  const bool shared = a_len < len_h;  // B holds a part of the heavy tile
  unsigned *flag = g.tri_flags + ((int64_t)pair * g.ntj + tile_j);
  double *slot_a = g.tri_part + ((int64_t)pair * g.ntj + tile_j) * 2 * 4096, *slot_b = slot_a + 4096;
  const int64_t ih = (int64_t)heavy * 64, il = (int64_t)light * 64;

  AccFrag<2, 2> acc;
  AccFrag<1, 2> fin;
  auto zero_acc = [&]() {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
  };
  auto finish_heavy_with = [&](const double *theirs) {  // fin += the other partial sum; epilogue; flag back to zero
    kg_tri_fetch<true>(fin, theirs);
    if (tid == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    kg_finish2(g, epi, fin, ih, j0, heavy, 0, lds);
  };

  if (role == 0) {
    // ---- A: the first a_len rows of the heavy tile ----
    zero_acc();
    kg_contract<2>(g, ih, j0, hb, hb + a_len, lds, acc);
    unsigned pre = 0;
    if (shared && tid == 0) pre = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (flies under the hand-over)
    kg_handover(acc, lds, fin, word, pre);
    if (!shared) {
      kg_finish2(g, epi, fin, ih, j0, heavy, 0, lds);
      return;
    }
    if (word[0]) {  // the usual case: B's partial sum is there (its share of the heavy tile is the smaller one)
      finish_heavy_with(slot_b);
      return;
    }
    __syncthreads();  // (everybody has read the word)
    kg_tri_publish(fin, slot_a);
    if (kg_tri_signal_and_wait(flag, word)) finish_heavy_with(slot_b);  // B arrived in between
    return;
  }

  // ---- B: the rest of the heavy tile, then the light tile ----
  if (shared) {
    zero_acc();
    kg_contract<2>(g, ih, j0, hb + a_len, he, lds, acc);
    kg_handover(acc, lds, fin);
    kg_tri_publish(fin, slot_b);
    if (!have_light) {  // (the middle row of an odd count pairs with itself)
      if (kg_tri_signal_and_wait(flag, word)) finish_heavy_with(slot_a);
      return;
    }
  }
  if (have_light) {
    unsigned before = 0;  // thread 0: what the flag held before this workgroup's add
    zero_acc();
    // the flag moves behind the light tile's first barrier: every wave has drained the stores of its partial sum there, and
    // nobody waits for the atomic's answer until the light tile is done
    kg_contract<2>(g, il, j0, lb, le, lds, acc, [&]() {
      if (shared && tid == 0) before = __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    kg_handover(acc, lds, fin, word, before);
    kg_finish2(g, epi, fin, il, j0, light, 0, lds);
    if (shared && word[0]) {  // A had been there first (rare): its partial sum waits in its slot, this one's in slot_b
      __syncthreads();  // (the light tile's epilogue slabs)
      kg_tri_fetch<false>(fin, slot_b);
      finish_heavy_with(slot_a);
    }
  }
#else
  (void)g, (void)epi;
#endif
}

}  // namespace plship
