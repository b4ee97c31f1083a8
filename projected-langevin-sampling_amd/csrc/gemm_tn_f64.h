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

#include <type_traits>

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
  int64_t kchunk;  // split-K: block (x, y) contracts k in [y * kchunk, min(K, (y + 1) * kchunk)); gridDim.y slabs
  int tri;         // 1: L[k][i] == 0 for k > i (upper-triangular k-major operand), a tile contracts k < i0 + BI only;
                   // 2: L[k][i] == 0 for k < i (lower-triangular), a tile contracts k >= i0 only
  double *tri_part;     // balanced triangular products (gemm_tn_f64_kg_tri_kernel): partial-sum slots and flag words of
  unsigned *tri_flags;  // the caller's scratch; NULL elsewhere
#ifdef PLS_STAMP
  unsigned long long *stamps;  // diagnostic build only: 4 s_memtime stamps per workgroup (never read by the kernel)
#endif
};

#ifdef PLS_STAMP
// 6 slots per workgroup: 4 s_memtime stamps, HW_ID, XCC_ID
#define PLS_STAMP_AT(slot)                                                                             \
  do {                                                                                                 \
    if (g.stamps && threadIdx.x == 0) {                                                                \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
      unsigned long long *w_ = g.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6;           \
      w_[(slot)] = t_;                                                                                 \
      if ((slot) == 0) {                                                                               \
        unsigned h_, x_;                                                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h_));                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_));                              \
        w_[4] = h_;                                                                                    \
        w_[5] = x_;                                                                                    \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
#else
#define PLS_STAMP_AT(slot) do {} while (0)
#endif

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

// The k-loop, specialised at compile time on
//   VEC : operands are 16-B aligned with even leading dimensions -> one global_load_dwordx4 per pair
//   EDGE: the tile overhangs I or J -> overhanging lanes read column 0 of their row (always inside the matrix)
//         and are zeroed when the pair is written to LDS (after the MFMAs, so the loads stay in flight)
// so that the steady-state loop contains no branch and no use of a loaded value before the MFMAs of the step.
//   DMA : full k-steps are copied global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: no staging VGPRs, no
//         ds_write instructions); one wave-instruction fills one 1-KiB k-row of a 128-wide tile, so the row pad stays.
//         Overhanging lanes then deposit real (finite or not) matrix data of other columns: harmless, because output
//         (i, j) only ever combines column i of L with column j of R and the epilogue drops i >= I, j >= J.  The K tail
//         (rows that must read as zero) always goes through registers.
//   TIU / AST / a_off: the wave contracts its first TIU (<= TI) row blocks, AST columns of the L tile apart, starting at
//         column a_off of the tile (the usual wave block: TIU = TI, AST = 16, a_off = the wave's row offset; the
//         row-interleaved tiles of gemm_tn_f64_rows.h: AST = 32, a_off = 16 * wave row, TIU = the blocks that hold rows)
template <int BI, int BJ, int WI, int WJ, int BK, bool VEC, bool EDGE, bool DMA, int TIU, int AST, int TI, int TJ>
__device__ __forceinline__ void gemm_tn_mainloop(const GemmShape &g, int64_t i0, int64_t j0, double *lds,
                                                 AccFrag<TI, TJ> &acc, int a_off) {
  static_assert(TIU >= 1 && TIU <= TI && (AST == 16 || AST == 32), "row blocks of the wave");
  constexpr int NW = (BI / WI) * (BJ / WJ);
  constexpr int NT = NW * 64;
  constexpr int PAD = 16;
  constexpr int SL = BI + PAD, SR = BJ + PAD;
  constexpr int LROWS = NT / (BI / 2);  // k-rows of the L tile covered by one pass of all threads
  constexpr int RROWS = NT / (BJ / 2);
  // a narrow L tile (BI = 16: the 16-row remainder configuration) is covered by the first BK * BI / 2 threads alone
  constexpr bool LPART = LROWS > BK;
  constexpr int LPASS = LPART ? 1 : BK / LROWS, RPASS = BK / RROWS;
  static_assert((LPART || BK % LROWS == 0) && BK % RROWS == 0 && BK % 4 == 0, "tile/thread mismatch");
  double *Ls = lds;                // [2][BK][SL]
  double *Rs = lds + 2 * BK * SL;  // [2][BK][SR]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = a_off, wj = (wave % (BJ / WJ)) * WJ;
  const int q = lane >> 4, c16 = lane & 15;

  const int lcol = (tid % (BI / 2)) * 2, lrow = tid / (BI / 2);
  const int rcol = (tid % (BJ / 2)) * 2, rrow = tid / (BJ / 2);
  const int64_t lrem = EDGE ? g.I - i0 - lcol : 2;  // valid columns from this thread's first column
  const int64_t rrem = EDGE ? g.J - j0 - rcol : 2;
  const bool l0 = lrem >= 1, l1 = lrem >= 2, r0 = rrem >= 1, r1 = rrem >= 2;
  const double *Lq = g.L + (l0 ? i0 + lcol : 0);
  const double *Rq = g.R + (r0 ? j0 + rcol : 0);

  double2_t lreg[LPASS], rreg[RPASS];
  bool lkin[LPASS], rkin[RPASS];  // only meaningful for the K-tail step

  typedef __attribute__((address_space(3))) void *lds_ptr_t;
  // DMA addressing (buffer_load_dwordx4 ... offen lds): a 128-bit buffer descriptor whose base is the wave-uniform
  // address of the step's first k-row (rebuilt with two scalar adds per step, so any matrix size works), the row
  // inside the step as an SGPR offset, and ONE loop-invariant 32-bit lane offset in a VGPR: no VALU in the k-loop.
  static_assert(!DMA || (BI == 128 && BJ == 128), "one wave-instruction must cover exactly one k-row");
  const int wrow = __builtin_amdgcn_readfirstlane(lrow);  // = wave index: uniform
  const int loff = (int)((l0 ? i0 + lcol : 0) * 8), roff = (int)((r0 ? j0 + rcol : 0) * 8);
  int lsoff[LPASS], rsoff[RPASS];
#pragma unroll
  for (int p = 0; p < LPASS; ++p) lsoff[p] = (int)((int64_t)(wrow + p * LROWS) * g.ldl * 8);
#pragma unroll
  for (int p = 0; p < RPASS; ++p) rsoff[p] = (int)((int64_t)(wrow + p * RROWS) * g.ldr * 8);
  const char *lnext = reinterpret_cast<const char *>(g.L), *rnext = reinterpret_cast<const char *>(g.R);
  const int64_t lstep = (int64_t)BK * g.ldl * 8, rstep = (int64_t)BK * g.ldr * 8;
  auto dma_load = [&](int buf) {  // copies the NEXT full k-step (lnext/rnext run one step ahead) into `buf`
    (void)sizeof(lds_ptr_t);
#if defined(__HIP_DEVICE_COMPILE__)  // (the buffer-resource type and builtins exist in the device pass only)
    const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(lnext), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rnext), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int p = 0; p < LPASS; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lr, (lds_ptr_t)(Ls + buf * BK * SL + (wrow + p * LROWS) * SL), 16, loff, lsoff[p], 0,
                                               0);
#pragma unroll
    for (int p = 0; p < RPASS; ++p)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr_t)(Rs + buf * BK * SR + (wrow + p * RROWS) * SR), 16, roff, rsoff[p], 0,
                                               0);
    lnext += lstep;
    rnext += rstep;
#else
    (void)buf, (void)loff, (void)roff, (void)lnext, (void)rnext, (void)lstep, (void)rstep, (void)lsoff, (void)rsoff;
#endif
  };

  auto load_global = [&](int64_t k0, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      if (LPART && lrow >= BK) break;  // (this thread has no element of the narrow L tile)
      int64_t k = k0 + lrow + p * LROWS;
      lkin[p] = true;
      if (TAIL) {
        lkin[p] = k < g.K;
        k = lkin[p] ? k : g.K - 1;
      }
      const double *src = Lq + k * g.ldl;
      if (VEC) {
        lreg[p] = *reinterpret_cast<const double2_t *>(src);
      } else {
        lreg[p].x = src[0];
        lreg[p].y = l1 ? src[1] : 0.0;
      }
    }
#pragma unroll
    for (int p = 0; p < RPASS; ++p) {
      int64_t k = k0 + rrow + p * RROWS;
      rkin[p] = true;
      if (TAIL) {
        rkin[p] = k < g.K;
        k = rkin[p] ? k : g.K - 1;
      }
      const double *src = Rq + k * g.ldr;
      if (VEC) {
        rreg[p] = *reinterpret_cast<const double2_t *>(src);
      } else {
        rreg[p].x = src[0];
        rreg[p].y = r1 ? src[1] : 0.0;
      }
    }
  };
  auto store_lds = [&](int buf, auto tail_tag) {
    constexpr bool TAIL = decltype(tail_tag)::value;
    double *l = Ls + buf * BK * SL;
    double *r = Rs + buf * BK * SR;
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      if (LPART && lrow >= BK) break;
      double2_t v = lreg[p];
      if (EDGE || TAIL) {
        v.x = (l0 && lkin[p]) ? v.x : 0.0;
        v.y = (l1 && lkin[p]) ? v.y : 0.0;
      }
      *reinterpret_cast<double2_t *>(l + (lrow + p * LROWS) * SL + lcol) = v;
    }
#pragma unroll
    for (int p = 0; p < RPASS; ++p) {
      double2_t v = rreg[p];
      if (EDGE || TAIL) {
        v.x = (r0 && rkin[p]) ? v.x : 0.0;
        v.y = (r1 && rkin[p]) ? v.y : 0.0;
      }
      *reinterpret_cast<double2_t *>(r + (rrow + p * RROWS) * SR + rcol) = v;
    }
  };
  // Fragments are double-buffered in registers: the ds_reads of k-quad q+1 are issued before the MFMAs of
  // k-quad q, so their LDS latency hides under 16 (TI*TJ) 64-cycle MFMAs instead of stalling the wave.
  // nq: k-quads of the step that hold data (4, except in the K tail: a rank of 129 contracts 33 quads, not 36)
  auto compute = [&](int buf, int nq = BK / 4) {
    const double *l = Ls + buf * BK * SL + q * SL + wi + c16;
    const double *r = Rs + buf * BK * SR + q * SR + wj + c16;
    double a[2][TIU], b[2][TJ];
#pragma unroll
    for (int t = 0; t < TIU; ++t) a[0][t] = l[t * AST];
#pragma unroll
    for (int t = 0; t < TJ; ++t) b[0][t] = r[t * 16];
#pragma unroll
    for (int kq = 0; kq < BK / 4; ++kq) {
      if (kq >= nq) break;  // wave-uniform
      const int cur = kq & 1, nxt = cur ^ 1;
      if (kq + 1 < BK / 4) {
#pragma unroll
        for (int t = 0; t < TIU; ++t) a[nxt][t] = l[(kq + 1) * 4 * SL + t * AST];
#pragma unroll
        for (int t = 0; t < TJ; ++t) b[nxt][t] = r[(kq + 1) * 4 * SR + t * 16];
      }
      // fence the prefetch: left alone, the scheduler sinks these reads to just in front of the MFMAs that use them
      // (ds_read, s_waitcnt lgkmcnt(0), MFMAs, ds_read, ...), i.e. the LDS latency is exposed once per k-quad -- with the
      // 2 x 2 tiles of the small configuration that is once per 256 MFMA-cycles
#ifndef PLS_NO_COMPUTE_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int ta = 0; ta < TIU; ++ta)
#pragma unroll
        for (int tb = 0; tb < TJ; ++tb)
          acc.v[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][ta], b[cur][tb], acc.v[ta][tb], 0, 0, 0);
#ifndef PLS_NO_COMPUTE_FENCE
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  using full_t = std::integral_constant<bool, false>;
  using tail_t = std::integral_constant<bool, true>;
  const int64_t nk = (g.K + BK - 1) / BK;
  const int64_t nk_full = g.K / BK;
  if (g.K > 0) {  // (K == 0: acc stays zero; the clamp k = K-1 would be out of bounds)
    if (nk_full > 0) {
      if (DMA) {
        dma_load(0);
      } else {
        load_global(0, full_t{});
        store_lds(0, full_t{});
      }
    } else {
      load_global(0, tail_t{});
      store_lds(0, tail_t{});
    }
  }
  __syncthreads();
  PLS_STAMP_AT(1);

  int64_t kt = 0;
  if (DMA) {
    // Rotated pipeline.  Per step: issue the DMA of the next tile; run k-quads 0..2 (their fragments prefetched one
    // quad ahead); drain the DMA and cross the barrier; fetch the NEXT step's first fragments from the other buffer;
    // only then issue the last quad's 16 MFMAs, which cover the barrier skew and that fetch's LDS latency.
    // (Issuing the DMA of step k+2 right after the barrier of step k -- a full step of latency budget instead of
    // three quarters -- measured 3% SLOWER, tools/ab_gemm.py: the eight DMA issues delay the fragment fetch the last
    // quad's MFMAs wait for.)
    auto read_frag = [&](int buf, int kq, double (&a)[TIU], double (&b)[TJ]) {
      const double *l = Ls + buf * BK * SL + (kq * 4 + q) * SL + wi + c16;
      const double *r = Rs + buf * BK * SR + (kq * 4 + q) * SR + wj + c16;
#pragma unroll
      for (int t = 0; t < TIU; ++t) a[t] = l[t * AST];
#pragma unroll
      for (int t = 0; t < TJ; ++t) b[t] = r[t * 16];
    };
    auto mfma_block = [&](const double (&a)[TIU], const double (&b)[TJ]) {
#pragma unroll
      for (int ta = 0; ta < TIU; ++ta)
#pragma unroll
        for (int tb = 0; tb < TJ; ++tb)
          acc.v[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ta], b[tb], acc.v[ta][tb], 0, 0, 0);
    };
    static_assert(BK == 16, "the rotated loop is written for 4 k-quads per step");
    if (nk_full > 1) {
      double fa[TIU], fb[TJ], ga[TIU], gb[TJ];
      read_frag(0, 0, fa, fb);
      // first = true: the very first 16 MFMAs of the tile take a literal zero as their C operand instead of reading
      // zero-initialised accumulators (64 v_mov per wave that would each queue behind a 64-cycle MFMA of the
      // co-resident workgroup: ~4k cycles of prologue per tile)
      auto body = [&](auto buf_tag, auto first_tag) {  // buf is a compile-time constant: LDS addresses fold
        constexpr int buf = decltype(buf_tag)::value;
        constexpr bool first = decltype(first_tag)::value;
        dma_load(buf ^ 1);
        read_frag(buf, 1, ga, gb);
        if constexpr (first) {
#pragma unroll
          for (int ta = 0; ta < TIU; ++ta)
#pragma unroll
            for (int tb = 0; tb < TJ; ++tb)
              acc.v[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ta], fb[tb], double4_t{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        } else {
          mfma_block(fa, fb);  // quad 0
        }
        read_frag(buf, 2, fa, fb);
        mfma_block(ga, gb);  // quad 1
        read_frag(buf, 3, ga, gb);
        mfma_block(fa, fb);  // quad 2
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // vmcnt(0): the DMA of step k+1 has landed; barrier: every wave is done reading `buf`
        __builtin_amdgcn_sched_barrier(0);
        read_frag(buf ^ 1, 0, fa, fb);      // next step's quad 0
        __builtin_amdgcn_sched_barrier(0);  // keep that fetch AHEAD of the 16 MFMAs that hide its latency
        mfma_block(ga, gb);                 // quad 3 of this step
      };
      using i0_t = std::integral_constant<int, 0>;
      using i1_t = std::integral_constant<int, 1>;
      // steps 0 .. nk_full - 2 run through body (each prefetches the following full step); the first one is peeled
      body(i0_t{}, std::true_type{});
      for (kt = 1; kt + 2 < nk_full; kt += 2) {
        body(i1_t{}, std::false_type{});
        body(i0_t{}, std::false_type{});
      }
      if (kt + 1 < nk_full) {
        body(i1_t{}, std::false_type{});
        ++kt;
      }
      // fa/fb hold quad 0 of step kt (the last full step, or the one before the tail): finish it here
      {
        const int buf = (int)(kt & 1);
        const bool more = kt + 1 < nk;
        if (more) load_global((kt + 1) * BK, tail_t{});
        read_frag(buf, 1, ga, gb);
        mfma_block(fa, fb);
        read_frag(buf, 2, fa, fb);
        mfma_block(ga, gb);
        read_frag(buf, 3, ga, gb);
        mfma_block(fa, fb);
        mfma_block(ga, gb);
        if (more) store_lds(buf ^ 1, tail_t{});
        __syncthreads();
        ++kt;
      }
    }
  } else {
    for (; kt + 1 < nk_full; ++kt) {  // steady state: the next step is a full one
      const int buf = (int)(kt & 1);
      load_global((kt + 1) * BK, full_t{});
      compute(buf);
      store_lds(buf ^ 1, full_t{});
      __syncthreads();
    }
  }
  for (; kt < nk; ++kt) {  // last full step and the K tail
    const int buf = (int)(kt & 1);
    const bool more = kt + 1 < nk;
    if (more) load_global((kt + 1) * BK, tail_t{});
    const int64_t kleft = g.K - kt * BK;
    compute(buf, kleft >= BK ? BK / 4 : (int)((kleft + 3) >> 2));
    if (more) store_lds(buf ^ 1, tail_t{});
    __syncthreads();
  }
}

// An epilogue may carry a PROLOGUE: work some workgroups do at the START of the launch for the launch before it (the Langevin
// epilogue finishes the previous step's energies there, under the landing of the first operand rows: prev_owner / prev_reduce /
// prev_store).  Detected by the member's presence.
template <class E, class = void>
struct epi_has_prev : std::false_type {};
template <class E>
struct epi_has_prev<E, std::void_t<decltype(&E::prev_owner)>> : std::true_type {};

// Epilogues whose per-element noise does not depend on the contraction can draw it BEFORE the k-loop (the k-split kernel of
// gemm_tn_f64_kg.h does, under the landing of its first operand rows: pregen_on / pregen_pairs / apply_pregen).
template <class E, class = void>
struct epi_has_pregen : std::false_type {};
template <class E>
struct epi_has_pregen<E, std::void_t<decltype(&E::pregen_on)>> : std::true_type {};

// largest leading dimension (doubles) the direct epilogue addresses with 32-bit byte offsets (67 rows * ld * 8 < 2^31)
constexpr int64_t kDirectMaxLd = (int64_t)1 << 21;

// One output tile (tile_i, tile_j) of the contraction: k-loop + epilogue.
template <int BI, int BJ, int WI, int WJ, int BK, class Epilogue>
__device__ __forceinline__ void gemm_tile(GemmShape g, const Epilogue &epi, int tile_i, int tile_j, double *lds) {
  constexpr int TI = WI / 16, TJ = WJ / 16;
  static_assert((BI / WI) * (BJ / WJ) * 2 * 16 * (WJ + 2) <= 2 * BK * ((BI + 16) + (BJ + 16)),
                "the epilogue slabs reuse the operand tiles' LDS");
  const int64_t i0 = (int64_t)tile_i * BI, j0 = (int64_t)tile_j * BJ;
  const int split = blockIdx.y;
  if (gridDim.y > 1) {  // split-K slab: shift the operands to this block's k-range
    const int64_t k0 = (int64_t)split * g.kchunk;
    g.L += k0 * g.ldl;
    g.R += k0 * g.ldr;
    g.K = (g.K - k0 < g.kchunk) ? g.K - k0 : g.kchunk;
  }

  if (g.tri == 1 && i0 + BI < g.K) g.K = i0 + BI;  // L[k][i] == 0 for k > i: the rows below the tile's last column are zero
  if (g.tri == 2) {  // L[k][i] == 0 for k < i: the rows above the tile's first column are zero
    const int64_t k0 = i0 < g.K ? i0 : g.K;
    g.L += k0 * g.ldl;
    g.R += k0 * g.ldr;
    g.K -= k0;
  }
  PLS_STAMP_AT(0);
  AccFrag<TI, TJ> acc;
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TJ; ++b) acc.v[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};

  const int wi0 = (int)(threadIdx.x >> 6) / (BJ / WJ) * WI;  // the wave's first row inside the tile
  // vector path: every row start is 16-B aligned; a pair that straddles the I/J edge stays inside its row's
  // padding because the leading dimension is even
  const bool vec = ((g.ldl & 1) == 0) && ((reinterpret_cast<uintptr_t>(g.L) & 15) == 0) && ((g.ldr & 1) == 0) &&
                   ((reinterpret_cast<uintptr_t>(g.R) & 15) == 0);
  const bool edge = (i0 + BI > g.I) || (j0 + BJ > g.J);
  constexpr bool kDma = (BI == 128 && BJ == 128);
  if (vec) {
    if (!edge)
      gemm_tn_mainloop<BI, BJ, WI, WJ, BK, true, false, kDma, TI, 16>(g, i0, j0, lds, acc, wi0);
    else
      gemm_tn_mainloop<BI, BJ, WI, WJ, BK, true, true, kDma, TI, 16>(g, i0, j0, lds, acc, wi0);
  } else {
    gemm_tn_mainloop<BI, BJ, WI, WJ, BK, false, true, false, TI, 16>(g, i0, j0, lds, acc, wi0);
  }

  PLS_STAMP_AT(2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wi = (wave / (BJ / WJ)) * WI, wj = (wave % (BJ / WJ)) * WJ;
  if constexpr (Epilogue::kDirect) {
    if constexpr (Epilogue::template direct_tile<TI, TJ>()) {
      // interior tile: registers -> global without the LDS transpose (see epilogue_direct)
      if (!edge && epi.direct_ld() < kDirectMaxLd) {
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        epi.template apply_direct<TI, TJ>(acc, i0 + (wu / (BJ / WJ)) * WI, j0 + (wu % (BJ / WJ)) * WJ, lane, split, lds + wu * 128);
        PLS_STAMP_AT(3);
        return;
      }
    }
  }
  epi.template apply<TI, TJ>(acc, i0 + wi, j0 + wj, lane, wave, g.I, g.J, tile_i, split, lds);
  PLS_STAMP_AT(3);
}

template <int BI, int BJ, int WI, int WJ, int BK, int MINW, class Epilogue>
__global__ __launch_bounds__((BI / WI) * (BJ / WJ) * 64, MINW) void gemm_tn_f64_kernel(GemmShape g, Epilogue epi) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  int tile_i, tile_j;
  if constexpr (Epilogue::kTag == 1) {  // (only the plain-store contraction is ever launched with a triangular operand)
   if (g.tri) {
    // Triangular operand: tile row t contracts (t + 1) * BI rows, so the rows are far from equal -- and the two workgroups
    // that share a CU get the SAME row (their block ids differ by a multiple of 8 * nti), i.e. a CU holding two copies of
    // the last row sets the launch time (0.24 ms for L xi at M = 1024 against 0.26 ms for the full product).  A workgroup
    // therefore takes tile rows t and nti - 1 - t one after the other: every workgroup contracts nti + 1 row-blocks.
    const int pairs = (g.nti + 1) >> 1;
    gemm_tile_coords(blockIdx.x, pairs, g.ntj, tile_i, tile_j);
    gemm_tile<BI, BJ, WI, WJ, BK>(g, epi, tile_i, tile_j, lds);
    const int other = g.nti - 1 - tile_i;
    if (other != tile_i) {
      __syncthreads();  // (the first tile's epilogue may still be reading its LDS slabs)
      gemm_tile<BI, BJ, WI, WJ, BK>(g, epi, other, tile_j, lds);
    }
    return;
   }
  }
  gemm_tile_coords(blockIdx.x, g.nti, g.ntj, tile_i, tile_j);
  if constexpr (epi_has_prev<Epilogue>::value) {
    // (workgroup-uniform.  Two workgroups share a CU here and the other one covers this prologue; workgroups of their own behind
    // the tiles -- what the k-split kernel does -- find no room beside 512 resident tiles and run as a tail: +2.5 us at
    // J = 8192 against +1.8 us this way)
    if (epi.prev_owner(tile_i, tile_j)) epi.prev_chunk(epi.prev_chunk_of(tile_j));
  }
  gemm_tile<BI, BJ, WI, WJ, BK>(g, epi, tile_i, tile_j, lds);
}

// ---- epilogues ------------------------------------------------------------------------------------------------
// The accumulator block of a wave leaves the registers through LDS, 16 rows at a time: the MFMA layout (4 rows x
// 16 columns per register) is written with compile-time register indices, then read back row-wise so that
//   * a lane owns ONE column and walks down the rows in a run-time loop (the per-element code -- cost derivative,
//     Box-Muller -- exists once, not once per accumulator register),
//   * every global store / load of the epilogue is a contiguous row segment of WJ doubles,
//   * rows i and i+4 (one Philox pair, philox.h) are handed to the functor together.
// fn(i_lo, j, v_lo, hi_valid, v_hi, rc) is called for every column j < J and row pair (i_lo, i_lo + 4) with i_lo < I.
// Per-row constants (y_i, c_i, 1/lambda_i) are loaded ONCE per wave into lane registers (lane l <- row iw + l, see
// load_row_constants) and handed over as rc = {k0[i_lo], k0[i_lo+4], k1[i_lo], k1[i_lo+4]}: a cross-lane read
// (done with all lanes active, before the edge predicate) instead of a dependent global load in every iteration of the
// row loop -- that chain cost ~9 us per tile.
constexpr int EPI_PAD = 2;  // doubles; keeps the 4 rows a ds_write_b64 touches on different banks

template <int WJ>
constexpr int epi_lds_doubles_per_wave() { return 2 * 16 * (WJ + EPI_PAD); }  // accumulator slab + companion slab

struct RowConsts {
  double k0_lo, k0_hi, k1_lo, k1_hi;
  double x_lo, x_hi;  // companion matrix X at (i_lo, j) and (i_lo + 4, j) (0 when X is not staged)
  int it;             // iteration of the slab's row loop (a compile-time constant in the functor once the loop is unrolled)
};

// X / ldx (optional): a companion matrix addressed like the output (the particles U of the Langevin and energy
// epilogues).  Its 16 x WJ slab is fetched with 16 row-wise coalesced loads that are ALL in flight together, one slab
// ahead of its use, and handed to the row loop through LDS -- a load inside the row loop exposes the memory latency
// once per iteration.  (Measured gain: 2-3 us of the 283 us fast-path step; the rest of that epilogue is instruction
// count)
// UNROLL: unroll factor of the row loop (2 interleaves two independent row pairs; measured neutral, left at 1).
template <int TI, int TJ, int NCONST = 2, int UNROLL = 1, class Fn>
__device__ __forceinline__ void epilogue_row_pairs(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave,
                                                   int64_t I, int64_t J, double *lds, double k0, double k1, Fn &&fn,
                                                   const double *X = nullptr, int64_t ldx = 0, const double *xpre = nullptr) {
  // xpre (optional, TI == 1): the lane's 16 / RPI companion values of slab 0, loaded by the caller earlier (in load_x's order)
  // NCONST: how many of the per-row constant registers (k0, k1) the functor uses (their cross-lane reads are skipped otherwise)
  constexpr int WJ = TJ * 16;
  constexpr int STRIDE = WJ + EPI_PAD;
  constexpr int RPI = 64 / WJ;  // row pairs handled per iteration by the 64 lanes (1 for WJ = 64, 2 for WJ = 32)
  constexpr int NXL = 16 / RPI;  // companion loads per lane and slab (each covers RPI rows)
  static_assert(WJ == 64 || WJ == 32, "wave tile width");
  double *w = lds + wave * epi_lds_doubles_per_wave<WJ>();
  double *wx = w + 16 * STRIDE;
  const int q = lane >> 4, c16 = lane & 15;
  const int col = lane % WJ, sub = lane / WJ;
  const int64_t j = jw + col;
  double xr[NXL];
  auto load_x = [&](int ta) {
#pragma unroll
    for (int k = 0; k < NXL; ++k) {
      const int64_t i = iw + ta * 16 + k * RPI + sub;
      xr[k] = (i < I && j < J) ? X[i * ldx + j] : 0.0;
    }
  };
  if (X) {
    if (xpre) {
#pragma unroll
      for (int k = 0; k < NXL; ++k) xr[k] = xpre[k];
    } else {
      load_x(0);
    }
  }
  // One copy of the per-element code: the slab loop and the row loop are run-time loops (`unroll 1`); only the
  // register -> LDS write needs compile-time accumulator indices, so it sits in a wave-uniform switch.  (Fully unrolled,
  // the cost/Box-Muller code was inlined 64 times -- ~0.5 MB of instructions per kernel, every epilogue an I-cache miss
  // streak: 24 us per tile instead of 11.)
  auto write_slab = [&](auto ta_tag) {
    constexpr int ta = decltype(ta_tag)::value;
    if constexpr (ta < TI) {
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[(4 * r + q) * STRIDE + tb * 16 + c16] = acc.v[ta][tb][r];
    }
  };
#pragma unroll 1
  for (int ta = 0; ta < TI; ++ta) {
    switch (ta) {
      case 0: write_slab(std::integral_constant<int, 0>{}); break;
      case 1: write_slab(std::integral_constant<int, 1>{}); break;
      case 2: write_slab(std::integral_constant<int, 2>{}); break;
      case 3: write_slab(std::integral_constant<int, 3>{}); break;
      case 4: write_slab(std::integral_constant<int, 4>{}); break;
      case 5: write_slab(std::integral_constant<int, 5>{}); break;
      case 6: write_slab(std::integral_constant<int, 6>{}); break;
      default: write_slab(std::integral_constant<int, 7>{}); break;
    }
    // No workgroup barrier here or after the row loop: the slab is private to this wave and a wave's LDS operations
    // execute in issue order, so its reads see its own writes.  (A __syncthreads() would also wait vmcnt(0), i.e. for
    // the global stores of the previous slab to COMPLETE: with the write path loaded by the co-resident workgroup that
    // made the epilogue 165k cycles per tile instead of 14k -- measured with tools/stamp_probe.py.)
    if (X) {
#pragma unroll
      for (int k = 0; k < NXL; ++k) wx[(k * RPI + sub) * STRIDE + col] = xr[k];
      if (ta + 1 < TI) load_x(ta + 1);  // flies during this slab's row loop
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll UNROLL
    for (int it = 0; it < 8 / RPI; ++it) {
      const int p = it * RPI + sub;           // pair index 0..7 inside the 16-row slab
      const int rr = (p >> 2) * 8 + (p & 3);  // rows rr and rr + 4
      const int64_t i_lo = iw + ta * 16 + rr;
      const double v_lo = w[rr * STRIDE + col], v_hi = w[(rr + 4) * STRIDE + col];
      const int lr = ta * 16 + rr;
      RowConsts rc{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, it};
      if (X) {
        rc.x_lo = wx[rr * STRIDE + col];
        rc.x_hi = wx[(rr + 4) * STRIDE + col];
      }
      if constexpr (NCONST >= 1) {
        rc.k0_lo = __shfl(k0, lr);
        rc.k0_hi = __shfl(k0, lr + 4);
      }
      if constexpr (NCONST >= 2) {
        rc.k1_lo = __shfl(k1, lr);
        rc.k1_hi = __shfl(k1, lr + 4);
      }
      (void)lr;
      if (i_lo < I && j < J) fn(i_lo, j, v_lo, i_lo + 4 < I, v_hi, rc);
    }
    __builtin_amdgcn_wave_barrier();
  }
  static_assert(TI <= 8, "extend the pass list");
}

// lane l of a wave holds the per-row constant of row iw + l (0 beyond I)
__device__ __forceinline__ double load_row_constants(const double *vec, int64_t iw, int lane, int64_t I) {
  return (iw + lane < I) ? vec[iw + lane] : 0.0;
}

// ---- direct epilogue (interior tiles, cheap per-element work) ----------------------------------------------------
// The f64 MFMA and the vector ALU share an issue port: every VALU instruction of an epilogue waits behind a 64-cycle
// MFMA of the co-resident workgroup, and takes its own cycles from the matrix pipe.  So the epilogues whose per-element
// work is a few flops leave the registers directly in the MFMA layout: register r of block (ta, tb) holds rows
// 16 ta + 4 r + (lane >> 4), column 16 tb + (lane & 15) -- one buffer_store_dwordx2 writes four 128-byte row segments.
// Addressing costs no VALU: a buffer descriptor at the wave's corner, ONE lane offset register, and the row/column
// block as a scalar offset.  fn(v, slot, tb) returns the value to store; slot = 4 ta + r indexes the 16 row groups.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int TI, int TJ, class Fn>
__device__ __forceinline__ void epilogue_direct(const AccFrag<TI, TJ> &acc, double *wave_corner, int64_t ld, int lane,
                                                Fn &&fn) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(wave_corner, 0, 0x7FFFFFF0, 0x00020000);
  const int voff = (int)(((int64_t)(lane >> 4) * ld + (lane & 15)) * 8);
  const int ld4 = (int)(ld * 32);  // bytes per 4 rows
#pragma unroll
  for (int ta = 0; ta < TI; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb) {
        const double v = fn(acc.v[ta][tb][r], ta * 4 + r, tb, rs, voff, (ta * 4 + r) * ld4 + tb * 128);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v), rs, voff, (ta * 4 + r) * ld4 + tb * 128, 0);
      }
#else
  (void)acc, (void)wave_corner, (void)ld, (void)lane, (void)fn;
#endif
}

struct EpiStore {  // C = alpha * acc + beta * C   (split-K: slab `split` of C, slabs `slab` doubles apart)
  static constexpr int kTag = 1;  // PLS_TAG_GEMM_STORE
  static constexpr bool kDirect = true;
  double *C0;
  int64_t ldc;
  double alpha, beta;
  int64_t slab;
  __device__ int64_t direct_ld() const { return ldc; }
  template <int TI, int TJ>
  static constexpr bool direct_tile() { return true; }
  template <int TI, int TJ>
  __device__ void apply_direct(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int split, double *) const {
#if defined(__HIP_DEVICE_COMPILE__)
    double *corner = C0 + (int64_t)split * slab + iw * ldc + jw;
    if (beta == 0.0 && alpha == 1.0) {  // (the usual case: no VALU instruction at all)
      epilogue_direct<TI, TJ>(acc, corner, ldc, lane, [&](double v, int, int, __amdgpu_buffer_rsrc_t, int, int) { return v; });
    } else if (beta == 0.0) {
      epilogue_direct<TI, TJ>(acc, corner, ldc, lane,
                              [&](double v, int, int, __amdgpu_buffer_rsrc_t, int, int) { return alpha * v; });
    } else {
      epilogue_direct<TI, TJ>(acc, corner, ldc, lane, [&](double v, int, int, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
        const double c = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
        return alpha * v + beta * c;
      });
    }
#else
    (void)acc, (void)iw, (void)jw, (void)lane, (void)split;
#endif
  }
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int, int split, double *lds) const {
    double *C = C0 + (int64_t)split * slab;
    if (beta == 0.0) {
      epilogue_row_pairs<TI, TJ, 0>(acc, iw, jw, lane, wave, I, J, lds, 0.0, 0.0,
                                 [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &) {
                                   C[i * ldc + j] = alpha * v0;
                                   if (hi) C[(i + 4) * ldc + j] = alpha * v1;
                                 });
    } else {
      epilogue_row_pairs<TI, TJ, 0>(acc, iw, jw, lane, wave, I, J, lds, 0.0, 0.0,
                                 [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &) {
                                   C[i * ldc + j] = alpha * v0 + beta * C[i * ldc + j];
                                   if (hi) C[(i + 4) * ldc + j] = alpha * v1 + beta * C[(i + 4) * ldc + j];
                                 });
    }
  }
};

}  // namespace plship
