// libplship.so -- C ABI (include/plship.h) over the gfx950 kernels of the projected-Langevin-sampling hot path.
// Everything here enqueues work on the caller's stream; nothing allocates or synchronises.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define PLS_SCALAR_POLY_CONSTANTS 1  // (fmath.h: polynomial constants as scalar operands in this unit's kernels)
#include "../../include/plship.h"
#include "chol.h"
#include "common.h"
#include "cost_device.h"
#include "gemm_api.h"
#include "gemm_tn_f64.h"
#include "gemm_launch.h"
#include "gemm_tn_f64_kg.h"
#include "gemm_tn_f64_rows.h"
#include "cost_epilogues.h"
#include "philox.h"
#include "small_rank.h"
#include "small_rank_launch.h"
#include "small_rank_step.h"
#include "ipb_prep.h"
#include "small_rank_step_launch.h"
#include "step_params.h"

namespace plship {

#ifdef PLS_STAMP
unsigned long long *g_stamp_buffer = nullptr;  // diagnostic build only (tools/stamp_probe.py, tools/kg_stamp_probe.py)
extern "C" void pls_debug_set_stamp_buffer(unsigned long long *p) { g_stamp_buffer = p; }
// an empty launch of a given geometry: what a launch costs before its first instruction (tools/kg_stamp_probe.py)
__global__ void debug_empty_kernel(unsigned long long *stamps) {
  if (stamps && threadIdx.x == 0) {
    unsigned long long t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    stamps[blockIdx.x] = t_;
  }
}
// s_memtime ticks per tick of the constant-rate wall clock (hipDeviceAttributeWallClockRate kHz): out[0] = s_memtime ticks,
// out[1] = wall-clock ticks over a ~20 us spin of one wave
__global__ void debug_calibrate_kernel(unsigned long long *out) {
  unsigned long long t0, t1;
  const unsigned long long w0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  unsigned long long w1 = w0;
  while (w1 - w0 < 2000) w1 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = w1 - w0;
  }
}
extern "C" void pls_debug_calibrate(unsigned long long *out, void *stream) {
  hipLaunchKernelGGL(debug_calibrate_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out);
}
extern "C" void pls_debug_empty_launch(int grid, int block, int lds_bytes, unsigned long long *stamps, void *stream) {
  hipLaunchKernelGGL(debug_empty_kernel, dim3(grid), dim3(block), lds_bytes, reinterpret_cast<hipStream_t>(stream), stamps);
}
#endif

thread_local std::string g_last_error;
thread_local Timeline g_tl;

// A route option (pls_set_option): one value PER CALLING THREAD, starting from the default -- a launch takes the routes of
// the thread that makes the call, so two bases driven by two host threads can choose differently and nothing is process-wide
// ("no global state, thread-compatible per context").
struct RouteOption {
  int64_t v;
  int64_t load() const { return v; }
  void store(int64_t x) { v = x; }
};

// ---------------------------------------------------------------------------------------------------------------
// GEMM epilogues that need the cost functions / the noise generator
// ---------------------------------------------------------------------------------------------------------------

// Gaussian fast energy: acc = (B U)_ij;  partial[tile_i][j] = sum over the tile's rows of
//   pscale * u_ij * (acc_ij - 2 c_i) + 0.5 * u_ij^2 / lam_i       (cost quadratic form + prior energy of those rows)
template <int BI, int BJ, int WI, int WJ>
struct EpiGaussianQuad {
  static constexpr int kTag = PLS_TAG_GEMM_COST_VALUE;
  static constexpr bool kDirect = false;
  double *partial;
  int64_t ldp;
  const double *U;
  int64_t ldu;
  const double *c, *lam;
  double pscale;
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int tile_i, int, double *lds) const {
    double s = 0.0;
    const double cl = load_row_constants(c, iw, lane, I);
    const double hil = (lam && iw + lane < I) ? 0.5 / lam[iw + lane] : 0.0;  // 0.5 / lambda_i (lam == NULL: no prior term)
    epilogue_row_pairs<TI, TJ>(acc, iw, jw, lane, wave, I, J, lds, cl, hil,
                               [&](int64_t, int64_t, double v0, bool hi, double v1, const RowConsts &rc) {
                                 const double u0 = rc.x_lo;
                                 s += pscale * u0 * (v0 - 2.0 * rc.k0_lo) + u0 * u0 * rc.k1_lo;
                                 if (hi) {
                                   const double u1 = rc.x_hi;
                                   s += pscale * u1 * (v1 - 2.0 * rc.k0_hi) + u1 * u1 * rc.k1_hi;
                                 }
                               },
                               U, ldu);
    if (WJ == 32) s += __shfl_xor(s, 32);
    constexpr int NWJ = BJ / WJ, NWI = BI / WI;
    const int wrow = wave / NWJ, wcol = wave % NWJ;
    double *red = lds;  // overlaps the waves' slabs: wait until every wave has left its row loops
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

// Gaussian/identity fast path: acc = (B U)_ij;  out = [U +] -eta*(acc - c_i)/sigma2 - eta*U_ij/lam_i + sqrt(2 eta)*xi_ij
struct EpiLangevinGaussian {
  static constexpr int kTag = PLS_TAG_GEMM_LANGEVIN_GAUSSIAN;
  static constexpr bool kDirect = true;
  __device__ int64_t direct_ld() const { return ldo > ldu ? ldo : ldu; }
  template <int TI, int TJ>
  static constexpr bool direct_tile() { return TI == 4 && TJ == 4; }  // the 128 x 128 configuration (64 x 64 per wave)

  // Interior tiles of the big configuration: the whole Langevin update in the MFMA register layout, no LDS transpose.
  // Register r of block (ta, tb) is row iw + 16 ta + 4 r + (lane >> 4), column jw + 16 tb + (lane & 15):
  //   * registers r = 0, 1 (rows q, q + 4) and r = 2, 3 (rows q + 8, q + 12) are the two Philox pairs of a lane;
  //   * U is read and `out` written with one buffer instruction per register (four 128-byte row segments), addressed by
  //     a descriptor at the wave's corner, one lane-offset VGPR and a scalar row / column offset: no address VALU;
  //   * a lane owns FOUR columns (tb): their step sizes and Philox columns are hoisted out of the slab loop;
  //   * c_i and 1 / lambda_i of the wave's 64 rows go through a 1 KiB wave-private LDS table (lane l computes row
  //     iw + l once, with the same IEEE division as the LDS path), read back per (ta, r) by one ds_read_b64;
  //   * the slab loop is a run-time loop (the accumulators of slab ta are copied out by a wave-uniform switch), so the
  //     eight inlined Philox / Box-Muller bodies exist once, not four times.
  template <int TI, int TJ>
  __device__ void apply_direct(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int, double *wlds) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const int q = lane >> 4, c16 = lane & 15;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(out + iw * ldo + jw, 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(U + iw * ldu + jw), 0, 0x7FFFFFF0, 0x00020000);
    const int voff_o = (int)(((int64_t)q * ldo + c16) * 8), voff_u = (int)(((int64_t)q * ldu + c16) * 8);
    const int ldo4 = (int)(ldo * 32), ldu4 = (int)(ldu * 32);  // bytes per 4 rows
    // per-row constants of the wave's 64 rows -> wave-private LDS
    wlds[lane] = c[iw + lane];
    wlds[64 + lane] = lam ? 1.0 / lam[iw + lane] : 0.0;  // (lam == NULL: no prior term)
    // per-column constants of this lane's TJ columns
    double a2[TJ], s2[TJ], es[TJ];
    uint32_t jg[TJ];
    const uint64_t nstep = nz.live_step();
#pragma unroll
    for (int tb = 0; tb < TJ; ++tb) {
      const int64_t jl = jw + tb * 16 + c16;
      const double eta = etap.at(jl);
      a2[tb] = -eta;
      s2[tb] = sqrt(2.0 * eta);
      jg[tb] = (uint32_t)nz.global_column(jl);
      es[tb] = 0.0;
    }
    const double pscale = 0.5 * inv_noise;
    __builtin_amdgcn_wave_barrier();
    // run-time loop over the 2 TI row pairs of the wave block (slot = 2 ta + rp: rows 8 slot + q and + 4); the pair's
    // 2 TJ accumulators are copied out by a wave-uniform switch, so the TJ inlined noise bodies exist once
#pragma unroll 1
    for (int slot = 0; slot < 2 * TI; ++slot) {
      double v[2][TJ];
#define PLS_PAIR_CASE(S)                                            \
  case S:                                                           \
    _Pragma("unroll") for (int tb = 0; tb < TJ; ++tb) {             \
      v[0][tb] = acc.v[((S) / 2) % TI][tb][2 * ((S) % 2)];          \
      v[1][tb] = acc.v[((S) / 2) % TI][tb][2 * ((S) % 2) + 1];      \
    }                                                               \
    break;
      switch (slot) {
        PLS_PAIR_CASE(0)
        PLS_PAIR_CASE(1)
        PLS_PAIR_CASE(2)
        PLS_PAIR_CASE(3)
        PLS_PAIR_CASE(4)
        PLS_PAIR_CASE(5)
        PLS_PAIR_CASE(6)
        default:
          PLS_PAIR_CASE(7)
      }
#undef PLS_PAIR_CASE
      const int so_u = slot * 2 * ldu4, so_o = slot * 2 * ldo4;  // scalar byte offsets of row 8 slot (4 rows = ld4 bytes)
      double ur[2][TJ];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int tb = 0; tb < TJ; ++tb)
          ur[h][tb] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ru, voff_u, so_u + h * ldu4 + tb * 128, 0));
      const double cr0 = wlds[slot * 8 + q], cr1 = wlds[slot * 8 + 4 + q];
      const double il0 = wlds[64 + slot * 8 + q], il1 = wlds[64 + slot * 8 + 4 + q];
      const int64_t irow = iw + slot * 8 + q;  // bit 2 clear: the pair is rows irow, irow + 4
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb) {
        double z[2] = {0.0, 0.0};
        if (nz.kind == PLS_NOISE_PHILOX) {
          normal_pair(nz.seed, nstep, irow, (int64_t)jg[tb], z[0], z[1]);
        } else if (nz.kind == PLS_NOISE_INJECTED) {
          const int64_t jl = jw + tb * 16 + c16;
          z[0] = nz.xi[irow * nz.ldxi + jl];
          z[1] = nz.xi[(irow + 4) * nz.ldxi + jl];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double vv = v[h][tb], u = ur[h][tb], cr = h ? cr1 : cr0, il = h ? il1 : il0;
          const double d = fma(a2[tb], fma(inv_noise, vv - cr, u * il), s2[tb] * z[h]);
          const double o = add_u ? u + d : d;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, o), ro, voff_o, so_o + h * ldo4 + tb * 128, 0);
          if (epart) es[tb] += pscale * u * (vv - 2.0 * cr) + 0.5 * u * u * il;
        }
      }
    }
    if (epart) {  // per wave row (64 data rows): the four lane groups hold different rows of the same column
#pragma unroll
      for (int tb = 0; tb < TJ; ++tb) {
        double t = es[tb];
        t += __shfl_xor(t, 16);
        t += __shfl_xor(t, 32);
        if (lane < 16) store_partial(epart + (iw >> 6) * ldp + jw + tb * 16 + lane, t);
      }
      finish_energies(jw & ~(int64_t)127);  // (the 128 x 128 configuration: the tile starts at jw rounded down to 128)
    }
#else
    (void)acc, (void)iw, (void)jw, (void)lane, (void)wlds;
#endif
  }
  double *out;
  int64_t ldo;
  const double *U;
  int64_t ldu;
  const double *c, *lam;
  EtaP etap;
  double inv_noise;
  int add_u;
  NoiseP nz;
  // optional by-product: energy partials of the INPUT particles (acc = B U is exactly what their cost needs):
  // epart[tile_i][j] = sum over the tile's rows of 0.5*inv_noise * u (acc - 2 c_i) + 0.5 u^2 / lam_i
  double *epart;
  int64_t ldp;
  int nwj, bj;  // tile geometry of the launch (waves along j, tile width), set by the launcher
  // Optional (pls_block_desc.energy_sync): the energies are FINISHED by this launch -- no second kernel.  Every workgroup
  // leaves its partial rows write-through (sc1), drains them, and bumps the counter of its 256-column chunk (one agent-scope
  // atomic by one lane); the workgroup whose add comes last for a chunk -- it knows from the value the add returned -- reads
  // all partial rows of the chunk's columns (sc1 loads), adds them in ascending row order, writes the energies and the
  // chunk's sum in the library's fixed order, and puts the counter back to zero.  What the finishing kernel computed, bit for
  // bit, whoever comes last; nobody waits.  (Visibility rules: csrc/gemm_tn_f64_kg.h, kg_tri_publish.)
  struct Finish {
    unsigned *sync;     // cdiv(J, 256) counters, zero on entry and on exit; NULL: gaussian_energy_finish_kernel follows
    double *e;          // (J) energies
    double *sums;       // cdiv(J, 256) chunk sums, may be NULL / pinned host memory
    double yscale;
    const double *yty;
    int nparts, nti;    // partial rows in epart; tile rows of the launch (arrivals per column tile)
  } fin;

  // Optional (pls_block_desc.energy_partials_prev): this launch FINISHES the energies of the launch before it -- whose partial
  // rows a kernel boundary has made visible -- at its start: the workgroup of tile row 0 whose tile begins a 256-column chunk
  // adds the chunk's partial rows in ascending order, + the constant, and the chunk's sum in the library's fixed order: the
  // finishing kernel's values, bit for bit, with no tail behind any launch (the energies arrive one launch later).
  struct Prev {
    const double *part;  // previous launch's partial rows (nparts x J, leading dimension J); NULL: nothing to finish
    double *e;           // (J) their energies
    double *sums;        // cdiv(J, 256) chunk sums, may be NULL / pinned host memory
    double yscale;
    const double *yty;
    int nparts;
  } prev;
  int pregen_flag;  // 1: the k-split kernel draws the noise in front of its k-loop (PLS_OPT_KG_NOISE_PREGEN; launch-uniform)
  int zero_row = -1;  // >= 0: the 64-row tilings also write zeros into this partial row (lagged buffers: one layout)

  __device__ __forceinline__ bool prev_owner(int tile_i, int tile_j) const {
    return prev.part != nullptr && tile_i == 0 && (((int64_t)tile_j * bj) & 255) == 0;
  }
  __device__ __forceinline__ int prev_chunk_of(int tile_j) const { return (int)(((int64_t)tile_j * bj) >> 8); }
  // chunks of 256 columns the previous launch's energies fall into (0: nothing to finish): the k-split kernel appends that
  // many workgroups BEHIND its tiles, each finishing one chunk (prev_chunk) while the tiles contract
  __host__ __device__ int prev_chunks() const { return prev.part ? (int)((ldp + 255) >> 8) : 0; }
  __device__ __forceinline__ void prev_reduce(int chunk, double &v, double &tot) const {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ double pws[4];
    const int tid = threadIdx.x;
    const int64_t J = ldp, c0 = (int64_t)chunk << 8, col = c0 + tid;
    v = 0.0;
    if (tid < 256 && col < J) {
      double s = 0.0;
      for (int p0 = 0; p0 < prev.nparts; p0 += 16) {  // sixteen loads in flight, added in ascending row order
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = (p0 + k < prev.nparts) ? prev.part[(int64_t)(p0 + k) * J + col] : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (p0 + k < prev.nparts) s += t[k];
      }
      v = s + prev.yscale * (*prev.yty);
    }
    double r = v;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) r += __shfl_xor(r, o);
    __syncthreads();  // (pws may still be read by an earlier use)
    if (tid < 256 && (tid & 63) == 0) pws[tid >> 6] = r;
    __syncthreads();
    tot = (pws[0] + pws[1]) + (pws[2] + pws[3]);
#else
    (void)chunk, (void)v, (void)tot;
#endif
  }
  __device__ __forceinline__ void prev_store(int chunk, double v, double tot) const {
    const int tid = threadIdx.x;
    const int64_t J = ldp, c0 = (int64_t)chunk << 8, col = c0 + tid;
    if (tid < 256 && col < J) prev.e[col] = v;
    if (tid == 0 && prev.sums) prev.sums[chunk] = tot;
  }
  __device__ __forceinline__ void prev_chunk(int chunk) const {
    double v, tot;
    prev_reduce(chunk, v, tot);
    prev_store(chunk, v, tot);
  }

  __device__ __forceinline__ void store_partial(double *p, double v) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (fin.sync)
      __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another workgroup
    else
      *p = v;
#else
    (void)p, (void)v;
#endif
  }

  // called by every thread of the workgroup after its partial rows are stored; j_tile = first column of the tile
  // (the partial rows are J long: ldp == J)
  __device__ __forceinline__ void finish_energies(int64_t j_tile) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (!fin.sync) return;  // (launch-uniform)
    __shared__ unsigned arrived;
    __shared__ double ws[4];
    const int tid = threadIdx.x;
    const int64_t J = ldp;
    const int64_t chunk = j_tile >> 8, c0 = chunk << 8, c1 = (c0 + 256 < J) ? c0 + 256 : J;
    const unsigned expect = (unsigned)fin.nti * (unsigned)((c1 - c0 + bj - 1) / bj);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have been performed
    __syncthreads();
    if (tid == 0) arrived = __hip_atomic_fetch_add(fin.sync + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (arrived + 1 != expect) return;  // (workgroup-uniform)
    double v = 0.0;
    const int64_t col = c0 + tid;
    if (tid < 256 && col < J) {
      double s = 0.0;
      for (int p0 = 0; p0 < fin.nparts; p0 += 8) {  // eight loads in flight, added in ascending row order
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          t[k] = (p0 + k < fin.nparts)
                     ? __hip_atomic_load(epart + (int64_t)(p0 + k) * ldp + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                     : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (p0 + k < fin.nparts) s += t[k];
      }
      v = s + fin.yscale * (*fin.yty);
      fin.e[col] = v;
    }
    if (fin.sums) {  // chunk256_sum's order: xor butterfly inside each of the four waves, then (w0 + w1) + (w2 + w3)
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (tid < 256 && (tid & 63) == 0) ws[tid >> 6] = v;
      __syncthreads();
      if (tid == 0) fin.sums[chunk] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    }
    if (tid == 0) __hip_atomic_store(fin.sync + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    (void)j_tile;
#endif
  }
  // In front of the k-loop (gemm_tn_f64_kg.h, two k-groups): what the epilogue of a lane's 16 x 32 block needs and the
  // contraction does not produce.  The noise of its four row pairs, in the order the row loop of apply<1, 2> takes them
  // (iteration `it` handles pair p = 2 it + lane / 32: rows rr and rr + 4 of the block with rr = 8 (p >> 2) + (p & 3)) -- the
  // same normal_pair calls as in the epilogue, the same bits --, the particles of its eight elements in load_x's order, the
  // per-row constants of row iw + lane.
  __device__ __forceinline__ bool pregen_on() const { return pregen_flag != 0; }
  __device__ __forceinline__ void pregen(int64_t iw, int64_t jw, int lane, int64_t I, int64_t J, double (&pz)[8], double (&px)[8],
                                         double &pcl, double &pil) const {
    const int col = lane & 31, sub = lane >> 5;
    const int64_t jl = jw + col;
    pcl = load_row_constants(c, iw, lane, I);
    const double lm = (lam && iw + lane < I) ? lam[iw + lane] : 1.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = iw + k * 2 + sub;
      px[k] = (i < I && jl < J) ? U[i * ldu + jl] : 0.0;
    }
    const int64_t jc = jl < J ? jl : J - 1;
    if (nz.kind == PLS_NOISE_PHILOX) {
      const int64_t jg = nz.global_column(jc);
      const uint64_t nstep = nz.live_step();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = it * 2 + sub, rr = (p >> 2) * 8 + (p & 3);
        normal_pair(nz.seed, nstep, iw + rr, jg, pz[2 * it], pz[2 * it + 1]);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int p = it * 2 + sub, rr = (p >> 2) * 8 + (p & 3);
        const int64_t i = iw + rr;
        const bool inj = nz.kind == PLS_NOISE_INJECTED && jl < J;
        pz[2 * it] = (inj && i < I) ? nz.xi[i * nz.ldxi + jl] : 0.0;
        pz[2 * it + 1] = (inj && i + 4 < I) ? nz.xi[(i + 4) * nz.ldxi + jl] : 0.0;
      }
    }
    pil = (lam && iw + lane < I) ? 1.0 / lm : 0.0;
  }
  template <int TI, int TJ>
  __device__ __forceinline__ void apply_pregen(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I,
                                               int64_t J, int tile_i, int split, double *lds, const double (&pz)[8],
                                               const double (&px)[8], double pcl, double pil) const {
    static_assert(TI == 1 && TJ == 2, "the pre-drawn pairs follow the row loop of a 16 x 32 block");
    apply_impl<TI, TJ, true>(acc, iw, jw, lane, wave, I, J, tile_i, split, lds, pz, px, pcl, pil);
  }
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J,
                        int tile_i, int split, double *lds) const {
    const double none[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    apply_impl<TI, TJ, false>(acc, iw, jw, lane, wave, I, J, tile_i, split, lds, none, none, 0.0, 0.0);
  }
  template <int TI, int TJ, bool PRE>
  __device__ __forceinline__ void apply_impl(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I,
                                             int64_t J, int tile_i, int, double *lds, const double (&pz)[8],
                                             const double (&px)[8], double pcl, double pil) const {
    double es = 0.0;
    const double pscale = 0.5 * inv_noise;
    const double cl = PRE ? pcl : load_row_constants(c, iw, lane, I);
    const double ilaml = PRE ? pil : ((lam && iw + lane < I) ? 1.0 / lam[iw + lane] : 0.0);
    const uint64_t nstep = nz.live_step();
    // a lane owns ONE column for the whole epilogue: its step size and Philox column are loop invariants
    const int64_t jl = jw + lane % (TJ * 16);
    const int64_t jc = jl < J ? jl : J - 1;
    const double eta = etap.at(jc), sq2eta = sqrt(2.0 * eta), neta = -eta;
    const int64_t jg = nz.global_column(jc);
    epilogue_row_pairs<TI, TJ, 2, PRE ? 4 : 1>(  // (PRE: the row loop unrolled, so that rc.it is a constant)
        acc, iw, jw, lane, wave, I, J, lds, cl, ilaml,
        [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &rc) {
          double z0 = 0.0, z1 = 0.0;
          if constexpr (PRE) {
            z0 = pz[2 * rc.it];
            z1 = pz[2 * rc.it + 1];
          } else if (nz.kind == PLS_NOISE_PHILOX) {
            normal_pair(nz.seed, nstep, i, jg, z0, z1);  // rows i and i + 4 share one Philox call
          } else if (nz.kind == PLS_NOISE_INJECTED) {
            z0 = nz.xi[i * nz.ldxi + j];
            if (hi) z1 = nz.xi[(i + 4) * nz.ldxi + j];
          }
          // (explicit fma's: both instantiations -- noise drawn here or in front of the k-loop -- must round alike, and left
          // to itself the compiler contracts a sum of products one way or another depending on the code around it)
          {
            const double u = rc.x_lo, ci = rc.k0_lo, il = rc.k1_lo;
            const double d = fma(neta, fma(inv_noise, v0 - ci, u * il), sq2eta * z0);
            out[i * ldo + j] = add_u ? u + d : d;
            es += fma(pscale * u, fma(-2.0, ci, v0), (0.5 * u) * u * il);
          }
          if (hi) {
            const double u = rc.x_hi, ci = rc.k0_hi, il = rc.k1_hi;
            const double d = fma(neta, fma(inv_noise, v1 - ci, u * il), sq2eta * z1);
            out[(i + 4) * ldo + j] = add_u ? u + d : d;
            es += fma(pscale * u, fma(-2.0, ci, v1), (0.5 * u) * u * il);
          }
        },
        U, ldu, PRE ? px : nullptr);
    if (epart) {  // wave-uniform; fixed-order cross-wave sum like EpiCostValue
      constexpr int WJ = TJ * 16;
      if (WJ == 32) es += __shfl_xor(es, 32);
      const int wrow = wave / nwj, wcol = wave % nwj;
      const int nwi = (int)(blockDim.x >> 6) / nwj;
      double *red = lds;  // overlaps the waves' slabs: wait until every wave has left its row loops
      __syncthreads();
      if (lane < WJ) red[wrow * bj + wcol * WJ + lane] = es;
      __syncthreads();
      const int t = threadIdx.x;
      if (t < bj) {
        double tot = 0.0;
        for (int w = 0; w < nwi; ++w) tot += red[w * bj + t];
        const int64_t j = (jw - wcol * WJ) + t;
        if (j < J) {
          if (bj == 128) {  // big configuration: partial rows are per 64 data rows (the direct path writes one per wave row)
            store_partial(epart + (int64_t)(2 * tile_i) * ldp + j, tot);
            store_partial(epart + (int64_t)(2 * tile_i + 1) * ldp + j, 0.0);
          } else {
            store_partial(epart + (int64_t)tile_i * ldp + j, tot);
            // (a lagged buffer holds 2 cdiv(mk, 128) rows whatever tiling wrote it: the 64-row tiles leave one row fewer when
            // mk mod 128 is in 1 .. 64, and the workgroups of tile row 0 zero it -- no memset node in front of the launch)
            if (zero_row >= 0 && tile_i == 0) store_partial(epart + (int64_t)zero_row * ldp + j, 0.0);
          }
        }
      }
      finish_energies(jw - wcol * WJ);
    }
  }
};


// Last product of the whitened inducing-point step: acc = Lc dS (the update mapped back from whitened coordinates),
//   out = [U +] acc + sqrt(2 eta_col) * e      (e: injected, already coloured noise of a parity run; NULL otherwise)
// (tagged as a plain store: with a triangular operand the launcher pairs tile rows t and nti - 1 - t like EpiStore's)
struct EpiIpbFinish {
  static constexpr int kTag = PLS_TAG_GEMM_STORE;
  static constexpr bool kDirect = false;
  double *out;
  int64_t ldo;
  const double *U;
  int64_t ldu;
  int add_u;
  EtaP etap;
  const double *xi;
  int64_t ldxi;
  template <int TI, int TJ>
  __device__ void apply(const AccFrag<TI, TJ> &acc, int64_t iw, int64_t jw, int lane, int wave, int64_t I, int64_t J, int, int,
                        double *lds) const {
    const int64_t jl = jw + lane % (TJ * 16);
    const int64_t jc = jl < J ? jl : J - 1;
    const double s2 = xi ? sqrt(2.0 * etap.at(jc)) : 0.0;
    epilogue_row_pairs<TI, TJ, 0>(
        acc, iw, jw, lane, wave, I, J, lds, 0.0, 0.0,
        [&](int64_t i, int64_t j, double v0, bool hi, double v1, const RowConsts &rc) {
          double o0 = add_u ? rc.x_lo + v0 : v0, o1 = add_u ? rc.x_hi + v1 : v1;
          if (xi) {
            o0 = fma(s2, xi[i * ldxi + j], o0);
            if (hi) o1 = fma(s2, xi[(i + 4) * ldxi + j], o1);
          }
          out[i * ldo + j] = o0;
          if (hi) out[(i + 4) * ldo + j] = o1;
        },
        add_u ? U : nullptr, ldu);
  }
};

// Split-K plan for the back-projection D (I x J) = L^T R with a long contraction (K = rows of the N chunk).  Two reasons
// to cut the contraction into slabs (summed in a fixed order by the update kernel: deterministic, no atomics):
//   occupancy -- too few 128x128 output tiles to put two workgroups on each of the 256 CUs (narrow particle shards);
//   locality  -- over a very long k-loop the co-resident workgroups drift apart and stop sharing operand panels in
//                their XCD's L2: at K = 1e5 one slab reads 66 GB through the fabric, 8 slabs 20 GB, at equal speed
//                (DESIGN.md "tuning log"); slabs of <= 16384 rows keep the drift inside the L2 window.
// Returns the number of slabs (<= 16).
static int64_t plan_split_k(int64_t I, int64_t J, int64_t K, int64_t *kchunk) {
  const int64_t tiles = cdiv(I, 128) * cdiv(J, 128);
  int64_t s = 1;
  if (tiles < 512) s = cdiv(512, tiles);
  const int64_t s_local = cdiv(K, 16384);
  if (s_local > s) s = s_local;
  if (s > 16) s = 16;
  // wave quantisation: tiles * s workgroups run in rounds of 512 (2 per CU); a few more slabs can fill the last round
  // (J = 2048: 128 tiles x 7 slabs = 1.75 rounds -> 87 % of the MFMA rate; x 8 = 2 rounds)
  if (s > 1) {
    auto waste = [&](int64_t sl) {
      const double rounds = (double)(tiles * sl) / 512.0;
      return std::ceil(rounds) / rounds;
    };
    int64_t best = s;
    for (int64_t sl = s + 1; sl <= 16 && sl <= s + 4; ++sl)
      if (waste(sl) < waste(best) - 0.03) best = sl;
    s = best;
  }
  // keep every slab's k-loop long enough to amortise its prologue / epilogue; a handful of tiles (small ranks AND few
  // particles) is latency-bound on its serial k-loop instead, so shorter slabs pay (M_k = 129, J = 256, N = 2000:
  // 12 workgroups walked 125 k-steps each)
  const int64_t min_chunk = tiles < 64 ? 256 : 1024;
  while (s > 1 && K / s < min_chunk) --s;
  int64_t kc = cdiv(cdiv(K, s), 16) * 16;
  s = cdiv(K, kc);
  *kchunk = (s > 1) ? kc : 0;
  return s;
}

// ---- few output tiles: 64 x 64 tiles with the k range split over wave groups inside the workgroup (gemm_tn_f64_kg.h) ----
static thread_local RouteOption g_ksplit_mode{1};          // pls_set_option(PLS_OPT_KSPLIT_MODE): 0 off, 1 auto, 2 / 3 force 2 / 1 k-groups
static thread_local RouteOption g_ksplit_max_tiles{256};   // pls_set_option(PLS_OPT_KSPLIT_MAX_TILES): 128 x 128 tiles below which it is taken

enum GemmCfg { CFG_BIG = 0, CFG_SMALL = 1, CFG_KG1 = 2, CFG_KG2 = 3 };

// Which configuration a contraction of this shape runs in (the energy partial layout of the fast path follows from it)
static GemmCfg pick_gemm_cfg(const double *L, int64_t ldl, const double *R, int64_t ldr, int64_t I, int64_t J, int64_t K,
                             int64_t nsplit = 1) {
  const int64_t mode = g_ksplit_mode.load();
  const int64_t tiles128 = cdiv(I, 128) * cdiv(J, 128) * nsplit;
  const bool aligned = ((ldl | ldr) & 1) == 0 && ((reinterpret_cast<uintptr_t>(L) | reinterpret_cast<uintptr_t>(R)) & 15) == 0 &&
                       ldl < ((int64_t)1 << 22) && ldr < ((int64_t)1 << 22);
  if (mode != 0 && aligned && nsplit == 1 && K >= 1) {
    if (mode == 2) return CFG_KG2;
    if (mode == 3) return CFG_KG1;
    if (tiles128 < g_ksplit_max_tiles.load() && K >= 32) return CFG_KG2;
  }
  return tiles128 >= 256 ? CFG_BIG : CFG_SMALL;
}

template <int KG, class Epi>
static int launch_gemm_kg(GemmShape g, const Epi &epi, hipStream_t st) {
  using G = KgGeom<KG>;
  constexpr size_t lds_bytes = (size_t)G::LDS_DOUBLES * sizeof(double);
  auto kern = gemm_tn_f64_kg_kernel<KG, Epi>;
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_ready)) return rc;
  g.nti = (int)cdiv(g.I, 64);
  g.ntj = (int)cdiv(g.J, 64);
#ifdef PLS_STAMP
  g.stamps = g_stamp_buffer;
#endif
  const int64_t nwg = (int64_t)g.nti * g.ntj;
  if (nwg <= 0) return PLS_OK;
  if (nwg > 0x7fffffff) return fail(PLS_ERR_INVALID_ARGUMENT, "gemm: too many tiles");
  int64_t extra = 0;  // workgroups behind the tiles that finish the previous launch's energies (gemm_tn_f64_kg_kernel)
  if constexpr (epi_has_prev<Epi>::value) extra = epi.prev_chunks();
  {
    LaunchScope scope(Epi::kTag, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nwg + extra), 1), dim3(G::NT), lds_bytes, st, g, epi);
  }
  return check_launch("gemm_tn_f64_kg");
}

// Balanced triangular product (gemm_tn_f64_kg_tri_kernel): tile rows paired, every pair shared by two workgroups with equal
// loads, the heavy tile finished by whichever arrives second.  scratch: the caller's, zero flag words on entry and on exit.
static thread_local RouteOption g_tri_balance{1};  // pls_set_option(PLS_OPT_TRI_BALANCE): 0 off (one tile per workgroup), 1 on

struct TriScratch {
  void *ptr = nullptr;
  size_t bytes = 0;
};

static bool tri_balanced_ok(int64_t I, int64_t J, int64_t K, int tri, const TriScratch &sc) {
  if (!tri || g_tri_balance.load() == 0 || !sc.ptr || (reinterpret_cast<uintptr_t>(sc.ptr) & 15) != 0) return false;
  const int64_t pairs = (cdiv(I, 64) + 1) / 2, ntj = cdiv(J, 64);
  // two or more tile rows (a single row has nothing to balance), flags within their 16 KB, the slots within the scratch
  return cdiv(I, 64) >= 2 && K >= 64 && pairs * ntj <= kKgTriFlagBytes / 4 && sc.bytes >= kg_tri_scratch_bytes(I, J);
}

template <class Epi>
static int launch_gemm_kg_tri(GemmShape g, const Epi &epi, const TriScratch &sc, hipStream_t st) {
  using G = KgGeom<2>;
  constexpr size_t lds_bytes = (size_t)G::LDS_DOUBLES * sizeof(double);
  auto kern = gemm_tn_f64_kg_tri_kernel<Epi>;
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_ready)) return rc;
  g.nti = (int)cdiv(g.I, 64);
  g.ntj = (int)cdiv(g.J, 64);
  g.tri_flags = static_cast<unsigned *>(sc.ptr);
  g.tri_part = reinterpret_cast<double *>(static_cast<char *>(sc.ptr) + kKgTriFlagBytes);
  const int64_t nwg = (int64_t)2 * ((g.nti + 1) / 2) * g.ntj;
  {
    LaunchScope scope(Epi::kTag, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, 1), dim3(G::NT), lds_bytes, st, g, epi);
  }
  return check_launch("gemm_tn_f64_kg_tri");
}

// launch_gemm of gemm_launch.h plus the k-split configurations (this translation unit's epilogues only)
template <class Epi>
static int launch_gemm_any(const double *L, int64_t ldl, const double *R, int64_t ldr, int64_t I, int64_t J, int64_t K,
                           const Epi &epi, hipStream_t st, int64_t kchunk = 0, int tri = 0, TriScratch scratch = TriScratch{}) {
  const int64_t nsplit = (kchunk > 0 && kchunk < K) ? cdiv(K, kchunk) : 1;
  const GemmCfg cfg = pick_gemm_cfg(L, ldl, R, ldr, I, J, K, nsplit);
  if (cfg == CFG_KG2 || cfg == CFG_KG1) {
    GemmShape g{L, ldl, R, ldr, I, J, K, 0, 0, 0, tri};
    if constexpr (Epi::kTag == PLS_TAG_GEMM_STORE) {  // (only the storing epilogues are ever launched with a triangular operand)
      if (cfg == CFG_KG2 && tri_balanced_ok(I, J, K, tri, scratch)) return launch_gemm_kg_tri(g, epi, scratch, st);
    }
    return cfg == CFG_KG2 ? launch_gemm_kg<2>(g, epi, st) : launch_gemm_kg<1>(g, epi, st);
  }
  return launch_gemm(L, ldl, R, ldr, I, J, K, epi, st, kchunk, tri);
}

// ---- a row count that is not a multiple of 128: equal-height tiles, 16-row blocks dealt to the wave rows (gemm_tn_f64_rows.h) ----
static thread_local RouteOption g_row_blocks_mode{1};  // pls_set_option(PLS_OPT_ROW_BLOCKS): 0 off (the round-2 pieces), 1 on

static bool gemm_rows_ok(const double *L, int64_t ldl, const double *R, int64_t ldr, int64_t I, int64_t J, int64_t K, int64_t ldc,
                         int64_t nsplit) {
  const bool aligned = ((ldl | ldr) & 1) == 0 && ((reinterpret_cast<uintptr_t>(L) | reinterpret_cast<uintptr_t>(R)) & 15) == 0;
  if (!(g_row_blocks_mode.load() != 0 && I > 128 && I % 128 != 0 && J >= 1 && K >= 1 && aligned && ldc >= 128 &&
        ldc < kDirectMaxLd && use_big_tiles(I, J, nsplit)))
    return false;
  // every tile contracts cdiv(blocks, tiles) blocks: above two tiles that may pad more than the remainder launches cost
  // (400 rows = 25 blocks run as 4 x 7: 9.8 ms against 9.5 ms for 384 + 16 rows, N = 1e5, J = 8192)
  const int64_t blocks = cdiv(I, 16), tiles = cdiv(I, 128);
  return tiles == 2 || tiles * cdiv(blocks, tiles) - blocks <= 1;
}

static int launch_gemm_rows(const double *L, int64_t ldl, const double *R, int64_t ldr, int64_t I, int64_t J, int64_t K,
                            const EpiStore &e, hipStream_t st, int64_t kchunk) {
  GemmShape g{L, ldl, R, ldr, I, J, K, 0, 0, kchunk, 0};
  constexpr size_t lds_bytes = (size_t)2 * 16 * ((128 + 16) + (128 + 16)) * sizeof(double);
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(gemm_tn_f64_rows_kernel), lds_bytes, lds_ready)) return rc;
  g.nti = (int)cdiv(I, 128);
  g.ntj = (int)cdiv(J, 128);
  const int tile_rows = 16 * (int)cdiv(cdiv(I, 16), g.nti);  // equal heights: 160 rows are 80 + 80, 129 are 80 + 49
  const int64_t nwg = (int64_t)g.nti * g.ntj;
  if (nwg > 0x7fffffff) return fail(PLS_ERR_INVALID_ARGUMENT, "gemm: too many tiles");
  unsigned nsplit = 1;
  if (kchunk > 0 && kchunk < K) nsplit = (unsigned)cdiv(K, kchunk);
  RowsStore rs{e.C0, e.ldc, e.alpha, e.beta, e.slab};
  {
    LaunchScope scope(EpiStore::kTag, st);
    hipLaunchKernelGGL(gemm_tn_f64_rows_kernel, dim3((unsigned)nwg, nsplit), dim3(256), lds_bytes, st, g, tile_rows, rs);
  }
  return check_launch("gemm_tn_f64_rows");
}

// cost-value GEMM (tile geometry is part of the epilogue type); returns the number of partial rows written
static int64_t cost_value_partial_rows(int64_t I, int64_t J) { return use_big_tiles(I, J) ? cdiv(I, 128) : cdiv(I, 64); }


// ---------------------------------------------------------------------------------------------------------------
// HBM-bound kernels
// ---------------------------------------------------------------------------------------------------------------

// exp(x) for x <= 0 (the RBF exponent): n = rint(x log2 e), r = x - n ln 2 (two-piece ln 2), degree-13 Taylor polynomial
// in Horner form (|r| <= 0.347: truncation 4e-18 relative), scaled by 2^n with v_ldexp (gradual underflow as libm).
// Within 1 ulp of the correctly rounded value; about half the instructions of the library exp (no special-case
// ladder, constants in scalar registers).
__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = fma_k(p, r, 2.0876756987868098e-09);
  p = fma_k(p, r, 2.5052108385441720e-08);
  p = fma_k(p, r, 2.7557319223985893e-07);
  p = fma_k(p, r, 2.7557319223985888e-06);
  p = fma_k(p, r, 2.4801587301587302e-05);
  p = fma_k(p, r, 1.9841269841269841e-04);
  p = fma_k(p, r, 1.3888888888888889e-03);
  p = fma_k(p, r, 8.3333333333333332e-03);
  p = fma_k(p, r, 4.1666666666666664e-02);
  p = fma_k(p, r, 1.6666666666666666e-01);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const double v = ldexp(p, (int)n);
  return (x < -745.2) ? 0.0 : v;
}

// k(x1, x2): a block covers GRAM_ROWS rows x 512 columns; a thread owns one PAIR of columns (its two x2 points stay in
// registers, pre-scaled by 1/lengthscale) and walks down the rows, whose pre-scaled x1 points sit in LDS (broadcast
// reads).  Writes are 16 B per lane, 4 KiB contiguous per row and block: the kernel is bound by the 8*n1*n2 bytes it
// writes once the per-element work is ~35 fp64 instructions (D = 8).
constexpr int GRAM_ROWS = 64;
// D_MAX: the input dimension rounded up to a power of two; the padding coordinates are zero on both sides, so the
// unrolled coordinate loops need no `k < d` test.
template <int KIND, int D_MAX>
__global__ __launch_bounds__(256) void kernel_gram_kernel(const double *__restrict__ x1, int64_t n1,
                                                           const double *__restrict__ x2, int64_t n2, int d,
                                                           const double *__restrict__ lengthscale, double outputscale,
                                                           double *__restrict__ out, int64_t ldout) {
  __shared__ double inv_ls[D_MAX];
  __shared__ __attribute__((aligned(16))) double a_s[GRAM_ROWS][D_MAX];  // rows of x1 (pre-scaled, zero-padded)
  const int t = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.y * GRAM_ROWS;
  const int nrows = (int)((n1 - row0 < GRAM_ROWS) ? (n1 - row0) : GRAM_ROWS);
  if (t < D_MAX) inv_ls[t] = (t < d) ? ((KIND == PLS_KERNEL_RBF_ARD) ? 1.0 / lengthscale[t] : 1.0) : 0.0;
  __syncthreads();
  for (int e = t; e < nrows * D_MAX; e += 256) {
    const int r = e / D_MAX, k = e % D_MAX;
    a_s[r][k] = (k < d) ? x1[(row0 + r) * d + k] * inv_ls[k] : 0.0;
  }
  __syncthreads();
  const int64_t col = ((int64_t)blockIdx.x * 256 + t) * 2;
  if (col >= n2) return;
  const bool two = col + 1 < n2;
  double b0[D_MAX], b1[D_MAX];
#pragma unroll
  for (int k = 0; k < D_MAX; ++k) {
    b0[k] = (k < d) ? x2[col * d + k] * inv_ls[k] : 0.0;
    b1[k] = (k < d && two) ? x2[(col + 1) * d + k] * inv_ls[k] : 0.0;
  }
  const bool vec = two && ((ldout & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  double *dst = out + row0 * ldout + col;
#pragma unroll 2
  for (int r = 0; r < nrows; ++r, dst += ldout) {
    double s0 = 0.0, s1 = 0.0;
    if (KIND == PLS_KERNEL_RBF_ARD) {
#pragma unroll
      for (int k = 0; k < D_MAX; ++k) {
        const double a = a_s[r][k];
        const double e0 = a - b0[k], e1 = a - b1[k];
        s0 = fma(e0, e0, s0);
        s1 = fma(e1, e1, s1);
      }
      s0 = outputscale * exp_nonpos(-0.5 * s0);
      s1 = outputscale * exp_nonpos(-0.5 * s1);
    } else {
#pragma unroll
      for (int k = 0; k < D_MAX; ++k) {
        const double a = a_s[r][k];
        s0 = fma(a, b0[k], s0);
        s1 = fma(a, b1[k], s1);
      }
    }
    if (vec) {
      *reinterpret_cast<double2_t *>(dst) = double2_t{s0, s1};
    } else {
      dst[0] = s0;
      if (two) dst[1] = s1;
    }
  }
}

template <int KIND>
static void launch_gram(dim3 grid, hipStream_t st, const double *x1, int64_t rows, const double *x2, int64_t n2, int d,
                        const double *lengthscale, double outputscale, double *out, int64_t ldout) {
#define PLS_GRAM_CASE(DM)                                                                                         \
  hipLaunchKernelGGL((kernel_gram_kernel<KIND, DM>), grid, dim3(256), 0, st, x1, rows, x2, n2, d, lengthscale, \
                     outputscale, out, ldout)
  if (d <= 1) PLS_GRAM_CASE(1);
  else if (d <= 2) PLS_GRAM_CASE(2);
  else if (d <= 4) PLS_GRAM_CASE(4);
  else if (d <= 8) PLS_GRAM_CASE(8);
  else if (d <= 16) PLS_GRAM_CASE(16);
  else if (d <= 32) PLS_GRAM_CASE(32);
  else PLS_GRAM_CASE(64);
#undef PLS_GRAM_CASE
}

__global__ __launch_bounds__(256) void debug_math_kernel(int op, const double *__restrict__ x, double *__restrict__ out,
                                                         int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (op == 0) out[i] = fast_exp(x[i]);
  else if (op == 1) out[i] = fast_log(x[i]);
  else if (op == 2) out[i] = fast_div(x[i], x[n + i]);         // numerators x[0 .. n), denominators x[n .. 2 n)
  else out[i] = fast_div_normal(x[i], x[n + i]);
}

// elementwise cost derivative (un-fused entry point)
__global__ __launch_bounds__(256) void cost_deriv_kernel(CostP cp, const double *__restrict__ F, int64_t ldf,
                                                          const double *__restrict__ y, int64_t n, int64_t j,
                                                          double *__restrict__ G, int64_t ldg) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= j) return;
  for (int64_t row = blockIdx.y; row < n; row += gridDim.y) G[row * ldg + col] = cost_deriv(cp, y[row], F[row * ldf + col]);
}

// stage 1 of the column sums: partial[rb][col] = sum over rows [rb*RB, rb*RB + RB) of cost(y, F)
constexpr int COST_RB = 256;
__global__ __launch_bounds__(256) void cost_value_partial_kernel(CostP cp, const double *__restrict__ F, int64_t ldf,
                                                                  const double *__restrict__ y, int64_t n, int64_t j,
                                                                  double *__restrict__ partial, int64_t ldp) {
  __shared__ double red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * COST_RB;
  double s = 0.0;
  if (col < j) {
    for (int rr = ty; rr < COST_RB; rr += 4) {
      const int64_t row = r0 + rr;
      if (row < n) s += cost_value(cp, y[row], F[row * ldf + col]);
    }
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < j) partial[(int64_t)blockIdx.y * ldp + col] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// stage 2: out[col] = (accumulate ? out[col] : 0) + sum_rb partial[rb][col]  (+ prior energy term)
//   prior_kind 0: none; 1: + 0.5 * sum_m P[m][col]^2 * scale_vec[m]  (ONB: P = U, scale_vec = 1/lam  -> pass lam, inverted here)
//              2: + scale * sum_m P[m][col]^2                       (IPB: P = W U, scale = M/2)
//   the partial sum enters as pscale * (sum + *padd) (Gaussian fast energy: pscale = 1/(2 sigma2), padd -> y^T y)
__global__ __launch_bounds__(256) void column_reduce_kernel(const double *__restrict__ partial, int64_t ldp,
                                                             int64_t nparts, int64_t j, double *__restrict__ out,
                                                             int accumulate, int prior_kind,
                                                             const double *__restrict__ P, int64_t ldpp, int64_t m,
                                                             const double *__restrict__ lam, double scale,
                                                             double pscale, const double *__restrict__ padd) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= j) return;
  // (Loads in batches of eight, sums in the order they always had: with a few hundred particles this kernel is two workgroups
  // whose threads walk nparts + m rows one dependent load at a time -- 40 us for 32 + 128 rows, the longest launch of a
  // training iteration at the reference's own problem sizes -- while the bits of the energies must not move.)
  double s = 0.0;
  for (int64_t p0 = 0; p0 < nparts; p0 += 8) {
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (p0 + k < nparts) ? partial[(p0 + k) * ldp + col] : 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (p0 + k < nparts) s += t[k];
  }
  s = pscale * (s + (padd ? *padd : 0.0)) + (accumulate ? out[col] : 0.0);
  if (prior_kind == 1) {
    double e = 0.0;
    for (int64_t r0 = 0; r0 < m; r0 += 8) {
      double u[8], l[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool in = r0 + k < m;
        u[k] = in ? P[(r0 + k) * ldpp + col] : 0.0;
        l[k] = in ? lam[r0 + k] : 1.0;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r0 + k < m) e += u[k] * u[k] / l[k];
    }
    s += 0.5 * e;
  } else if (prior_kind == 2) {
    double e = 0.0;
    for (int64_t r0 = 0; r0 < m; r0 += 8) {
      double u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] = (r0 + k < m) ? P[(r0 + k) * ldpp + col] : 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (r0 + k < m) e += u[k] * u[k];
    }
    s += scale * e;
  }
  out[col] = s;
}

// out = [U +] -eta * D - eta * pscale_i * P + sq2eta * noise     (rows x J)
//   ONB: P = U, pscale_i = 1/lam_i (pass lam, lam_is_vec = 1);  IPB: P = W U, pscale = M (lam = NULL, pconst = M)
// One thread per (row pair {ib, ib+4}, column): the pair shares one Philox call (philox.h).
// (out may alias slab 0 of D element for element: each thread reads D[..][i][col] before it writes out[i][col].)
// D may be split into nslab slabs (split-K partial sums of the back-projection), slab_stride doubles apart.
__global__ __launch_bounds__(256) void langevin_update_kernel(double *out, int64_t ldo,
                                                               const double *__restrict__ U, int64_t ldu,
                                                               const double *D, int64_t ldd, int nslab,
                                                               int64_t slab_stride,
                                                               const double *__restrict__ P, int64_t ldp,
                                                               const double *__restrict__ lam, double pconst,
                                                               int64_t rows, int64_t j, EtaP etap,
                                                               int add_u, NoiseP nz,
                                                               const double *__restrict__ dsub = nullptr,
                                                               double dsub_scale = 0.0) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= j) return;
  const double eta = etap.at(col), sq2eta = sqrt(2.0 * eta);
  const int64_t jg = nz.global_column(col);
  const int64_t npairs = cdiv(rows, 8) * 4;
  for (int64_t pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
    const int64_t ib = (pr >> 2) * 8 + (pr & 3);
    if (ib >= rows) continue;
    double z0 = 0.0, z1 = 0.0;
    if (nz.kind == PLS_NOISE_PHILOX) {
      normal_pair(nz.seed, nz.live_step(), ib, jg, z0, z1);
    } else if (nz.kind == PLS_NOISE_INJECTED) {
      z0 = nz.xi[ib * nz.ldxi + col];
      if (ib + 4 < rows) z1 = nz.xi[(ib + 4) * nz.ldxi + col];
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t i = ib + 4 * h;
      if (i < rows) {
        const double ps = lam ? 1.0 / lam[i] : pconst;
        double drift = D[i * ldd + col];
        for (int s0 = 1; s0 < nslab; s0 += 8) {  // split-K slabs, fixed order; eight loads in flight
          double t[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] = (s0 + k < nslab) ? D[(int64_t)(s0 + k) * slab_stride + i * ldd + col] : 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (s0 + k < nslab) drift += t[k];
        }
        if (dsub) drift -= dsub_scale * dsub[i];  // (IPB Gaussian fast path: drift = (B V - c) / sigma2)
        const double d = -eta * drift - eta * ps * P[i * ldp + col] + sq2eta * (h ? z1 : z0);
        out[i * ldo + col] = add_u ? U[i * ldu + col] + d : d;
      }
    }
  }
}

// IPB Gaussian fast path: e[col] = sum_i v (0.5 d - c_i / sigma2) + (M/2) v^2  + y^T y / (2 sigma2),
//   v = V[i][col] = (K^-1 U)_i,  d = D[i][col] = (Kzx Kxz V)_i / sigma2  (the quadratic form of the Gaussian cost in V).
// 64 columns x 4 row slices per block, slices summed through LDS in a fixed order.
__global__ __launch_bounds__(256) void ipb_gaussian_energy_kernel(const double *__restrict__ V, int64_t ldv,
                                                                   const double *__restrict__ D, int64_t ldd,
                                                                   const double *__restrict__ c, int64_t m, int64_t j,
                                                                   double inv_noise, double mhalf, double *__restrict__ e) {
  __shared__ double part[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + cl;
  double s = 0.0;
  if (col < j)
    for (int64_t i = sl; i < m; i += 4) {
      const double v = V[i * ldv + col];
      s += v * (0.5 * D[i * ldd + col] - inv_noise * c[i]) + mhalf * v * v;
    }
  part[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && col < j) e[col] = ((part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl])) + 0.5 * inv_noise * c[m];
}

// Sum of one value per thread over a 256-thread workgroup in a FIXED order: xor butterfly inside each wave (offsets 32, 16,
// 8, 4, 2, 1), then (w0 + w1) + (w2 + w3).  Every mean of per-particle energies in the library is built from these chunk
// sums added in ascending chunk order (block_means_kernel, the fused finish kernel below, the host side of the training
// loops), so all loop variants report bit-identical energies.  Returns the sum in every thread; `ws`: 4 doubles of LDS.
__device__ __forceinline__ double chunk256_sum(double v, double *ws) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();  // (ws may still be read from a previous call)
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  return (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// e[col] = sum_p partial[p][col] + yscale * (*yty);  sums (optional): sums[blockIdx.x] = the chunk sum of this workgroup's 256
// entries of e (device or pinned host memory: the training loops read the mean energy without a second launch)
__global__ __launch_bounds__(256) void gaussian_energy_finish_kernel(const double *__restrict__ partial, int64_t ldp,
                                                                      int64_t nparts, int64_t j, double *__restrict__ e,
                                                                      double yscale, const double *__restrict__ yty,
                                                                      double *sums) {
  __shared__ double ws[4];
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (col < j) {
    double s = 0.0;
    for (int64_t p = 0; p < nparts; ++p) s += partial[p * ldp + col];
    v = s + yscale * (*yty);
    e[col] = v;
  }
  if (sums) {  // (kernel-uniform)
    const double t = chunk256_sum(v, ws);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void link_transform_kernel(int link, double jitter, const double *__restrict__ in,
                                                              int64_t ldin, int64_t rows, int64_t cols,
                                                              const double *__restrict__ col_offset,
                                                              double *__restrict__ out, int64_t ldout) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const double off = col_offset ? col_offset[col] : 0.0;
  for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
    double slope;
    out[row * ldout + col] = link_eval(link, in[row * ldin + col] + off, jitter, &slope);
  }
}

// out[row] = sum_j (S[row][j] - shift[row])^power: one block per row, fixed-order tree reduction
__global__ __launch_bounds__(256) void row_power_sums_kernel(const double *__restrict__ S, int64_t lds, int64_t cols,
                                                              const double *__restrict__ shift, int power,
                                                              double *__restrict__ out) {
  __shared__ double red[256];
  const double *row = S + (int64_t)blockIdx.x * lds;
  const double sh = shift ? shift[blockIdx.x] : 0.0;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < cols; k += 256) {
    const double v = row[k] - sh;
    s += (power == 2) ? v * v : v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// per-row quantiles: the row is sorted in LDS (bitonic network over the next power of two, +inf padding)
__global__ __launch_bounds__(256) void row_quantiles_kernel(const double *__restrict__ samples, int64_t lds_, int64_t cols,
                                                             int npad, const double *__restrict__ q, int nq,
                                                             double *__restrict__ out, int64_t ldout) {
  extern __shared__ __attribute__((aligned(16))) double qbuf[];
  __shared__ int has_nan;
  const int tid = threadIdx.x;
  const double *row = samples + (int64_t)blockIdx.x * lds_;
  if (tid == 0) has_nan = 0;
  __syncthreads();
  bool nan_seen = false;
  for (int k = tid; k < npad; k += 256) {
    const double v = (k < cols) ? row[k] : __builtin_inf();
    nan_seen |= (v != v);
    qbuf[k] = v;
  }
  if (nan_seen) has_nan = 1;
  __syncthreads();
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < (npad >> 1); t += 256) {
        const int pos = 2 * t - (t & (stride - 1));
        const int partner = pos + stride;
        const bool up = (pos & size) == 0;
        const double a = qbuf[pos], b = qbuf[partner];
        if ((a > b) == up) {
          qbuf[pos] = b;
          qbuf[partner] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int k = tid; k < nq; k += 256) {
    double r;
    if (has_nan) {
      r = __builtin_nan("");
    } else {
      const double pos = q[k] * (double)(cols - 1);
      const double lo_f = floor(pos);
      const int lo = (int)lo_f;
      const int hi = (lo + 1 < cols) ? lo + 1 : (int)cols - 1;
      const double w = pos - lo_f, a = qbuf[lo], b = qbuf[hi];
      r = (w < 0.5) ? a + w * (b - a) : b - (b - a) * (1.0 - w);  // torch.lerp's two-sided form
    }
    out[(int64_t)blockIdx.x * ldout + k] = r;
  }
}

// per-row quantiles of LONG rows (more samples than one LDS sort holds: a calibration split above 16384 points, the
// gathered samples of a J-sharded run): no sort at all, the two order statistics a quantile interpolates between are
// SELECTED.  A double maps to a 64-bit key with the same order; eight passes over the row, most significant byte first,
// each a 256-bin histogram of the next byte among the samples that still match the key prefix found so far, fix one
// byte of the k-th smallest key.  The upper neighbour is the same value (ties) or the smallest larger key: one more
// pass.  A workgroup serves up to SEL_NQ quantiles of one row at once (one histogram each); nothing is written but the
// results, no workspace.  Histogram updates are aggregated per wave (the leading bytes -- sign, exponent -- are the same
// for nearly every sample: 64 lanes would otherwise serialise on one LDS counter).
constexpr int SEL_NQ = 4;

__device__ __forceinline__ uint64_t order_key(double v) {  // monotone: v < w  <=>  key(v) < key(w)  (no NaN)
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(1024) void row_quantiles_select_kernel(const double *__restrict__ samples, int64_t lds_, int64_t cols,
                                                                    const double *__restrict__ q, int nq, int q0,
                                                                    double *__restrict__ out, int64_t ldout) {
  __shared__ unsigned int hist[SEL_NQ][256];
  __shared__ uint64_t prefix[SEL_NQ], minabove[SEL_NQ];
  __shared__ long long rank[SEL_NQ];
  __shared__ unsigned int tie[SEL_NQ];
  __shared__ int has_nan;
  const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;  // 256 threads, 1024 when there are only a few rows
  const double *row = samples + (int64_t)blockIdx.x * lds_;
  const int nloc = (nq - q0 < SEL_NQ) ? nq - q0 : SEL_NQ;  // quantiles of this workgroup: q0 .. q0 + nloc - 1
  if (tid < SEL_NQ) {
    prefix[tid] = 0;
    minabove[tid] = ~0ull;
    tie[tid] = 0;
    if (tid < nloc) rank[tid] = (long long)floor(q[q0 + tid] * (double)(cols - 1));  // lower order statistic, 0-based
  }
  if (tid == 0) has_nan = 0;
  for (int pass = 7; pass >= 0; --pass) {
    for (int e = tid; e < SEL_NQ * 256; e += nt) (&hist[0][0])[e] = 0;
    __syncthreads();
    const int shift = 8 * pass;
    uint64_t pf[SEL_NQ];
#pragma unroll
    for (int i = 0; i < SEL_NQ; ++i) pf[i] = prefix[i];
    bool nan_seen = false;
    for (int64_t k0 = 0; k0 < cols; k0 += nt) {
      const int64_t k = k0 + tid;
      const bool in = k < cols;
      const double v = in ? row[k] : 0.0;
      nan_seen |= (v != v);
      const uint64_t key = order_key(v);
      const unsigned byte = (unsigned)(key >> shift) & 255u;
#pragma unroll
      for (int i = 0; i < SEL_NQ; ++i) {
        if (i >= nloc) break;
        const bool match = in && (pass == 7 || ((key ^ pf[i]) >> (shift + 8)) == 0);
        unsigned long long todo = __ballot(match);
        while (todo) {  // one LDS update per distinct byte value in the wave
          const int leader = __ffsll((long long)todo) - 1;
          const unsigned lb = (unsigned)__shfl((int)byte, leader);
          const unsigned long long same = __ballot(match && byte == lb) & todo;
          if (lane == leader) atomicAdd(&hist[i][lb], (unsigned)__popcll(same));
          todo &= ~same;
        }
      }
    }
    if (pass == 7 && nan_seen) has_nan = 1;
    __syncthreads();
    if (tid < nloc) {  // the bin that holds the rank: serial scan of 256 counters (once per pass)
      long long r = rank[tid];
      unsigned b = 0;
      for (; b < 255; ++b) {
        const unsigned c = hist[tid][b];
        if (r < (long long)c) break;
        r -= c;
      }
      rank[tid] = r;
      prefix[tid] |= (uint64_t)b << shift;
      if (pass == 0) tie[tid] = hist[tid][b];  // samples equal to the selected value; r = position among them
    }
    __syncthreads();
  }
  // upper neighbour: the selected value again if the ties reach past it, otherwise the smallest larger key
  {
    uint64_t pf[SEL_NQ], best[SEL_NQ];
#pragma unroll
    for (int i = 0; i < SEL_NQ; ++i) {
      pf[i] = prefix[i];
      best[i] = ~0ull;
    }
    for (int64_t k = tid; k < cols; k += nt) {
      const uint64_t key = order_key(row[k]);
#pragma unroll
      for (int i = 0; i < SEL_NQ; ++i)
        if (key > pf[i] && key < best[i]) best[i] = key;
    }
#pragma unroll
    for (int i = 0; i < SEL_NQ; ++i)
      if (i < nloc) atomicMin(reinterpret_cast<unsigned long long *>(&minabove[i]), (unsigned long long)best[i]);
    __syncthreads();
  }
  if (tid < nloc) {
    double r;
    if (has_nan) {
      r = __builtin_nan("");
    } else {
      const double pos = q[q0 + tid] * (double)(cols - 1);
      const double lo_f = floor(pos), w = pos - lo_f;
      const double a = key_value(prefix[tid]);
      const bool last = (int64_t)lo_f + 1 >= cols;
      const double b = (last || rank[tid] + 1 < (long long)tie[tid]) ? a : key_value(minabove[tid]);
      r = (w < 0.5) ? a + w * (b - a) : b - (b - a) * (1.0 - w);  // torch.lerp's two-sided form
    }
    out[(int64_t)blockIdx.x * ldout + q0 + tid] = r;
  }
}

__global__ __launch_bounds__(256) void counter_add_kernel(uint64_t *counter, uint64_t increment) { *counter += increment; }

__global__ __launch_bounds__(256) void normal_fill_kernel(double *__restrict__ out, int64_t ldo, int64_t rows,
                                                           int64_t j, uint64_t seed, uint64_t step, int64_t j_offset,
                                                           const uint64_t *__restrict__ step_base, int64_t block_cols) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= j) return;
  if (step_base) step += *step_base;
  j_offset += (block_cols > 0 ? col % block_cols : col) - col;  // Philox column = j_offset + column inside the block
  const int64_t npairs = cdiv(rows, 8) * 4;
  for (int64_t pr = blockIdx.y; pr < npairs; pr += gridDim.y) {
    const int64_t ib = (pr >> 2) * 8 + (pr & 3);
    if (ib >= rows) continue;
    double z0, z1;
    normal_pair(seed, step, ib, j_offset + col, z0, z1);
    out[ib * ldo + col] = z0;
    if (ib + 4 < rows) out[(ib + 4) * ldo + col] = z1;
  }
}

// c = A y  (Mk rows, N long): one block per row, deterministic tree reduction
__global__ __launch_bounds__(256) void matvec_rows_kernel(const double *__restrict__ A, int64_t lda, int64_t n,
                                                           const double *__restrict__ y, double *__restrict__ c) {
  __shared__ double red[256];
  const double *row = A + (int64_t)blockIdx.x * lda;
  double s = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += 256) s = fma(row[k], y[k], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) c[blockIdx.x] = red[0];
}

// out[i] = alpha * in[i], i < n
__global__ __launch_bounds__(256) void scale_copy_kernel(const double *__restrict__ in, double alpha, double *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = alpha * in[i];
}

__global__ __launch_bounds__(256) void scale_copy_2d_kernel(const double *__restrict__ in, int64_t ldin, double *__restrict__ out,
                                                             int64_t ldout, int64_t rows, int64_t cols, double alpha) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < rows * cols) out[(i / cols) * ldout + i % cols] = alpha * in[(i / cols) * ldin + i % cols];
}

// ---------------------------------------------------------------------------------------------------------------
// greedy conditional-variance inducing-point selection (reference: src/inducing_point_selectors/conditional_variance.py)
// ---------------------------------------------------------------------------------------------------------------
struct CvState {       // device-resident scalars of the selection loop
  int64_t pivot;       // index chosen for the current iteration
  double pivot_d;      // d[pivot] before this iteration's update
  int64_t count;       // points selected so far
  int64_t stopped;     // the threshold test fired: later iterations are no-ops
};

__device__ inline double cv_kernel_eval(int kind, const double *__restrict__ x, int64_t a, int64_t b, int d,
                                        const double *__restrict__ inv_ls, double outputscale) {
  double s = 0.0;
  if (kind == PLS_KERNEL_RBF_ARD) {
    for (int k = 0; k < d; ++k) {
      const double e = (x[a * d + k] - x[b * d + k]) * inv_ls[k];
      s = fma(e, e, s);
    }
    return outputscale * exp(-0.5 * s);
  }
  for (int k = 0; k < d; ++k) s = fma(x[a * d + k], x[b * d + k], s);
  return s;
}

constexpr int CV_BLOCK = 256;

// d[n] = k(x_n, x_n) + jitter; chosen[n] = 0
__global__ __launch_bounds__(CV_BLOCK) void cv_init_kernel(int kind, const double *__restrict__ x, int64_t n, int d,
                                                            const double *__restrict__ lengthscale, double outputscale,
                                                            double jitter, double *__restrict__ di,
                                                            unsigned char *__restrict__ chosen, CvState *st) {
  __shared__ double inv_ls[64];
  if ((int)threadIdx.x < d) inv_ls[threadIdx.x] = (kind == PLS_KERNEL_RBF_ARD) ? 1.0 / lengthscale[threadIdx.x] : 1.0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * CV_BLOCK + threadIdx.x;
  if (i == 0) {
    st->count = 0;
    st->stopped = 0;
  }
  if (i >= n) return;
  di[i] = cv_kernel_eval(kind, x, i, i, d, inv_ls, outputscale) + jitter;
  chosen[i] = 0;
}

// stage 1 of the pivot search: per block the largest d among the not-yet-chosen points (smaller index wins ties) and sum(d)
__global__ __launch_bounds__(CV_BLOCK) void cv_argmax_partial_kernel(const double *__restrict__ di,
                                                                      const unsigned char *__restrict__ chosen, int64_t n,
                                                                      double *__restrict__ pval, int64_t *__restrict__ pidx,
                                                                      double *__restrict__ psum) {
  __shared__ double sv[CV_BLOCK], ss[CV_BLOCK];
  __shared__ int64_t si[CV_BLOCK];
  double best = -1.0, sum = 0.0;
  int64_t bi = -1;
  for (int64_t i = (int64_t)blockIdx.x * CV_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * CV_BLOCK) {
    const double v = di[i];
    sum += v;
    if (!chosen[i] && v > best) {
      best = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  ss[threadIdx.x] = sum;
  __syncthreads();
  for (int w = CV_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      const double ov = sv[threadIdx.x + w];
      const int64_t oi = si[threadIdx.x + w];
      if (oi >= 0 && (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && (si[threadIdx.x] < 0 || oi < si[threadIdx.x])))) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
      ss[threadIdx.x] += ss[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pval[blockIdx.x] = sv[0];
    pidx[blockIdx.x] = si[0];
    psum[blockIdx.x] = ss[0];
  }
}

// stage 2: one block; records the pivot (and its d), appends it to `indices`, applies the threshold rule
//   first != 0: the very first pick (argmax of the diagonal, no threshold test, conditional_variance.py:70)
__global__ __launch_bounds__(CV_BLOCK) void cv_argmax_final_kernel(const double *__restrict__ pval,
                                                                    const int64_t *__restrict__ pidx,
                                                                    const double *__restrict__ psum, int nparts,
                                                                    const double *__restrict__ di,
                                                                    unsigned char *__restrict__ chosen, int64_t *indices,
                                                                    int64_t m, double threshold, int first, CvState *st) {
  __shared__ double sv[CV_BLOCK], ss[CV_BLOCK];
  __shared__ int64_t si[CV_BLOCK];
  double best = -1.0, sum = 0.0;
  int64_t bi = -1;
  for (int p = threadIdx.x; p < nparts; p += CV_BLOCK) {
    const double v = pval[p];
    const int64_t i = pidx[p];
    sum += psum[p];
    if (i >= 0 && (v > best || (v == best && (bi < 0 || i < bi)))) {
      best = v;
      bi = i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  ss[threadIdx.x] = sum;
  __syncthreads();
  for (int w = CV_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      const double ov = sv[threadIdx.x + w];
      const int64_t oi = si[threadIdx.x + w];
      if (oi >= 0 && (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && (si[threadIdx.x] < 0 || oi < si[threadIdx.x])))) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
      ss[threadIdx.x] += ss[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && !st->stopped) {
    if (st->count < m && si[0] >= 0) {
      st->pivot = si[0];
      st->pivot_d = di[si[0]];
      indices[st->count] = si[0];
      chosen[si[0]] = 1;
      st->count += 1;
    }
    // conditional_variance.py:108-113: after picking, stop if tr(Kff - Qff) = sum(d) fell below the threshold
    if (!first && ss[0] < threshold) st->stopped = 1;
  }
}

// one greedy iteration i (0-based): e = (round20(k(X, x_j)) + jitter [n == j] - c[:i, j] . c[:i, n]) / sqrt(d_j)
// The dot product over the i previous rows is the whole cost (M^2/2 * N * 8 bytes over the run): a block covers
// CV_COLS columns with CV_BLOCK / CV_COLS row slices (wave w takes rows w, w + 4, ...; two independent partial sums
// each), so that 8x more loads are in flight than with one thread per column -- the N = 1e5 columns alone are 1.5
// waves per SIMD of dependent fma chains; the slices are summed through LDS in a fixed order.
constexpr int CV_COLS = 64;
__global__ __launch_bounds__(CV_BLOCK) void cv_update_kernel(int kind, const double *__restrict__ x, int64_t n, int d,
                                                              const double *__restrict__ lengthscale, double outputscale,
                                                              double jitter, int64_t iter, double *__restrict__ ci,
                                                              double *__restrict__ di, const CvState *__restrict__ st) {
  constexpr int NSL = CV_BLOCK / CV_COLS;
  __shared__ double inv_ls[64];
  __shared__ double part[NSL][CV_COLS];
  if ((int)threadIdx.x < d) inv_ls[threadIdx.x] = (kind == PLS_KERNEL_RBF_ARD) ? 1.0 / lengthscale[threadIdx.x] : 1.0;
  __syncthreads();
  if (st->stopped || st->count != iter + 1) return;  // (stopped early, or ran out of candidates; uniform)
  const int64_t j = st->pivot;
  const double dj = sqrt(st->pivot_d);
  const int c = threadIdx.x % CV_COLS, sl = threadIdx.x / CV_COLS;
  const int64_t col = (int64_t)blockIdx.x * CV_COLS + c;
  const bool in = col < n;
  const int64_t cc = in ? col : 0;
  double d0 = 0.0, d1 = 0.0;
  int64_t t = sl;
  for (; t + NSL < iter; t += 2 * NSL) {
    d0 = fma(ci[t * n + j], ci[t * n + cc], d0);
    d1 = fma(ci[(t + NSL) * n + j], ci[(t + NSL) * n + cc], d1);
  }
  if (t < iter) d0 = fma(ci[t * n + j], ci[t * n + cc], d0);
  part[sl][c] = d0 + d1;
  __syncthreads();
  if (sl != 0 || !in) return;
  double dot = part[0][c];
#pragma unroll
  for (int k = 1; k < NSL; ++k) dot += part[k][c];
  double g = cv_kernel_eval(kind, x, col, j, d, inv_ls, outputscale);
  g = rint(g * 1e20) / 1e20;  // np.round(., 20) (conditional_variance.py:93)
  if (col == j) g += jitter;
  const double e = (g - dot) / dj;
  ci[iter * n + col] = e;
  const double nd = di[col] - e * e;
  di[col] = nd > 0.0 ? nd : 0.0;
}

static unsigned rows_grid(int64_t items) { return (unsigned)(items < 1 ? 1 : (items > 1024 ? 1024 : items)); }

// ---------------------------------------------------------------------------------------------------------------
// shared host helpers for the fused paths
// ---------------------------------------------------------------------------------------------------------------
static int validate_cost(const pls_cost_desc *c) {
  PLS_REQUIRE(c != nullptr, "cost descriptor is NULL");
  PLS_REQUIRE(c->cost >= PLS_COST_GAUSSIAN && c->cost <= PLS_COST_MULTIMODAL, "unknown cost kind %d", c->cost);
  PLS_REQUIRE(c->link >= PLS_LINK_IDENTITY && c->link <= PLS_LINK_PROBIT, "unknown link kind %d", c->link);
  PLS_REQUIRE(c->deriv_mode == PLS_DERIV_REFERENCE || c->deriv_mode == PLS_DERIV_AUTOGRAD, "unknown deriv_mode %d",
              c->deriv_mode);
  if (c->cost == PLS_COST_GAUSSIAN) PLS_REQUIRE(c->p[0] > 0.0, "gaussian cost needs observation_noise > 0");
  if (c->cost == PLS_COST_STUDENT_T)
    PLS_REQUIRE(c->p[0] > 0.0 && c->p[1] > 0.0, "student-t cost needs degrees_of_freedom > 0 and scale > 0");
  if (c->cost == PLS_COST_MULTIMODAL)
    PLS_REQUIRE(c->p[0] > 0.0 && c->p[2] > 0.0 && c->p[2] < 1.0, "multimodal cost needs sigma > 0, 0 < bernoulli_noise < 1");
  return PLS_OK;
}

static int validate_noise(const pls_noise_desc *n, int64_t rows, int64_t j) {
  if (!n) return PLS_OK;
  PLS_REQUIRE(n->kind >= PLS_NOISE_NONE && n->kind <= PLS_NOISE_PHILOX, "unknown noise kind %d", n->kind);
  if (n->kind == PLS_NOISE_INJECTED) {
    PLS_REQUIRE(n->xi != nullptr && n->ldxi > 0, "injected noise needs xi and ldxi");
    // (the kernels read xi[row * ldxi + column] for every particle column: a narrower matrix would be read out of bounds)
    PLS_REQUIRE(n->ldxi >= j, "injected noise: leading dimension %lld < %lld particle columns", (long long)n->ldxi, (long long)j);
  }
  (void)rows;
  return PLS_OK;
}

// ---- small projection ranks: fused kernels (small_rank.h) -------------------------------------------------------
static thread_local RouteOption g_small_rank_max{128};  // pls_set_option(PLS_OPT_SMALL_RANK_MAX)
static thread_local RouteOption g_ipb_explicit_inverse{0};  // pls_set_option(PLS_OPT_IPB_EXPLICIT_INVERSE)
static thread_local RouteOption g_kg_noise_pregen{1};      // pls_set_option(PLS_OPT_KG_NOISE_PREGEN): Philox noise in front of the k-split kernel's k-loop
static thread_local RouteOption g_energy_fused_finish{1};   // pls_set_option(PLS_OPT_ENERGY_FUSED_FINISH): honour pls_block_desc.energy_sync
static thread_local RouteOption g_ipb_step_operator{1};     // pls_set_option(PLS_OPT_IPB_STEP_OPERATOR): 1 = Pt route when the descriptor has it
static thread_local RouteOption g_small_rank_step{1};  // pls_set_option(PLS_OPT_SMALL_RANK_STEP): 0 never, 1 launch-bound problems, 2 wherever it applies
static thread_local RouteOption g_ipb_prep{1};         // pls_set_option(PLS_OPT_IPB_PREP): solve + coloured noise of a small inducing-point step in one launch
static thread_local RouteOption g_solve_mode{1};  // pls_set_option(PLS_OPT_SOLVE_MODE): 0 block substitution, 1 inverse-factor products where available
int64_t solve_mode() { return g_solve_mode.load(); }

static bool small_rank_ok(const double *Lb, int64_t ldlb, int64_t kdim) {
  return kdim >= 1 && kdim <= g_small_rank_max.load() && (ldlb & 1) == 0 && (reinterpret_cast<uintptr_t>(Lb) & 15) == 0;
}

// (the launchers of the fused small-rank kernels live in their own translation units: small_rank_launch.h)

// Streams N in chunks:  G_c = cost'(Lf[:, chunk]^T V)  ->  D (+)= Lb[chunk, :]^T G_c.
//   Lf (K x N, ldlf): forward operand (A or Kzx), V (K x J) particles in the basis the forward map expects
//   Lb (N x K, ldlb): back-projection operand (At or Kxz)
//   D  (K x J): receives the drift  Lb^T cost'(...)
// Optional by-product of stream_drift: the energy of the particles the drift is evaluated at (cost value of the same F
// plus the prior term), so that a training loop needs no separate energy pass.
struct EnergySink {
  double *partial = nullptr;  // [rows_cap][j] workspace for per-wave-row (or per-slab) cost partial sums
  int64_t rows_cap = 0;
  double *e = nullptr;  // (j,) receives cost_j + prior_j
  int prior_kind = 0;
  const double *P = nullptr;  // prior operand (U for the ONB, V = K^-1 U for the IPB)
  int64_t ldp = 0, m = 0;
  const double *lam = nullptr;
  double scale = 0.0;
};

static int stream_drift(const double *Lf, int64_t ldlf, const double *Lb, int64_t ldlb, int64_t kdim, int64_t n,
                        const double *V, int64_t ldv, int64_t j, const CostP &cp, const double *y, double *D,
                        int64_t ldd, int64_t max_slabs, int64_t slab_stride, int64_t *slabs_used, double *Gbuf,
                        int64_t n_chunk, hipStream_t st, const EnergySink *es = nullptr) {
  auto reduce_partials = [&](int64_t rows, int accumulate, bool last) {
    hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, es->partial, j, rows, j, es->e,
                       accumulate, last ? es->prior_kind : 0, es->P, es->ldp, es->m, es->lam, es->scale, 1.0,
                       (const double *)nullptr);
    return check_launch("column_reduce");
  };
  if (small_rank_ok(Lb, ldlb, kdim)) {  // few basis functions: G stays in registers (small_rank.h)
    int64_t rows_per_split = 0;
    const int64_t ns = small_rank_splits(j, n, &rows_per_split);
    if (ns <= max_slabs && (!es || ns <= es->rows_cap)) {
      SmallRankP p{Lb, ldlb, V, ldv, y, n, j, (int)kdim, rows_per_split, D, ldd, slab_stride, cp, es ? es->partial : nullptr, j};
      *slabs_used = ns;
      int rc = es ? launch_small_rank_drift_value(p, ns, st) : launch_small_rank_drift(p, ns, st);
      if (rc || !es) return rc;
      return reduce_partials(ns, 0, true);
    }
  }
  // one split-K plan for every chunk (slab s accumulates over the chunks; the update kernel sums the slabs)
  int64_t kchunk = 0;
  int64_t nslab = plan_split_k(kdim, j, n < n_chunk ? n : n_chunk, &kchunk);
  if (nslab > max_slabs) {
    nslab = 1;
    kchunk = 0;
  }
  *slabs_used = nslab;
  for (int64_t r0 = 0, c = 0; r0 < n; r0 += n_chunk, ++c) {
    const int64_t rows = (n - r0 < n_chunk) ? (n - r0) : n_chunk;
    int rc;
    // wave rows of the forward launch: 64 data rows each with the 128x128 tiles, 32 with the 64x64 ones
    const int64_t wave_rows = cdiv(rows, use_big_tiles(rows, j, 1) ? 64 : 32);
    double *vp = nullptr;
    if (es) {
      if (wave_rows > es->rows_cap)
        return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "step energy by-product: %lld partial rows > %lld", (long long)wave_rows,
                    (long long)es->rows_cap);
      vp = es->partial;
    }
    rc = launch_cost_deriv_gemm(Lf + r0, ldlf, V, ldv, rows, j, kdim, Gbuf, j, y + r0, cp, vp, j, st);
    if (rc) return rc;
    if (es) {
      rc = reduce_partials(wave_rows, c == 0 ? 0 : 1, r0 + n_chunk >= n);
      if (rc) return rc;
    }
    // slab s accumulates rows [s * kchunk, (s + 1) * kchunk) of every chunk; the first chunk has the planned row count,
    // so it writes (beta = 0) every slab; a shorter last chunk simply leaves its missing slabs untouched
    EpiStore e2{D, ldd, 1.0, c == 0 ? 0.0 : 1.0, slab_stride};
    const int64_t kc2 = nslab > 1 ? kchunk : 0;
    const int64_t main_rows = kdim / 128 * 128, rem_rows = kdim - main_rows;
    if (gemm_rows_ok(Lb + r0 * ldlb, ldlb, Gbuf, j, kdim, j, rows, ldd, nslab)) {
      // a rank that is not a multiple of 128 in ONE launch whose MFMA count follows cdiv(kdim, 16)
      rc = launch_gemm_rows(Lb + r0 * ldlb, ldlb, Gbuf, j, kdim, j, rows, e2, st, kc2);
    } else if (main_rows > 0 && rem_rows > 0 && rem_rows <= 112 && use_big_tiles(kdim, j, nslab)) {
      // a rank that is not a multiple of 128 (129: one more 128-row tile would compute 256 rows for 129): the full
      // 128-row tiles with the big configuration, the remainder in pieces of 64, 32 and 16 rows (at most 15 idle rows)
      rc = launch_gemm(Lb + r0 * ldlb, ldlb, Gbuf, j, main_rows, j, rows, e2, st, kc2);
      for (int64_t at = main_rows; !rc && at < kdim;) {
        const int64_t left = kdim - at;
        EpiStore e3{D + at * ldd, ldd, 1.0, c == 0 ? 0.0 : 1.0, slab_stride};
        if (left > 32) {
          const int64_t take = left < 64 ? left : 64;
          GemmShape g3{Lb + r0 * ldlb + at, ldlb, Gbuf, j, take, j, rows, 0, 0, kc2};
          rc = launch_gemm_cfg<64, 64, 32, 32>(g3, e3, st);
          at += take;
        } else if (left > 16) {
          GemmShape g3{Lb + r0 * ldlb + at, ldlb, Gbuf, j, left, j, rows, 0, 0, kc2};
          rc = launch_gemm_cfg<32, 128, 32, 32>(g3, e3, st);
          at += left;
        } else {
          GemmShape g3{Lb + r0 * ldlb + at, ldlb, Gbuf, j, left, j, rows, 0, 0, kc2};
          rc = launch_gemm_cfg<16, 128, 16, 32>(g3, e3, st);
          at += left;
        }
      }
    } else {
      rc = launch_gemm(Lb + r0 * ldlb, ldlb, Gbuf, j, kdim, j, rows, e2, st, kc2);
    }
    if (rc) return rc;
  }
  return PLS_OK;
}

// c_j partials over N chunks -> cost_out[j] (deterministic)
static int stream_cost(const double *Lf, int64_t ldlf, const double *Lb, int64_t ldlb, int64_t kdim, int64_t n,
                       const double *V, int64_t ldv, int64_t j, const CostP &cp, const double *y, double *partial,
                       int64_t partial_rows, int64_t n_chunk, double *e_out, int prior_kind, const double *P, int64_t ldp,
                       int64_t m, const double *lam, double scale, hipStream_t st) {
  if (Lb && small_rank_ok(Lb, ldlb, kdim)) {  // few basis functions: one fused pass, F never written
    int64_t rows_per_split = 0;
    const int64_t ns = small_rank_splits(j, n, &rows_per_split);
    if (ns <= partial_rows) {
      SmallRankP p{Lb, ldlb, V, ldv, y, n, j, (int)kdim, rows_per_split, partial, j, j, cp, nullptr, 0};
      int rc = launch_small_rank_value(p, ns, st);
      if (rc) return rc;
      hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, partial, j, ns, j, e_out, 0,
                         prior_kind, P, ldp, m, lam, scale, 1.0, (const double *)nullptr);
      return check_launch("column_reduce");
    }
  }
  int64_t nchunks = cdiv(n, n_chunk);
  for (int64_t r0 = 0, c = 0; r0 < n; r0 += n_chunk, ++c) {
    const int64_t rows = (n - r0 < n_chunk) ? (n - r0) : n_chunk;
    int rc = launch_cost_value_gemm(Lf + r0, ldlf, V, ldv, rows, j, kdim, partial, j, y + r0, cp, st);
    if (rc) return rc;
    const bool last = (c == nchunks - 1);
    hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, partial, j,
                       cost_value_partial_rows(rows, j), j, e_out, c == 0 ? 0 : 1, last ? prior_kind : 0, P, ldp, m,
                       lam, scale, 1.0, (const double *)nullptr);
    rc = check_launch("column_reduce");
    if (rc) return rc;
  }
  return PLS_OK;
}

int gemm_tn_ex(const double *L, int64_t ldl, const double *R, int64_t ldr, double *C, int64_t ldc, int64_t I, int64_t J,
               int64_t K, double alpha, double beta, int tri, hipStream_t st, void *tri_scratch, size_t tri_scratch_bytes) {
  PLS_REQUIRE(L && R && C, "gemm_tn: NULL pointer");
  PLS_REQUIRE(I >= 0 && J >= 0 && K >= 0, "gemm_tn: negative size");
  PLS_REQUIRE(ldl >= I && ldr >= J && ldc >= J, "gemm_tn: leading dimension too small (ldl=%lld I=%lld ldr=%lld J=%lld ldc=%lld)",
              (long long)ldl, (long long)I, (long long)ldr, (long long)J, (long long)ldc);
  if (I == 0 || J == 0) return PLS_OK;
  EpiStore e{C, ldc, alpha, beta, 0};
  // many tiles and a row count off the 128-row grid (the projection A = V~^T k(Z,X) of a thresholded basis): row blocks
  if (tri == 0 && gemm_rows_ok(L, ldl, R, ldr, I, J, K, ldc, 1)) return launch_gemm_rows(L, ldl, R, ldr, I, J, K, e, st, 0);
  return launch_gemm_any(L, ldl, R, ldr, I, J, K, e, st, 0, tri, TriScratch{tri_scratch, tri_scratch_bytes});
}

// out[b] = mean of e[b * bc, min(j, (b + 1) * bc)): one workgroup per column block; chunk sums of 256 consecutive entries
// (chunk256_sum's order) added in ascending order
__global__ __launch_bounds__(256) void block_means_kernel(const double *__restrict__ e, int64_t j, int64_t bc, double *out) {
  __shared__ double wsum[512][4];
  const int64_t c0 = (int64_t)blockIdx.x * bc;
  const int64_t c1 = (c0 + bc < j) ? c0 + bc : j;
  const int64_t nchunk = cdiv(c1 - c0, 256);
  double total = 0.0;
  for (int64_t k0 = 0; k0 < nchunk; k0 += 512) {
    const int64_t kb = (nchunk - k0 < 512) ? nchunk - k0 : 512;
    for (int64_t k = 0; k < kb; ++k) {
      const int64_t c = c0 + (k0 + k) * 256 + threadIdx.x;
      double v = (c < c1) ? e[c] : 0.0;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if ((threadIdx.x & 63) == 0) wsum[k][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int64_t k = 0; k < kb; ++k) total += (wsum[k][0] + wsum[k][1]) + (wsum[k][2] + wsum[k][3]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = total / (double)(c1 - c0);
}

// out[i] = chunk sum of e[256 i, min(j, 256 (i + 1)))
__global__ __launch_bounds__(256) void chunk_sums_kernel(const double *__restrict__ e, int64_t j, double *out) {
  __shared__ double ws[4];
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const double t = chunk256_sum(c < j ? e[c] : 0.0, ws);
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// out[b] = e[16 b] + e[16 b + 1] + ... (ascending) over the columns of block b that exist
__global__ __launch_bounds__(256) void sums16_kernel(const double *__restrict__ e, int64_t j, double *out) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b * 16 >= j) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += (b * 16 + k < j) ? e[b * 16 + k] : 0.0;
  out[b] = s;
}

}  // namespace plship

using namespace plship;

// The Gaussian/identity step as a function of its operator: out = [U +] -eta (inv_noise (B U - c) + U / lam) + sqrt(2 eta) xi,
// one fused contraction.  The orthonormal basis passes (A A^T, A y, lam, 1 / sigma2); the inducing-point basis in
// whitened coordinates passes (Q, c~, NULL, 1) -- see pls_ipb_build_whitened.  energy_in: per-particle energy of the INPUT
// particles from the same product,  sum_i u_i (inv_noise / 2 (B u)_i - inv_noise c_i) + u_i^2 / (2 lam_i) + yscale * yty.
struct FastOp {
  const double *B;
  int64_t ldb;
  const double *c, *lam;
  int64_t mk;
  double inv_noise, yscale;
  const double *yty;
};

// LAGGED energies of a training loop (pls_block_desc.energy_partials ...): launch k leaves only the partial rows of its energy
// by-product; launch k + 1 finishes them at its START (EpiLangevinGaussian::Prev), under the landing of its first operand
// rows, so no launch carries the reduction's serial tail (4.4-5 us at the end of every step launch otherwise).
struct EnergyLag {
  double *partials_out = nullptr;
  const double *partials_prev = nullptr;
  double *e_prev = nullptr;
  double *sums_prev = nullptr;
  int flush = 0;
};

static EnergyLag make_lag(const pls_block_desc *b) {
  EnergyLag l;
  if (b) {
    l.partials_out = b->energy_partials;
    l.partials_prev = b->energy_partials_prev;
    l.e_prev = b->energy_prev;
    l.sums_prev = b->energy_sums_prev;
    l.flush = b->energy_flush;
  }
  return l;
}

static int validate_lag(const pls_block_desc *b) {
  if (!b) return PLS_OK;
  PLS_REQUIRE(!b->energy_partials_prev || b->energy_prev, "step_blocks: energy_partials_prev needs energy_prev");
  PLS_REQUIRE(!b->energy_flush || b->energy_partials_prev, "step_blocks: energy_flush needs energy_partials_prev");
  PLS_REQUIRE(!b->energy_partials || b->energy_partials != b->energy_partials_prev,
              "step_blocks: energy_partials and energy_partials_prev must be different buffers");
  return PLS_OK;
}

static int fast_step_launch(const FastOp &op, const double *U, int64_t ldu, int64_t j, const EtaP &etap, const NoiseP &nz,
                            double *out, int64_t ldo, int out_mode, double *energy_in, void *workspace,
                            size_t workspace_bytes, hipStream_t st, const char *who, double *esums = nullptr,
                            uint32_t *esync = nullptr, const EnergyLag &lag = EnergyLag{}) {
  const bool big = pick_gemm_cfg(op.B, op.ldb, U, ldu, op.mk, j, op.mk) == CFG_BIG;
  const int64_t parts = big ? 2 * cdiv(op.mk, 128) : cdiv(op.mk, 64);  // partial rows THIS launch's tiling leaves
  // A LAGGED buffer is finished by another launch, whose tiling may differ (the choice follows the alignment and leading
  // dimension of its own particle tensor and the k-split options): such a buffer always holds 2 cdiv(mk, 128) rows -- what
  // pls_energy_partials_bytes sizes it for -- and a launch that writes fewer zeroes the rest, so that whoever finishes it adds
  // the same rows whatever either launch chose.
  const int64_t lag_rows = 2 * cdiv(op.mk, 128);
  if (lag.flush) {  // no step: the partial rows of the LAST launch of a loop are finished by the finishing kernel
    hipLaunchKernelGGL(gaussian_energy_finish_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, lag.partials_prev, j, lag_rows, j,
                       lag.e_prev, op.yscale, op.yty, lag.sums_prev);
    return check_launch("gaussian_energy_finish");
  }
  double *epart = lag.partials_out;
  if (energy_in && !epart) {
    if (!workspace || workspace_bytes < (size_t)parts * j * sizeof(double))
      return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "%s: energy by-product needs %zu workspace bytes", who,
                  (size_t)parts * j * sizeof(double));
    epart = static_cast<double *>(workspace);
  }
  EpiLangevinGaussian e{out, ldo, U, ldu, op.c, op.lam, etap, op.inv_noise, out_mode, nz, epart, j, 2, big ? 128 : 64, {}, {}};
  e.pregen_flag = g_kg_noise_pregen.load() != 0;
  if (lag.partials_prev) e.prev = EpiLangevinGaussian::Prev{lag.partials_prev, lag.e_prev, lag.sums_prev, op.yscale, op.yty, (int)lag_rows};
  const bool lagged = lag.partials_out != nullptr;  // (the NEXT launch, or a flush, finishes this launch's energies)
  if (lagged && parts < lag_rows) e.zero_row = (int)parts;  // (lag_rows - parts is 0 or 1)
  const bool fused_finish = !lagged && energy_in && esync && g_energy_fused_finish.load() != 0;
  if (fused_finish)  // the step launch finishes the energies itself (pls_block_desc.energy_sync)
    e.fin = EpiLangevinGaussian::Finish{esync, energy_in, esums, op.yscale, op.yty, (int)parts, (int)cdiv(op.mk, big ? 128 : 64)};
  int rc = launch_gemm_any(op.B, op.ldb, U, ldu, op.mk, j, op.mk, e, st);
  if (rc || lagged || !energy_in || fused_finish) return rc;
  hipLaunchKernelGGL(gaussian_energy_finish_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, epart, j, parts, j,
                     energy_in, op.yscale, op.yty, esums);
  return check_launch("gaussian_energy_finish");
}

// e_j = (inv_noise / 2) u^T B u - inv_noise c^T u + sum_i u_i^2 / (2 lam_i) + yscale * yty: one contraction, reduced per
// tile then per column
static int fast_energy_launch(const FastOp &op, const double *U, int64_t ldu, int64_t j, double *e, void *workspace,
                              size_t workspace_bytes, hipStream_t st, const char *who) {
  const GemmCfg cfg = pick_gemm_cfg(op.B, op.ldb, U, ldu, op.mk, j, op.mk);
  const int64_t parts = cfg == CFG_BIG ? cdiv(op.mk, 128) : cdiv(op.mk, 64);
  if (!workspace || workspace_bytes < (size_t)parts * j * sizeof(double))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "%s: workspace %zu bytes too small", who, workspace_bytes);
  double *partial = static_cast<double *>(workspace);
  GemmShape g{op.B, op.ldb, U, ldu, op.mk, j, op.mk, 0, 0, 0};
  const double ps = 0.5 * op.inv_noise;
  int rc;
  if (cfg == CFG_BIG) {
    EpiGaussianQuad<128, 128, 64, 64> ep{partial, j, U, ldu, op.c, op.lam, ps};
    rc = launch_gemm_cfg<128, 128, 64, 64>(g, ep, st);
  } else if (cfg == CFG_KG2) {
    EpiGaussianQuad<64, 64, 16, 32> ep{partial, j, U, ldu, op.c, op.lam, ps};
    rc = launch_gemm_kg<2>(g, ep, st);
  } else if (cfg == CFG_KG1) {
    EpiGaussianQuad<64, 64, 32, 32> ep{partial, j, U, ldu, op.c, op.lam, ps};
    rc = launch_gemm_kg<1>(g, ep, st);
  } else {
    EpiGaussianQuad<64, 64, 32, 32> ep{partial, j, U, ldu, op.c, op.lam, ps};
    rc = launch_gemm_cfg<64, 64, 32, 32>(g, ep, st);
  }
  if (rc) return rc;
  hipLaunchKernelGGL(gaussian_energy_finish_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, partial, j, parts, j, e,
                     op.yscale, op.yty, (double *)nullptr);
  return check_launch("gaussian_energy_finish");
}


// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

const char *pls_last_error(void) { return g_last_error.c_str(); }
int pls_abi_version(void) { return PLSHIP_ABI_VERSION; }

size_t pls_tri_scratch_bytes(int64_t m, int64_t j) { return (m > 0 && j > 0) ? kg_tri_scratch_bytes(m, j) : 0; }

size_t pls_step_sync_words(int64_t j) { return j > 0 ? small_rank_step_sync_words(j) : 0; }

size_t pls_energy_partials_bytes(int64_t rows, int64_t j) {
  return (rows > 0 && j > 0) ? (size_t)(2 * cdiv(rows, 128)) * j * sizeof(double) : 0;
}

int pls_set_option(int32_t option, int64_t value) {
  switch (option) {
    case PLS_OPT_SMALL_RANK_MAX:
      PLS_REQUIRE(value >= 0 && value <= 128, "set_option: small-rank limit %lld outside 0..128", (long long)value);
      g_small_rank_max.store(value);
      return PLS_OK;
    case PLS_OPT_IPB_EXPLICIT_INVERSE:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: ipb explicit inverse must be 0 or 1");
      g_ipb_explicit_inverse.store(value);
      return PLS_OK;
    case PLS_OPT_SOLVE_MODE:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: solve mode must be 0 or 1");
      g_solve_mode.store(value);
      return PLS_OK;
    case PLS_OPT_KSPLIT_MODE:
      PLS_REQUIRE(value >= 0 && value <= 3, "set_option: k-split mode must be 0..3");
      g_ksplit_mode.store(value);
      return PLS_OK;
    case PLS_OPT_KSPLIT_MAX_TILES:
      PLS_REQUIRE(value >= 0, "set_option: k-split tile limit must be >= 0");
      g_ksplit_max_tiles.store(value);
      return PLS_OK;
    case PLS_OPT_ROW_BLOCKS:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: row-block mode must be 0 or 1");
      g_row_blocks_mode.store(value);
      return PLS_OK;
    case PLS_OPT_TRI_BALANCE:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: triangular balance must be 0 or 1");
      g_tri_balance.store(value);
      return PLS_OK;
    case PLS_OPT_IPB_STEP_OPERATOR:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: step operator mode must be 0 or 1");
      g_ipb_step_operator.store(value);
      return PLS_OK;
    case PLS_OPT_ENERGY_FUSED_FINISH:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: fused energy finish must be 0 or 1");
      g_energy_fused_finish.store(value);
      return PLS_OK;
    case PLS_OPT_KG_NOISE_PREGEN:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: noise pre-generation must be 0 or 1");
      g_kg_noise_pregen.store(value);
      return PLS_OK;
    case PLS_OPT_SMALL_RANK_STEP:
      PLS_REQUIRE(value >= 0 && value <= 2, "set_option: small-rank step mode must be 0 (never), 1 (launch-bound) or 2 (always)");
      g_small_rank_step.store(value);
      return PLS_OK;
    case PLS_OPT_IPB_PREP:
      PLS_REQUIRE(value == 0 || value == 1, "set_option: ipb prep mode must be 0 or 1");
      g_ipb_prep.store(value);
      return PLS_OK;
    default: return fail(PLS_ERR_INVALID_ARGUMENT, "set_option: unknown option %d", (int)option);
  }
}

int pls_debug_math(int32_t op, const double *x, double *out, int64_t n, void *stream) {
  PLS_REQUIRE(op >= 0 && op <= 3, "debug_math: op must be 0 (exp), 1 (log), 2 (fast_div) or 3 (fast_div_normal)");
  PLS_REQUIRE(x && out && n >= 0, "debug_math: bad arguments");
  if (n == 0) return PLS_OK;
  hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, S(stream), op, x, out, n);
  return check_launch("debug_math");
}

int64_t pls_get_option(int32_t option) {
  switch (option) {
    case PLS_OPT_SMALL_RANK_MAX: return g_small_rank_max.load();
    case PLS_OPT_IPB_EXPLICIT_INVERSE: return g_ipb_explicit_inverse.load();
    case PLS_OPT_SOLVE_MODE: return g_solve_mode.load();
    case PLS_OPT_KSPLIT_MODE: return g_ksplit_mode.load();
    case PLS_OPT_KSPLIT_MAX_TILES: return g_ksplit_max_tiles.load();
    case PLS_OPT_ROW_BLOCKS: return g_row_blocks_mode.load();
    case PLS_OPT_TRI_BALANCE: return g_tri_balance.load();
    case PLS_OPT_IPB_STEP_OPERATOR: return g_ipb_step_operator.load();
    case PLS_OPT_ENERGY_FUSED_FINISH: return g_energy_fused_finish.load();
    case PLS_OPT_KG_NOISE_PREGEN: return g_kg_noise_pregen.load();
    case PLS_OPT_SMALL_RANK_STEP: return g_small_rank_step.load();
    case PLS_OPT_IPB_PREP: return g_ipb_prep.load();
    default: return -1;
  }
}

int pls_timeline_begin(int32_t capacity) {
  PLS_REQUIRE(capacity > 0 && capacity <= (1 << 20), "timeline_begin: capacity must be in (0, 2^20]");
  PLS_REQUIRE(!g_tl.on, "timeline_begin: a timeline is already running on this thread");
  g_tl.ev.resize(2 * (size_t)capacity);
  g_tl.tag.assign(capacity, 0);
  for (size_t i = 0; i < g_tl.ev.size(); ++i) {
    hipError_t e = hipEventCreate(&g_tl.ev[i]);
    if (e != hipSuccess) {
      for (size_t k = 0; k < i; ++k) (void)hipEventDestroy(g_tl.ev[k]);
      g_tl.ev.clear();
      return fail(PLS_ERR_HIP, "timeline_begin: hipEventCreate: %s", hipGetErrorString(e));
    }
  }
  g_tl.capacity = capacity;
  g_tl.count = 0;
  g_tl.on = true;
  return PLS_OK;
}

int pls_timeline_end(float *ms, int32_t *tags, int32_t capacity, int32_t *count) {
  PLS_REQUIRE(g_tl.on, "timeline_end: no timeline is running on this thread");
  g_tl.on = false;
  const int recorded = g_tl.count < g_tl.capacity ? g_tl.count : g_tl.capacity;
  int rc = PLS_OK;
  for (int i = 0; i < recorded; ++i) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(g_tl.ev[2 * i + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, g_tl.ev[2 * i], g_tl.ev[2 * i + 1]);
    if (e != hipSuccess && rc == PLS_OK) rc = fail(PLS_ERR_HIP, "timeline_end: %s", hipGetErrorString(e));
    if (ms && tags && i < capacity) {
      ms[i] = t;
      tags[i] = g_tl.tag[i];
    }
  }
  for (hipEvent_t e : g_tl.ev) (void)hipEventDestroy(e);
  g_tl.ev.clear();
  if (count) *count = g_tl.count;
  return rc;
}

int pls_kernel_gram(int32_t kernel_kind, const double *x1, int64_t n1, const double *x2, int64_t n2, int64_t d,
                    const double *lengthscale, double outputscale, double *out, int64_t ldout, void *stream) {
  PLS_REQUIRE(kernel_kind == PLS_KERNEL_RBF_ARD || kernel_kind == PLS_KERNEL_LINEAR, "unknown kernel kind %d", kernel_kind);
  PLS_REQUIRE(x1 && x2 && out, "kernel_gram: NULL pointer");
  PLS_REQUIRE(n1 >= 0 && n2 >= 0 && d >= 1, "kernel_gram: bad sizes n1=%lld n2=%lld d=%lld", (long long)n1, (long long)n2, (long long)d);
  PLS_REQUIRE(d <= 64, "kernel_gram: input dimension %lld > 64 is not supported", (long long)d);
  PLS_REQUIRE(ldout >= n2, "kernel_gram: ldout < n2");
  PLS_REQUIRE(kernel_kind != PLS_KERNEL_RBF_ARD || lengthscale, "kernel_gram: RBF needs lengthscale");
  if (n1 == 0 || n2 == 0) return PLS_OK;
  PLS_REQUIRE(cdiv(n2, 512) <= 0x7fffffff, "kernel_gram: n2 too large");
  const int64_t max_rows = 65535LL * GRAM_ROWS;  // gridDim.y <= 65535: taller Gram matrices go in row slabs
  for (int64_t r0 = 0; r0 < n1; r0 += max_rows) {
    const int64_t rows = (n1 - r0 < max_rows) ? n1 - r0 : max_rows;
    dim3 g2((unsigned)cdiv(n2, 512), (unsigned)cdiv(rows, GRAM_ROWS));
    LaunchScope scope(PLS_TAG_KERNEL_GRAM, S(stream));
    if (kernel_kind == PLS_KERNEL_RBF_ARD)
      launch_gram<PLS_KERNEL_RBF_ARD>(g2, S(stream), x1 + r0 * d, rows, x2, n2, (int)d, lengthscale, outputscale, out + r0 * ldout, ldout);
    else
      launch_gram<PLS_KERNEL_LINEAR>(g2, S(stream), x1 + r0 * d, rows, x2, n2, (int)d, lengthscale, outputscale, out + r0 * ldout, ldout);
    int rc = check_launch("kernel_gram");
    if (rc) return rc;
  }
  return PLS_OK;
}

int pls_gemm_tn(const double *L, int64_t ldl, const double *R, int64_t ldr, double *C, int64_t ldc, int64_t I, int64_t J,
                int64_t K, double alpha, double beta, void *stream) {
  PLS_REQUIRE(L && R && C, "gemm_tn: NULL pointer");
  PLS_REQUIRE(I >= 0 && J >= 0 && K >= 0, "gemm_tn: negative size");
  PLS_REQUIRE(ldl >= I && ldr >= J && ldc >= J, "gemm_tn: leading dimension too small (ldl=%lld I=%lld ldr=%lld J=%lld ldc=%lld)",
              (long long)ldl, (long long)I, (long long)ldr, (long long)J, (long long)ldc);
  return gemm_tn_ex(L, ldl, R, ldr, C, ldc, I, J, K, alpha, beta, 0, S(stream));
}

int pls_block_means(const double *e, int64_t j, int64_t block_cols, double *out, void *stream) {
  PLS_REQUIRE(e && out && j >= 0 && block_cols > 0, "block_means: bad arguments");
  if (j == 0) return PLS_OK;
  PLS_REQUIRE(cdiv(j, block_cols) <= 0x7fffffff, "block_means: too many blocks");
  hipLaunchKernelGGL(block_means_kernel, dim3((unsigned)cdiv(j, block_cols)), dim3(256), 0, S(stream), e, j, block_cols, out);
  return check_launch("block_means");
}

int pls_sums16(const double *e, int64_t j, double *out, void *stream) {
  PLS_REQUIRE(e && out && j >= 0, "sums16: bad arguments");
  if (j == 0) return PLS_OK;
  hipLaunchKernelGGL(sums16_kernel, dim3((unsigned)cdiv(cdiv(j, 16), 256)), dim3(256), 0, S(stream), e, j, out);
  return check_launch("sums16");
}

int pls_chunk_sums(const double *e, int64_t j, double *out, void *stream) {
  PLS_REQUIRE(e && out && j >= 0, "chunk_sums: bad arguments");
  if (j == 0) return PLS_OK;
  hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, S(stream), e, j, out);
  return check_launch("chunk_sums");
}

int pls_cost_derivative(const pls_cost_desc *cost, const double *F, int64_t ldf, const double *y, int64_t n, int64_t j,
                        double *G, int64_t ldg, void *stream) {
  int rc = validate_cost(cost);
  if (rc) return rc;
  PLS_REQUIRE(F && y && G, "cost_derivative: NULL pointer");
  PLS_REQUIRE(n >= 0 && j >= 0 && ldf >= j && ldg >= j, "cost_derivative: bad sizes");
  if (n == 0 || j == 0) return PLS_OK;
  hipLaunchKernelGGL(cost_deriv_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(n)), dim3(256), 0, S(stream),
                     make_costp(cost), F, ldf, y, n, j, G, ldg);
  return check_launch("cost_derivative");
}

size_t pls_cost_value_workspace_bytes(int64_t n, int64_t j) {
  if (n <= 0 || j <= 0) return 0;
  return (size_t)cdiv(n, COST_RB) * (size_t)j * sizeof(double);
}

int pls_cost_value(const pls_cost_desc *cost, const double *F, int64_t ldf, const double *y, int64_t n, int64_t j,
                   double *c, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_cost(cost);
  if (rc) return rc;
  PLS_REQUIRE(F && y && c, "cost_value: NULL pointer");
  PLS_REQUIRE(n >= 0 && j >= 0 && ldf >= j, "cost_value: bad sizes");
  if (j == 0) return PLS_OK;
  const size_t need = pls_cost_value_workspace_bytes(n, j);
  if (workspace_bytes < need || (need && !workspace))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "cost_value: workspace %zu < %zu bytes", workspace_bytes, need);
  double *partial = static_cast<double *>(workspace);
  const int64_t nparts = cdiv(n, COST_RB);
  if (nparts > 0) {
    PLS_REQUIRE(nparts <= 65535, "cost_value: n too large for one call");
    hipLaunchKernelGGL(cost_value_partial_kernel, dim3((unsigned)cdiv(j, 64), (unsigned)nparts), dim3(256), 0, S(stream),
                       make_costp(cost), F, ldf, y, n, j, partial, j);
    rc = check_launch("cost_value_partial");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, S(stream), partial, j, nparts, j, c,
                     0, 0, (const double *)nullptr, (int64_t)0, (int64_t)0, (const double *)nullptr, 0.0, 1.0, (const double *)nullptr);
  return check_launch("column_reduce");
}

int pls_link_transform(int32_t link, double jitter, const double *in, int64_t ldin, int64_t rows, int64_t cols,
                       const double *col_offset, double *out, int64_t ldout, void *stream) {
  PLS_REQUIRE(link >= PLS_LINK_IDENTITY && link <= PLS_LINK_PROBIT, "link_transform: unknown link %d", link);
  PLS_REQUIRE(in && out && rows >= 0 && cols >= 0 && ldin >= cols && ldout >= cols, "link_transform: bad arguments");
  if (rows == 0 || cols == 0) return PLS_OK;
  hipLaunchKernelGGL(link_transform_kernel, dim3((unsigned)cdiv(cols, 256), rows_grid(rows)), dim3(256), 0, S(stream),
                     link, jitter, in, ldin, rows, cols, col_offset, out, ldout);
  return check_launch("link_transform");
}

int pls_row_power_sums(const double *samples, int64_t lds, int64_t rows, int64_t cols, const double *shift, int32_t power,
                       double *out, void *stream) {
  PLS_REQUIRE(samples && out && rows >= 0 && cols >= 0 && lds >= cols, "row_power_sums: bad arguments");
  PLS_REQUIRE(power == 1 || power == 2, "row_power_sums: power must be 1 or 2");
  PLS_REQUIRE(rows <= 0x7fffffff, "row_power_sums: too many rows");
  if (rows == 0) return PLS_OK;
  hipLaunchKernelGGL(row_power_sums_kernel, dim3((unsigned)rows), dim3(256), 0, S(stream), samples, lds, cols, shift, power, out);
  return check_launch("row_power_sums");
}

int pls_row_quantiles(const double *samples, int64_t lds, int64_t rows, int64_t cols, const double *q, int32_t nq, double *out,
                      int64_t ldout, void *stream) {
  PLS_REQUIRE(samples && q && out, "row_quantiles: NULL pointer");
  PLS_REQUIRE(rows >= 0 && cols >= 1 && lds >= cols && nq >= 1 && ldout >= nq, "row_quantiles: bad sizes");
  PLS_REQUIRE(rows <= 0x7fffffff, "row_quantiles: too many rows");
  if (rows == 0) return PLS_OK;
  if (cols > 16384) {  // longer than one LDS sort: radix selection, SEL_NQ quantiles per launch
    for (int q0 = 0; q0 < nq; q0 += SEL_NQ) {
      hipLaunchKernelGGL(row_quantiles_select_kernel, dim3((unsigned)rows), dim3(rows < 256 ? 1024 : 256), 0, S(stream), samples, lds, cols, q,
                         (int)nq, q0, out, ldout);
      int rc = check_launch("row_quantiles_select");
      if (rc) return rc;
    }
    return PLS_OK;
  }
  int npad = 2;
  while (npad < cols) npad <<= 1;
  const size_t bytes = (size_t)npad * sizeof(double);
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(row_quantiles_kernel), 16384 * sizeof(double), lds_ready)) return rc;
  hipLaunchKernelGGL(row_quantiles_kernel, dim3((unsigned)rows), dim3(256), bytes, S(stream), samples, lds, cols, npad, q, (int)nq,
                     out, ldout);
  return check_launch("row_quantiles");
}

int pls_normal_fill(double *out, int64_t ldout, int64_t rows, int64_t j, uint64_t seed, uint64_t step, int64_t j_offset,
                    void *stream) {
  PLS_REQUIRE(out, "normal_fill: NULL pointer");
  PLS_REQUIRE(rows >= 0 && j >= 0 && ldout >= j, "normal_fill: bad sizes");
  if (rows == 0 || j == 0) return PLS_OK;
  hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(cdiv(rows, 8) * 4)), dim3(256), 0,
                     S(stream), out, ldout, rows, j, seed, step, j_offset, (const uint64_t *)nullptr, (int64_t)0);
  return check_launch("normal_fill");
}

int pls_counter_add(uint64_t *counter, uint64_t increment, void *stream) {
  PLS_REQUIRE(counter != nullptr, "counter_add: NULL pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, S(stream), counter, increment);
  return check_launch("counter_add");
}

// ---- orthonormal basis -----------------------------------------------------------------------------------------

static int validate_onb(const pls_onb_desc *b) {
  PLS_REQUIRE(b != nullptr, "onb descriptor is NULL");
  PLS_REQUIRE(b->mk > 0 && b->n > 0, "onb: mk and n must be positive");
  PLS_REQUIRE(b->A && b->At && b->lam, "onb: A, At and lam must be set");
  PLS_REQUIRE(b->lda >= b->n && b->ldat >= b->mk, "onb: leading dimension too small");
  if (b->B) PLS_REQUIRE(b->ldb >= b->mk && b->c, "onb: B needs ldb >= mk and c");
  return PLS_OK;
}

int pls_onb_build_projection(const double *Vs, int64_t ldvs, const double *Kzx, int64_t ldkzx, int64_t m, int64_t mk,
                             int64_t n, double *A, int64_t lda, double *At, int64_t ldat, void *stream) {
  PLS_REQUIRE(Vs && Kzx && A && At, "onb_build_projection: NULL pointer");
  PLS_REQUIRE(m > 0 && mk > 0 && n > 0 && mk <= m, "onb_build_projection: bad sizes");
  PLS_REQUIRE(ldvs >= mk && ldkzx >= n && lda >= n && ldat >= mk, "onb_build_projection: leading dimension too small");
  // A (mk x n) = Vs^T Kzx : L = Vs (m x mk), R = Kzx (m x n)
  int rc = pls_gemm_tn(Vs, ldvs, Kzx, ldkzx, A, lda, mk, n, m, 1.0, 0.0, stream);
  if (rc) return rc;
  // At (n x mk) = Kzx^T Vs : L = Kzx (m x n), R = Vs (m x mk)
  return pls_gemm_tn(Kzx, ldkzx, Vs, ldvs, At, ldat, n, mk, m, 1.0, 0.0, stream);
}

int pls_onb_build_gaussian(const pls_onb_desc *basis, const double *y, double *B, int64_t ldb, double *c, void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  PLS_REQUIRE(y && B && c && ldb >= basis->mk, "onb_build_gaussian: bad arguments");
  // B = A A^T = At^T At : L = R = At (n x mk)
  rc = pls_gemm_tn(basis->At, basis->ldat, basis->At, basis->ldat, B, ldb, basis->mk, basis->mk, basis->n, 1.0, 0.0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)basis->mk), dim3(256), 0, S(stream), basis->A, basis->lda,
                     basis->n, y, c);
  rc = check_launch("matvec_rows");
  if (rc) return rc;
  hipLaunchKernelGGL(matvec_rows_kernel, dim3(1), dim3(256), 0, S(stream), y, basis->n, basis->n, y, c + basis->mk);  // y^T y
  return check_launch("matvec_rows");
}

int pls_onb_forward(const pls_onb_desc *basis, const double *U, int64_t ldu, int64_t j, double *F, int64_t ldf,
                    void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  PLS_REQUIRE(U && F && j >= 0 && ldu >= j && ldf >= j, "onb_forward: bad arguments");
  return pls_gemm_tn(basis->A, basis->lda, U, ldu, F, ldf, basis->n, j, basis->mk, 1.0, 0.0, stream);
}

int pls_onb_particle_update(const pls_onb_desc *basis, const double *U, int64_t ldu, const double *G, int64_t ldg,
                            int64_t j, double eta, const pls_noise_desc *noise, double *dU, int64_t lddu, void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  rc = validate_noise(noise, basis->mk, j);
  if (rc) return rc;
  PLS_REQUIRE(U && G && dU && j >= 0 && ldu >= j && ldg >= j && lddu >= j, "onb_particle_update: bad arguments");
  PLS_REQUIRE(eta >= 0.0, "onb_particle_update: step size must be >= 0");
  if (j == 0) return PLS_OK;
  // D = A G into dU, then dU = -eta*D - eta*U/lam + sqrt(2 eta) xi in place (element-wise, race free)
  rc = pls_gemm_tn(basis->At, basis->ldat, G, ldg, dU, lddu, basis->mk, j, basis->n, 1.0, 0.0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(cdiv(basis->mk, 8) * 4)), dim3(256), 0,
                     S(stream), dU, lddu, U, ldu, dU, lddu, 1, (int64_t)0, U, ldu, basis->lam, 0.0, basis->mk, j,
                     make_etap(eta, nullptr), 0, make_noisep(noise));
  return check_launch("langevin_update");
}

static bool onb_fast_path(const pls_onb_desc *b, const pls_cost_desc *c, int force_generic) {
  return !force_generic && b->B && b->c && c->cost == PLS_COST_GAUSSIAN && c->link == PLS_LINK_IDENTITY;
}

static int64_t onb_max_slabs(int64_t mk, int64_t j, int64_t n) {
  int64_t kc;
  int64_t s = plan_split_k(mk, j, n, &kc);
  if (mk <= 256) {  // the fused small-rank kernels cut the rows into their own slabs (sized independently of the options)
    int64_t rows;
    const int64_t sr = small_rank_splits(j, n, &rows);
    if (sr > s) s = sr;
  }
  return s;
}

// partial rows of the step's energy by-product: one per 32 data rows of a chunk (the 64x64-tile worst case), or one per
// small-rank row slab (<= 32); sized for the chunk, not for N
static int64_t energy_partial_rows(int64_t n_chunk) { return cdiv(n_chunk, 32) < 32 ? 32 : cdiv(n_chunk, 32); }
static size_t onb_energy_partial_bytes(int64_t n_chunk, int64_t j) {
  return align_up((size_t)energy_partial_rows(n_chunk) * j * sizeof(double), 256);
}

// largest chunk (all rows, else a multiple of 128) whose G block and partial rows fit into `left` bytes
static int64_t onb_pick_chunk(int64_t n, size_t left, int64_t j, int64_t min_rows) {
  auto fits = [&](int64_t c) { return onb_energy_partial_bytes(c, j) + (size_t)c * j * sizeof(double) <= left; };
  if (fits(n)) return n;
  const double per_row = (double)j * sizeof(double) * (1.0 + 1.0 / 32.0);
  int64_t c = (int64_t)(((double)left - 32.0 * j * sizeof(double) - 512.0) / per_row);
  if (c > n) c = n;
  c = c / 128 * 128;
  while (c > min_rows && !fits(c)) c -= 128;
  while (c + 128 <= n && fits(c + 128)) c += 128;
  return c < min_rows ? min_rows : c;
}

// workspace of the one-launch small-rank step: the slabs' partial drifts and cost sums, then (callers that hand in no
// pls_block_desc.step_sync) the arrival counters
static size_t sr_step_workspace_bytes(int64_t mk, int64_t j, int64_t n) {
  if (mk < 1 || mk > 128) return 0;
  return align_up(small_rank_step_slab_bytes(j, n, (int)mk), 256) + align_up(small_rank_step_sync_words(j) * sizeof(uint32_t), 256);
}

size_t pls_onb_step_workspace_bytes(const pls_onb_desc *basis, int64_t j, int64_t n_chunk) {
  if (!basis || j <= 0) return 0;
  if (n_chunk <= 0 || n_chunk > basis->n) n_chunk = basis->n;
  // D slabs (split-K of the back-projection, each mk x j) + cost partial rows of the energy by-product + G chunk
  const size_t general = (size_t)onb_max_slabs(basis->mk, j, basis->n) * align_up((size_t)basis->mk * j * sizeof(double), 256) +
                         onb_energy_partial_bytes(n_chunk, j) + (size_t)n_chunk * j * sizeof(double);
  const size_t one_launch = sr_step_workspace_bytes(basis->mk, j, basis->n);
  return general > one_launch ? general : one_launch;
}

// The one-launch step (csrc/small_rank_step.h) applies to the orthonormal basis with <= 128 functions whose back-projection
// operand the LDS-DMA can stream (16-byte aligned rows); option 1 takes it while the problem is launch-bound -- few enough
// particle columns for one workgroup per 16 of them, and a step of at most 8 GFLOP (0.1 ms of matrix pipe): beyond, the
// slab kernels of small_rank.h share every tile of the operand between four column groups, which is what counts there.
static bool sr_step_route_for(const double *Lb, int64_t ldlb, int64_t mk, int64_t n, const double *y, int64_t j) {
  const int64_t mode = g_small_rank_step.load();
  if (mode == 0 || !small_rank_ok(Lb, ldlb, mk)) return false;
  if (reinterpret_cast<uintptr_t>(y) & 15) return false;  // (the targets travel by 16-byte LDS-DMA like the rows)
  if (mode >= 2) return true;
  return j <= 4096 && 4.0 * (double)n * (double)mk * (double)j <= 8e9;
}
static bool sr_step_route(const pls_onb_desc *b, const double *y, int64_t j) {
  return sr_step_route_for(b->At, b->ldat, b->mk, b->n, y, j);
}

// operands of the one-launch step for either basis: Lb (n x mk), the coordinates Vf the forward map contracts (and the prior
// term weighs: by 1 / lam, or by pconst when lam is NULL), the particles Uadd the update is added to (NULL: Vf)
struct SrStepOperands {
  const double *Lb;
  int64_t ldlb, mk, n;
  const double *Vf;
  int64_t ldvf;
  const double *Uadd;
  int64_t lduadd;
  const double *lam;
  double pconst;
  int64_t n_data = -1;  // rows of Lb that are data rows (the rest: prior rows, small_rank_step.h); -1: all n
};

// bytes the one-launch step takes from the workspace for j columns: its slabs, and its counters when the caller brings none
static size_t sr_step_need_bytes(int64_t mk, int64_t n, int64_t j, const pls_block_desc *blocks, const double *energy_in) {
  int64_t rows = 0;
  int64_t ns = small_rank_step_splits(j, n, (int)mk, &rows);
#ifdef PLS_SRS_PROBE
  if (const char *f = getenv("PLS_SRS_FORCE_NS")) {
    ns = atoi(f);
    rows = (cdiv(n, ns) + 63) / 64 * 64;
    ns = cdiv(n, rows);
  }
#endif
  const size_t slab_bytes = ns > 1 ? align_up((size_t)cdiv(j, 16) * ns * ((size_t)mk + 1) * 16 * sizeof(double), 256) : 0;
  const bool need_sync = ns > 1 || (blocks && energy_in && blocks->energy_sums);
  const bool own_sync = blocks && blocks->step_sync;
  return slab_bytes + ((need_sync && !own_sync) ? align_up(small_rank_step_sync_words(j) * sizeof(uint32_t), 256) : 0);
}
static bool sr_step_fits(int64_t mk, int64_t n, int64_t j, const pls_block_desc *blocks, const double *energy_in, size_t avail) {
  return sr_step_need_bytes(mk, n, j, blocks, energy_in) <= avail;
}

static int sr_step_launch(const SrStepOperands &basis_ops, const CostP &cp, const double *y, int64_t j,
                          const EtaP &etap, const NoiseP &nz, double *out, int64_t ldo, int out_mode, double *energy_in,
                          const pls_block_desc *blocks, void *workspace, size_t workspace_bytes, hipStream_t st, bool *taken) {
  const SrStepOperands *basis = &basis_ops;
  *taken = false;
  int64_t rows = 0;
  int64_t ns = small_rank_step_splits(j, basis->n, (int)basis->mk, &rows);
#ifdef PLS_SRS_PROBE
  if (const char *f = getenv("PLS_SRS_FORCE_NS")) {
    ns = atoi(f);
    rows = (cdiv(basis->n, ns) + 63) / 64 * 64;
    ns = cdiv(basis->n, rows);
  }
#endif
  double *esums = (blocks && energy_in) ? blocks->energy_sums : nullptr;
  const size_t slab_bytes = ns > 1 ? align_up((size_t)cdiv(j, 16) * ns * ((size_t)basis->mk + 1) * 16 * sizeof(double), 256) : 0;
  const bool need_sync = ns > 1 || esums != nullptr;
  uint32_t *sync = blocks ? blocks->step_sync : nullptr;
  const size_t sync_bytes = small_rank_step_sync_words(j) * sizeof(uint32_t);
  const size_t need = slab_bytes + ((need_sync && !sync) ? align_up(sync_bytes, 256) : 0);
  if (need > 0 && (!workspace || workspace_bytes < need)) return PLS_OK;  // (the general route states its own needs)
  if (need_sync && !sync) {  // counters from the workspace: zeroed by a memset node in front of the launch
    sync = reinterpret_cast<uint32_t *>(static_cast<char *>(workspace) + slab_bytes);
    hipError_t e = hipMemsetAsync(sync, 0, sync_bytes, st);
    if (e != hipSuccess) return fail(PLS_ERR_HIP, "onb_step: hipMemsetAsync: %s", hipGetErrorString(e));
  }
  const int64_t ncb = cdiv(j, 16);
  SrStepP p{};
  p.Lb = basis->Lb;
  p.ldlb = basis->ldlb;
  p.U = basis->Vf;
  p.ldu = basis->ldvf;
  p.Uadd = basis->Uadd;
  p.lduadd = basis->lduadd;
  p.y = y;
  p.lam = basis->lam;
  p.pconst = basis->pconst;
  p.N = basis->n;
  p.Ndata = basis->n_data >= 0 ? basis->n_data : basis->n;
  p.J = j;
  p.K = (int)basis->mk;
  p.rows_per_split = rows;
  p.nsplit = (int)ns;
  p.cp = cp;
  p.out = out;
  p.ldo = ldo;
  p.add_u = out_mode;
  p.etap = etap;
  p.nz = nz;
  p.cb_sync = sync;
  p.chunk_sync = sync ? sync + ncb : nullptr;
  p.slab = static_cast<double *>(workspace);
  p.vslab = p.slab ? p.slab + ncb * ns * basis->mk * 16 : nullptr;
  p.e = energy_in;
  p.esums = esums;
  p.sums16 = (blocks && energy_in) ? blocks->energy_sums16 : nullptr;
#ifdef PLS_SRS_PROBE
  p.debug_stop = getenv("PLS_SRS_STOP") ? atoi(getenv("PLS_SRS_STOP")) : 0;
#endif
  *taken = true;
  if (p.Ndata < p.N) return energy_in ? launch_small_rank_step_prior_value(p, st) : launch_small_rank_step_prior(p, st);
  return energy_in ? launch_small_rank_step_value(p, st) : launch_small_rank_step(p, st);
}

// pls_block_desc.energy_sums on the routes whose kernels do not leave them as a by-product (the generic N x M x J step, the
// inducing-point step outside whitened coordinates): one small launch over the finished per-particle energies, so that a
// caller who asked for the sums gets them whatever route the descriptor and the options select.
static int finish_energy_sums(const pls_block_desc *blocks, const double *energy_in, int64_t j, hipStream_t st) {
  if (!blocks || !energy_in) return PLS_OK;
  if (blocks->energy_sums) {
    hipLaunchKernelGGL(chunk_sums_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, st, energy_in, j, blocks->energy_sums);
    if (int rc = check_launch("chunk_sums")) return rc;
  }
  if (blocks->energy_sums16) {
    hipLaunchKernelGGL(sums16_kernel, dim3((unsigned)cdiv(cdiv(j, 16), 256)), dim3(256), 0, st, energy_in, j, blocks->energy_sums16);
    if (int rc = check_launch("sums16")) return rc;
  }
  return PLS_OK;
}

// pls_block_desc.energy_sums16 on the routes that finish energy_in themselves (the Gaussian/identity fast paths; not with lagged
// energies, whose values arrive one launch later)
static int finish_sums16(const pls_block_desc *blocks, const double *energy_in, int64_t j, hipStream_t st) {
  if (!blocks || !energy_in || !blocks->energy_sums16 || blocks->energy_partials) return PLS_OK;
  hipLaunchKernelGGL(sums16_kernel, dim3((unsigned)cdiv(cdiv(j, 16), 256)), dim3(256), 0, st, energy_in, j, blocks->energy_sums16);
  return check_launch("sums16");
}

static int validate_blocks(const pls_block_desc *b, int64_t j) {
  if (!b) return PLS_OK;
  PLS_REQUIRE(b->block_cols > 0 && b->eta != nullptr, "step_blocks: block_cols must be > 0 and eta set");
  (void)j;
  return PLS_OK;
}

static int onb_step_impl(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                         int64_t j, double eta, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out,
                         int64_t ldo, int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace,
                         size_t workspace_bytes, void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  rc = validate_noise(noise, basis->mk, j);
  if (rc) return rc;
  rc = validate_blocks(blocks, j);
  if (rc) return rc;
  rc = validate_lag(blocks);
  if (rc) return rc;
  if (blocks && blocks->energy_flush) {  // finish the last launch's partial rows; no step (U / ldu as in the step calls)
    PLS_REQUIRE(onb_fast_path(basis, cost, force_generic), "onb_step: energy_flush is for the Gaussian/identity fast path");
    PLS_REQUIRE(U && j > 0 && ldu >= j, "onb_step: energy_flush needs the particle matrix of the step calls");
    const FastOp op{basis->B, basis->ldb, basis->c, basis->lam, basis->mk, 1.0 / cost->p[0], 0.5 / cost->p[0], basis->c + basis->mk};
    return fast_step_launch(op, U, ldu, j, make_etap(eta, blocks), make_noisep(noise, blocks), nullptr, 0, 0, nullptr, nullptr, 0,
                            S(stream), "onb_step", nullptr, nullptr, make_lag(blocks));
  }
  PLS_REQUIRE(U && out && y, "onb_step: NULL pointer");
  PLS_REQUIRE(out != U, "onb_step: out must not alias U (ping-pong the particle buffers)");
  PLS_REQUIRE(j >= 0 && ldu >= j && ldo >= j, "onb_step: bad sizes");
  PLS_REQUIRE(eta >= 0.0, "onb_step: step size must be >= 0");
  PLS_REQUIRE(out_mode == 0 || out_mode == 1, "onb_step: out_mode must be 0 (delta) or 1 (new state)");
  if (j == 0) return PLS_OK;
  const CostP cp = make_costp(cost);
  const NoiseP nz = make_noisep(noise, blocks);
  const EtaP etap = make_etap(eta, blocks);
  hipStream_t st = S(stream);
  if (onb_fast_path(basis, cost, force_generic)) {
    const FastOp op{basis->B, basis->ldb, basis->c, basis->lam, basis->mk, 1.0 / cost->p[0], 0.5 / cost->p[0], basis->c + basis->mk};
    rc = fast_step_launch(op, U, ldu, j, etap, nz, out, ldo, out_mode, energy_in, workspace, workspace_bytes, st, "onb_step",
                          blocks ? blocks->energy_sums : nullptr, blocks ? blocks->energy_sync : nullptr, make_lag(blocks));
    return rc ? rc : finish_sums16(blocks, energy_in, j, st);
  }
  PLS_REQUIRE(!blocks || (!blocks->energy_partials && !blocks->energy_partials_prev),
              "onb_step: lagged energies (energy_partials) exist on the Gaussian/identity fast path only");
  if (sr_step_route(basis, y, j)) {  // launch-bound problems: the whole step, its energies and their chunk sums in ONE launch
    bool taken = false;
    const SrStepOperands ops{basis->At, basis->ldat, basis->mk, basis->n, U, ldu, nullptr, 0, basis->lam, 0.0};
    rc = sr_step_launch(ops, cp, y, j, etap, nz, out, ldo, out_mode, energy_in, blocks, workspace, workspace_bytes, st, &taken);
    if (rc || taken) return rc;
  }
  // workspace: [D slabs][cost partial rows (energy by-product)][G chunk]; the chunk length follows from what is left
  const size_t d_bytes = align_up((size_t)basis->mk * j * sizeof(double), 256);
  const int64_t max_slabs = onb_max_slabs(basis->mk, j, basis->n);
  const int64_t min_rows = basis->n < 128 ? basis->n : 128;
  const size_t need_min = max_slabs * d_bytes + onb_energy_partial_bytes(min_rows, j) + (size_t)min_rows * j * sizeof(double);
  if (!workspace || workspace_bytes < need_min)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "onb_step: workspace %zu bytes, need at least %zu", workspace_bytes,
                pls_onb_step_workspace_bytes(basis, j, 128));
  double *D = static_cast<double *>(workspace);
  const size_t left = workspace_bytes - max_slabs * d_bytes;
  const int64_t n_chunk = onb_pick_chunk(basis->n, left, j, min_rows);
  double *vpart = reinterpret_cast<double *>(static_cast<char *>(workspace) + max_slabs * d_bytes);
  double *Gbuf = reinterpret_cast<double *>(static_cast<char *>(workspace) + max_slabs * d_bytes +
                                            onb_energy_partial_bytes(n_chunk, j));
  EnergySink sink;
  if (energy_in) {  // e_j = cost_j(F(U)) + 1/2 sum_m U_mj^2 / lam_m of the INPUT particles (orthonormal.py:120-125)
    sink.partial = vpart;
    sink.rows_cap = energy_partial_rows(n_chunk);
    sink.e = energy_in;
    sink.prior_kind = 1;
    sink.P = U;
    sink.ldp = ldu;
    sink.m = basis->mk;
    sink.lam = basis->lam;
  }
  int64_t nslab = 1;
  rc = stream_drift(basis->A, basis->lda, basis->At, basis->ldat, basis->mk, basis->n, U, ldu, j, cp, y, D, j, max_slabs,
                    (int64_t)(d_bytes / sizeof(double)), &nslab, Gbuf, n_chunk, st, energy_in ? &sink : nullptr);
  if (rc) return rc;
  {
    LaunchScope scope(PLS_TAG_LANGEVIN_UPDATE, st);
    hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(cdiv(basis->mk, 8) * 4)), dim3(256), 0,
                       st, out, ldo, U, ldu, D, j, (int)nslab, (int64_t)(d_bytes / sizeof(double)), U, ldu, basis->lam, 0.0,
                       basis->mk, j, etap, out_mode, nz);
  }
  rc = check_launch("langevin_update");
  if (rc) return rc;
  return finish_energy_sums(blocks, energy_in, j, st);
}

int pls_onb_step(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                 int64_t j, double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode,
                 int32_t force_generic, double *energy_in, void *workspace, size_t workspace_bytes, void *stream) {
  return onb_step_impl(basis, cost, y, U, ldu, j, eta, nullptr, noise, out, ldo, out_mode, force_generic, energy_in, workspace,
                       workspace_bytes, stream);
}

int pls_onb_step_blocks(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                        int64_t j, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                        int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace,
                        size_t workspace_bytes, void *stream) {
  PLS_REQUIRE(blocks != nullptr, "onb_step_blocks: block descriptor is NULL");
  return onb_step_impl(basis, cost, y, U, ldu, j, 0.0, blocks, noise, out, ldo, out_mode, force_generic, energy_in, workspace,
                       workspace_bytes, stream);
}

size_t pls_onb_energy_workspace_bytes(const pls_onb_desc *basis, int64_t j, int64_t n_chunk) {
  if (!basis || j <= 0) return 0;
  if (n_chunk <= 0 || n_chunk > basis->n) n_chunk = basis->n;
  int64_t parts = cdiv(n_chunk, 64) < 2 ? 2 : cdiv(n_chunk, 64);  // (the chunk planner needs two partial rows)
  if (cdiv(basis->mk, 64) > parts) parts = cdiv(basis->mk, 64);  // the Gaussian quadratic form reduces over M_k instead
  return (size_t)parts * j * sizeof(double);
}

int pls_onb_energy(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, const double *U, int64_t ldu,
                   int64_t j, double *e, int32_t force_generic, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  PLS_REQUIRE(U && e && y && j >= 0 && ldu >= j, "onb_energy: bad arguments");
  if (j == 0) return PLS_OK;
  if (onb_fast_path(basis, cost, force_generic)) {
    // cost_j = (u^T B u - 2 c^T u + y^T y) / (2 sigma2): one Mk x Mk x J contraction, reduced per tile then per column
    const FastOp op{basis->B, basis->ldb, basis->c, basis->lam, basis->mk, 1.0 / cost->p[0], 0.5 / cost->p[0], basis->c + basis->mk};
    return fast_energy_launch(op, U, ldu, j, e, workspace, workspace_bytes, S(stream), "onb_energy");
  }
  // rows per chunk such that the partial buffer ((chunk/64) x j doubles) fits
  const int64_t max_parts = (int64_t)(workspace_bytes / ((size_t)j * sizeof(double)));
  if (!workspace || max_parts < 2)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "onb_energy: workspace %zu bytes too small", workspace_bytes);
  int64_t n_chunk = max_parts * 64;
  if (n_chunk > basis->n) n_chunk = basis->n;
  else n_chunk = n_chunk / 128 * 128;
  return stream_cost(basis->A, basis->lda, basis->At, basis->ldat, basis->mk, basis->n, U, ldu, j, make_costp(cost), y,
                     static_cast<double *>(workspace), max_parts, n_chunk, e, 1, U, ldu, basis->mk, basis->lam, 0.0,
                     S(stream));
}

int pls_onb_prior_energy(const pls_onb_desc *basis, const double *U, int64_t ldu, int64_t j, const double *cost,
                         double *e, void *stream) {
  int rc = validate_onb(basis);
  if (rc) return rc;
  PLS_REQUIRE(U && e && j >= 0 && ldu >= j, "onb_prior_energy: bad arguments");
  if (j == 0) return PLS_OK;
  hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, S(stream), cost, j,
                     (int64_t)(cost ? 1 : 0), j, e, 0, 1, U, ldu, basis->mk, basis->lam, 0.0, 1.0, (const double *)nullptr);
  return check_launch("column_reduce");
}

// ---- inducing-point basis ---------------------------------------------------------------------------------------

static int validate_ipb(const pls_ipb_desc *b) {
  PLS_REQUIRE(b != nullptr, "ipb descriptor is NULL");
  PLS_REQUIRE(b->m > 0 && b->n > 0, "ipb: m and n must be positive");
  PLS_REQUIRE(b->Kzx && b->Kxz, "ipb: Kzx and Kxz must be set");
  PLS_REQUIRE((b->Sf && b->Sb) || b->W || (b->Linv && b->LinvT),
              "ipb: k(Z,Z) must enter as substitution operators Sf / Sb (pls_chol_factor), as the inverse factor Linv / LinvT or as W");
  PLS_REQUIRE((b->Linv == nullptr) == (b->LinvT == nullptr), "ipb: Linv and LinvT go together");
  if (b->Linv) PLS_REQUIRE(b->ldlinv >= b->m && b->ldlinvt >= b->m, "ipb: ldlinv / ldlinvt < m");
  if (b->Q) PLS_REQUIRE(b->ldq >= b->m && b->ct, "ipb: Q needs ldq >= m and ct");
  PLS_REQUIRE((b->Sf == nullptr) == (b->Sb == nullptr), "ipb: Sf and Sb go together");
  PLS_REQUIRE(b->ldkzx >= b->n && b->ldkxz >= b->m, "ipb: leading dimension too small");
  if (b->W) PLS_REQUIRE(b->ldw >= b->m, "ipb: ldw < m");
  if (b->Sf) PLS_REQUIRE(b->ldsf >= b->m && b->ldsb >= b->m, "ipb: ldsf / ldsb < m");
  if (b->LcT) PLS_REQUIRE(b->ldlct >= b->m, "ipb: ldlct < m");
  if (b->B) PLS_REQUIRE(b->ldb >= b->m && b->c, "ipb: fast-path constants need ldb >= m and c");
  return PLS_OK;
}

static pls_chol_desc ipb_factor(const pls_ipb_desc *b) {
  return pls_chol_desc{b->m, nullptr, 0, b->LcT, b->ldlct, b->Sf, b->ldsf, b->Sb, b->ldsb, b->Linv, b->ldlinv, b->LinvT, b->ldlinvt,
                       b->tri_scratch, b->tri_scratch_bytes};
}

// V (m x j, ld j) = k(Z,Z)^-1 U = Lc^-T Lc^-1 U.  Two triangular products with the inverse factor when the descriptor
// carries it and `tmp` (m x j doubles) is there to hold Lc^-1 U; else forward + backward block substitution in one launch;
// or -- A/B option, or a descriptor without a factor -- the contraction with the explicit inverse W.
static int ipb_apply_kinv(const pls_ipb_desc *b, const double *U, int64_t ldu, int64_t j, double *V, void *stream,
                          double *tmp = nullptr) {
  const bool explicit_inverse = (!b->Sf && !b->Linv) || (g_ipb_explicit_inverse.load() != 0 && b->W);
  if (explicit_inverse) return pls_gemm_tn(b->W, b->ldw, U, ldu, V, j, b->m, j, b->m, 1.0, 0.0, stream);  // W symmetric
  const pls_chol_desc f = ipb_factor(b);
  return chol_full_solve(&f, U, ldu, j, V, j, tmp, S(stream));
}

// ---- the Gaussian/identity step in whitened coordinates ----------------------------------------------------------------
// With S = Lc^-1 U (k(Z,Z) = Lc Lc^T) the update of inducing_point.py:117-150 under gaussian.py:86-88,
//     dU = -eta ((B V - c) / sigma2 + M V) + sqrt(2 eta) Lc xi,   V = k(Z,Z)^-1 U = Lc^-T S,
// reads  dU = Lc dS,  dS = -eta (Q S - c~) + sqrt(2 eta) xi  with  Q = Lc^-1 (B / sigma2 + M I) Lc^-T,  c~ = Lc^-1 c / sigma2
// (built once per sigma2 by pls_ipb_build_whitened), and the energy of inducing_point.py:95-115 is
// S^T Q S / 2 - c~^T S + y^T y / (2 sigma2).  dS is exactly the orthonormal basis' fused fast-path kernel with operator
// (Q, c~) and no prior term -- the Philox noise of that kernel is the xi of the reference's e = Lc xi.  A step is then
// forward solve (M^2 J flop) + Q S (2 M^2 J) + Lc dS (M^2 J) in three launches, against solve (2 M^2 J) + B V (2 M^2 J) +
// Lc xi (M^2 J) + noise + update in six; a loop that keeps S between steps pays 2 M^2 J per step.
static bool ipb_whitened_ok(const pls_ipb_desc *b, const pls_cost_desc *c) {
  return b->Q && b->ct && b->LcT && b->ldq >= b->m && b->q_inv_noise == 1.0 / c->p[0] && (b->LinvT || (b->Sf && b->Sb)) &&
         g_ipb_explicit_inverse.load() == 0;
}

static FastOp ipb_whitened_op(const pls_ipb_desc *b) {
  return FastOp{b->Q, b->ldq, b->ct, nullptr, b->m, 1.0, 0.5 * b->q_inv_noise, b->ct + b->m};
}

int pls_ipb_forward(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, double *F, int64_t ldf,
                    void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(U && F && j >= 0 && ldu >= j && ldf >= j, "ipb_forward: bad arguments");
  if (j == 0) return PLS_OK;
  const size_t need = (size_t)basis->m * j * sizeof(double);
  if (!workspace || workspace_bytes < need)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_forward: workspace %zu < %zu bytes", workspace_bytes, need);
  double *V = static_cast<double *>(workspace);
  rc = ipb_apply_kinv(basis, U, ldu, j, V, stream);
  if (rc) return rc;
  return pls_gemm_tn(basis->Kzx, basis->ldkzx, V, j, F, ldf, basis->n, j, basis->m, 1.0, 0.0, stream);
}

// shared tail of the IPB update: out = [U +] -eta*D - eta*M*V + sqrt(2 eta) e
// e = Lc xi for the library's own noise: Philox normals into xi_buf, a triangular product into e_buf; `nz` then names e_buf as
// injected noise (the inducing-point update adds N(0, k(Z,Z)) samples: inducing_point.py:133-137)
static int ipb_colour_noise(const pls_ipb_desc *basis, NoiseP &nz, int64_t j, double *xi_buf, double *e_buf, hipStream_t st) {
  if (nz.kind != PLS_NOISE_PHILOX) return PLS_OK;
  if (!basis->LcT) return fail(PLS_ERR_INVALID_ARGUMENT, "ipb: Philox noise needs the Cholesky factor LcT");
  hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(cdiv(basis->m, 8) * 4)), dim3(256), 0, st,
                     xi_buf, j, basis->m, j, nz.seed, nz.step, nz.j_offset, nz.step_base, nz.block_cols);
  int rc = check_launch("normal_fill");
  if (rc) return rc;
  // e = Lc xi :  L[k][i] = LcT[k][i] = Lc[i][k], zero for k > i: a triangular product (half the contraction)
  rc = gemm_tn_ex(basis->LcT, basis->ldlct, xi_buf, j, e_buf, j, basis->m, j, basis->m, 1.0, 0.0, 1, st, basis->tri_scratch,
                  basis->tri_scratch_bytes);
  if (rc) return rc;
  nz.kind = PLS_NOISE_INJECTED;
  nz.xi = e_buf;
  nz.ldxi = j;
  return PLS_OK;
}

static int ipb_finish(const pls_ipb_desc *basis, const double *U, int64_t ldu, const double *D, int nslab,
                      int64_t slab_stride, const double *V, int64_t j, double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int add_u,
                      double *xi_buf, double *e_buf, hipStream_t st, const double *dsub = nullptr, double dsub_scale = 0.0,
                      const pls_block_desc *blocks = nullptr) {
  NoiseP nz = make_noisep(noise, blocks);
  if (int rc = ipb_colour_noise(basis, nz, j, xi_buf, e_buf, st)) return rc;
  hipLaunchKernelGGL(langevin_update_kernel, dim3((unsigned)cdiv(j, 256), rows_grid(cdiv(basis->m, 8) * 4)), dim3(256), 0, st,
                     out, ldo, U, ldu, D, j, nslab, slab_stride, V, j, (const double *)nullptr, (double)basis->m, basis->m, j,
                     make_etap(eta, blocks), add_u, nz, dsub, dsub_scale);
  return check_launch("langevin_update");
}

static bool ipb_fast_path(const pls_ipb_desc *b, const pls_cost_desc *c, int force_generic) {
  return !force_generic && b->B && b->c && c->cost == PLS_COST_GAUSSIAN && c->link == PLS_LINK_IDENTITY;
}

// Gaussian/identity constants of the inducing-point basis: B = Kzx Kxz (M x M), c[0..M) = Kzx y, c[M] = y^T y.
// With V = K^-1 U the data drift is Kzx (Kxz V - y) / sigma2 = (B V - c) / sigma2: two M x M x J products per step
// instead of two N x M x J ones (the same algebra as pls_onb_build_gaussian).
int pls_ipb_build_gaussian(const pls_ipb_desc *basis, const double *y, double *B, int64_t ldb, double *c, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(y && B && c && ldb >= basis->m, "ipb_build_gaussian: bad arguments");
  rc = pls_gemm_tn(basis->Kxz, basis->ldkxz, basis->Kxz, basis->ldkxz, B, ldb, basis->m, basis->m, basis->n, 1.0, 0.0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)basis->m), dim3(256), 0, S(stream), basis->Kzx, basis->ldkzx,
                     basis->n, y, c);
  rc = check_launch("matvec_rows");
  if (rc) return rc;
  hipLaunchKernelGGL(matvec_rows_kernel, dim3(1), dim3(256), 0, S(stream), y, basis->n, basis->n, y, c + basis->m);  // y^T y
  return check_launch("matvec_rows");
}

int pls_ipb_particle_update(const pls_ipb_desc *basis, const double *U, int64_t ldu, const double *G, int64_t ldg,
                            int64_t j, double eta, const pls_noise_desc *noise, double *dU, int64_t lddu,
                            void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_noise(noise, basis->m, j);
  if (rc) return rc;
  PLS_REQUIRE(U && G && dU && j >= 0 && ldu >= j && ldg >= j && lddu >= j, "ipb_particle_update: bad arguments");
  PLS_REQUIRE(eta >= 0.0, "ipb_particle_update: step size must be >= 0");
  if (j == 0) return PLS_OK;
  const size_t mj = align_up((size_t)basis->m * j * sizeof(double), 256);
  if (!workspace || workspace_bytes < 4 * mj)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_particle_update: workspace %zu < %zu bytes", workspace_bytes, 4 * mj);
  char *w = static_cast<char *>(workspace);
  double *V = reinterpret_cast<double *>(w), *D = reinterpret_cast<double *>(w + mj);
  double *xi = reinterpret_cast<double *>(w + 2 * mj), *e = reinterpret_cast<double *>(w + 3 * mj);
  rc = ipb_apply_kinv(basis, U, ldu, j, V, stream, xi);  // (xi is free until the noise is drawn)
  if (rc) return rc;
  rc = pls_gemm_tn(basis->Kxz, basis->ldkxz, G, ldg, D, j, basis->m, j, basis->n, 1.0, 0.0, stream);
  if (rc) return rc;
  return ipb_finish(basis, U, ldu, D, 1, (int64_t)0, V, j, eta, noise, dU, lddu, 0, xi, e, S(stream));
}

size_t pls_ipb_step_workspace_bytes(const pls_ipb_desc *basis, int64_t j, int64_t n_chunk) {
  if (!basis || j <= 0) return 0;
  if (n_chunk <= 0 || n_chunk > basis->n) n_chunk = basis->n;
  // V, xi, e (m x j each) + D slabs + cost partial rows of the energy by-product + G chunk
  const size_t mj = align_up((size_t)basis->m * j * sizeof(double), 256);
  const size_t general = (size_t)(3 + onb_max_slabs(basis->m, j, basis->n)) * mj + onb_energy_partial_bytes(n_chunk, j) +
                         (size_t)n_chunk * j * sizeof(double);
  const size_t one_launch = 3 * mj + sr_step_workspace_bytes(basis->m, j, basis->n);  // (V, xi, e, then the step's slabs)
  return general > one_launch ? general : one_launch;
}

static int ipb_step_impl(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu, int64_t j,
                         double eta, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                         int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace, size_t workspace_bytes,
                         void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  rc = validate_noise(noise, basis->m, j);
  if (rc) return rc;
  rc = validate_blocks(blocks, j);
  if (rc) return rc;
  PLS_REQUIRE(!blocks || (!blocks->energy_partials && !blocks->energy_partials_prev && !blocks->energy_flush),
              "ipb_step: lagged energies (energy_partials) exist on pls_onb_step_blocks and pls_ipb_whitened_step_blocks only");
  PLS_REQUIRE(U && out && y, "ipb_step: NULL pointer");
  PLS_REQUIRE(out != U, "ipb_step: out must not alias U");
  PLS_REQUIRE(j >= 0 && ldu >= j && ldo >= j && eta >= 0.0, "ipb_step: bad sizes");
  PLS_REQUIRE(out_mode == 0 || out_mode == 1, "ipb_step: out_mode must be 0 or 1");
  if (j == 0) return PLS_OK;
  const size_t mj = align_up((size_t)basis->m * j * sizeof(double), 256);
  const int64_t max_slabs = onb_max_slabs(basis->m, j, basis->n);
  const size_t fixed = (size_t)(3 + max_slabs) * mj;
  const int64_t min_rows = basis->n < 128 ? basis->n : 128;
  if (!workspace ||
      workspace_bytes < fixed + onb_energy_partial_bytes(min_rows, j) + (size_t)min_rows * j * sizeof(double))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_step: workspace %zu bytes, need at least %zu", workspace_bytes,
                pls_ipb_step_workspace_bytes(basis, j, 128));
  char *w = static_cast<char *>(workspace);
  double *V = reinterpret_cast<double *>(w), *xi = reinterpret_cast<double *>(w + mj);
  double *e = reinterpret_cast<double *>(w + 2 * mj), *D = reinterpret_cast<double *>(w + 3 * mj);
  const int64_t n_chunk = onb_pick_chunk(basis->n, workspace_bytes - fixed, j, min_rows);
  double *vpart = reinterpret_cast<double *>(w + fixed);
  double *Gbuf = reinterpret_cast<double *>(w + fixed + onb_energy_partial_bytes(n_chunk, j));
  hipStream_t st = S(stream);
  if (ipb_fast_path(basis, cost, force_generic) && ipb_whitened_ok(basis, cost)) {
    // whitened route: S = Lc^-1 U -> dS (fused kernel, energy by-product) -> out = [U +] Lc dS
    double *Sw = V, *Wd = xi, *epart = e;  // (m x j each; the partial rows of the energy fit: 2 cdiv(m, 128) <= m + 1)
    NoiseP nz = make_noisep(noise, blocks);
    const bool injected = nz.kind == PLS_NOISE_INJECTED;
    const double *e_inj = injected ? nz.xi : nullptr;
    const int64_t ld_inj = nz.ldxi;
    if (injected) nz.kind = PLS_NOISE_NONE;  // (the injected noise is already coloured: it enters after the product with Lc)
    if (basis->Pt && !energy_in && g_ipb_step_operator.load() != 0) {
      // dS straight from U: Q S = Q Lc^-1 U = P U with P^T = Lc^-T Q in the descriptor (pls_ipb_build_step_operator), so
      // the forward solve is folded into the operator -- two launches and 3 M^2 J flop per call instead of three and 4 M^2 J.
      // (The energy by-product is a quadratic form in S, so a call that wants it keeps the route below.)
      FastOp op = ipb_whitened_op(basis);
      op.B = basis->Pt;
      op.ldb = basis->ldpt;
      rc = fast_step_launch(op, U, ldu, j, make_etap(eta, blocks), nz, Wd, j, 0, nullptr, nullptr, 0, st, "ipb_step");
    } else {
      const pls_chol_desc f = ipb_factor(basis);
      rc = chol_forward_solve(&f, U, ldu, j, Sw, j, st);
      if (rc) return rc;
      rc = fast_step_launch(ipb_whitened_op(basis), Sw, j, j, make_etap(eta, blocks), nz, Wd, j, 0, energy_in, epart,
                            2 * mj, st, "ipb_step", blocks ? blocks->energy_sums : nullptr, blocks ? blocks->energy_sync : nullptr);
    }
    if (rc) return rc;
    rc = finish_sums16(blocks, energy_in, j, st);
    if (rc) return rc;
    EpiIpbFinish fin{out, ldo, U, ldu, out_mode, make_etap(eta, blocks), e_inj, ld_inj};
    return launch_gemm_any(basis->LcT, basis->ldlct, Wd, j, basis->m, j, basis->m, fin, st, 0, 1,
                           TriScratch{basis->tri_scratch, basis->tri_scratch_bytes});
  }
  // launch-bound problems (the reference's curve experiments build this basis with 10-100 inducing points and 50-100
  // particles): the one-launch small-rank step after the solve and the coloured noise -- and those two in one launch of
  // their own when the descriptor carries the inverse factor (csrc/ipb_prep.h): 2 launches per step instead of 8
  const bool fast = ipb_fast_path(basis, cost, force_generic);
  const bool one_launch = !fast && sr_step_route_for(basis->Kxz, basis->ldkxz, basis->m, basis->n, y, j) &&
                          sr_step_fits(basis->m, basis->n, j, blocks, energy_in, workspace_bytes - 3 * mj);
  bool coloured = false;  // the noise of this step already sits in e
  if (one_launch && g_ipb_prep.load() != 0 && basis->m <= IPB_PREP_MAX_M && basis->Linv && basis->LinvT && solve_mode() != 0 &&
      !(g_ipb_explicit_inverse.load() != 0 && basis->W)) {
    const NoiseP nz0 = make_noisep(noise, blocks);
    const int draw = nz0.kind == PLS_NOISE_PHILOX ? 1 : 0;
    if (draw && !basis->LcT) return fail(PLS_ERR_INVALID_ARGUMENT, "ipb: Philox noise needs the Cholesky factor LcT");
    const IpbPrepP pp{basis->LinvT, basis->ldlinvt, basis->Linv, basis->ldlinv, basis->LcT, basis->ldlct, U, ldu, V, j, e, j,
                      (int)basis->m, j, draw, nz0};
    rc = launch_ipb_prep(pp, st);
    coloured = draw != 0;
  } else {
    rc = ipb_apply_kinv(basis, U, ldu, j, V, stream, xi);  // (xi is free until the noise is drawn)
  }
  if (rc) return rc;
  if (ipb_fast_path(basis, cost, force_generic)) {
    const double inv_noise = 1.0 / cost->p[0];
    rc = pls_gemm_tn(basis->B, basis->ldb, V, j, D, j, basis->m, j, basis->m, inv_noise, 0.0, stream);  // B symmetric
    if (rc) return rc;
    if (energy_in) {
      hipLaunchKernelGGL(ipb_gaussian_energy_kernel, dim3((unsigned)cdiv(j, 64)), dim3(256), 0, st, V, j, D, j, basis->c,
                         basis->m, j, inv_noise, 0.5 * (double)basis->m, energy_in);
      rc = check_launch("ipb_gaussian_energy");
      if (rc) return rc;
    }
    rc = ipb_finish(basis, U, ldu, D, 1, (int64_t)0, V, j, eta, noise, out, ldo, out_mode, xi, e, st, basis->c, inv_noise,
                    blocks);
    if (rc) return rc;
    return finish_energy_sums(blocks, energy_in, j, st);
  }
  if (one_launch) {
    NoiseP nz = make_noisep(noise, blocks);
    if (coloured) {
      nz.kind = PLS_NOISE_INJECTED;
      nz.xi = e;
      nz.ldxi = j;
    } else {
      rc = ipb_colour_noise(basis, nz, j, xi, e, st);
      if (rc) return rc;
    }
    bool taken = false;
    const SrStepOperands ops{basis->Kxz, basis->ldkxz, basis->m, basis->n, V, j, U, ldu, nullptr, (double)basis->m};
    rc = sr_step_launch(ops, make_costp(cost), y, j, make_etap(eta, blocks), nz, out, ldo, out_mode, energy_in, blocks, D,
                        workspace_bytes - 3 * mj, st, &taken);
    if (rc) return rc;
    return taken ? PLS_OK : fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_step: the one-launch step refused a workspace it was sized for");
  }
  EnergySink sink;
  if (energy_in) {  // e_j = cost_j(F(U)) + (M/2) ||K^-1 U_j||^2 of the INPUT particles (inducing_point.py:95-115)
    sink.partial = vpart;
    sink.rows_cap = energy_partial_rows(n_chunk);
    sink.e = energy_in;
    sink.prior_kind = 2;
    sink.P = V;
    sink.ldp = j;
    sink.m = basis->m;
    sink.scale = 0.5 * (double)basis->m;
  }
  int64_t nslab = 1;
  rc = stream_drift(basis->Kzx, basis->ldkzx, basis->Kxz, basis->ldkxz, basis->m, basis->n, V, j, j, make_costp(cost), y,
                    D, j, max_slabs, (int64_t)(mj / sizeof(double)), &nslab, Gbuf, n_chunk, st, energy_in ? &sink : nullptr);
  if (rc) return rc;
  rc = ipb_finish(basis, U, ldu, D, (int)nslab, (int64_t)(mj / sizeof(double)), V, j, eta, noise, out, ldo, out_mode, xi, e,
                  st, nullptr, 0.0, blocks);
  if (rc) return rc;
  return finish_energy_sums(blocks, energy_in, j, st);
}

int pls_ipb_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu, int64_t j,
                 double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode, int32_t force_generic,
                 double *energy_in, void *workspace, size_t workspace_bytes, void *stream) {
  return ipb_step_impl(basis, cost, y, U, ldu, j, eta, nullptr, noise, out, ldo, out_mode, force_generic, energy_in, workspace,
                       workspace_bytes, stream);
}

int pls_ipb_step_blocks(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                        int64_t j, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                        int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace,
                        size_t workspace_bytes, void *stream) {
  PLS_REQUIRE(blocks != nullptr, "ipb_step_blocks: block descriptor is NULL");
  return ipb_step_impl(basis, cost, y, U, ldu, j, 0.0, blocks, noise, out, ldo, out_mode, force_generic, energy_in, workspace,
                       workspace_bytes, stream);
}

size_t pls_ipb_energy_workspace_bytes(const pls_ipb_desc *basis, int64_t j, int64_t n_chunk) {
  if (!basis || j <= 0) return 0;
  if (n_chunk <= 0 || n_chunk > basis->n) n_chunk = basis->n;
  const int64_t parts = cdiv(n_chunk, 64) < 2 ? 2 : cdiv(n_chunk, 64);
  const size_t mj = align_up((size_t)basis->m * j * sizeof(double), 256);
  const size_t generic = mj + (size_t)parts * j * sizeof(double);
  return generic > 2 * mj ? generic : 2 * mj;  // (the Gaussian fast path keeps V and B V)
}

int pls_ipb_energy(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, const double *U, int64_t ldu,
                   int64_t j, double *e, int32_t force_generic, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  PLS_REQUIRE(U && e && y && j >= 0 && ldu >= j, "ipb_energy: bad arguments");
  if (j == 0) return PLS_OK;
  const size_t mj = align_up((size_t)basis->m * j * sizeof(double), 256);
  if (!workspace || workspace_bytes < mj + 2 * (size_t)j * sizeof(double))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_energy: workspace %zu bytes too small", workspace_bytes);
  double *V = static_cast<double *>(workspace);
  if (ipb_fast_path(basis, cost, force_generic)) {
    if (workspace_bytes < 2 * mj) return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_energy: workspace %zu < %zu bytes", workspace_bytes, 2 * mj);
    double *D = reinterpret_cast<double *>(static_cast<char *>(workspace) + mj);
    const double inv_noise = 1.0 / cost->p[0];
    if (ipb_whitened_ok(basis, cost)) {  // S = Lc^-1 U, then the quadratic form in whitened coordinates
      const pls_chol_desc f = ipb_factor(basis);
      rc = chol_forward_solve(&f, U, ldu, j, V, j, S(stream));
      if (rc) return rc;
      return fast_energy_launch(ipb_whitened_op(basis), V, j, j, e, D, workspace_bytes - mj, S(stream), "ipb_energy");
    }
    rc = ipb_apply_kinv(basis, U, ldu, j, V, stream, D);
    if (rc) return rc;
    rc = pls_gemm_tn(basis->B, basis->ldb, V, j, D, j, basis->m, j, basis->m, inv_noise, 0.0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(ipb_gaussian_energy_kernel, dim3((unsigned)cdiv(j, 64)), dim3(256), 0, S(stream), V, j, D, j, basis->c,
                       basis->m, j, inv_noise, 0.5 * (double)basis->m, e);
    return check_launch("ipb_gaussian_energy");
  }
  double *partial = reinterpret_cast<double *>(static_cast<char *>(workspace) + mj);
  const int64_t max_parts = (int64_t)((workspace_bytes - mj) / ((size_t)j * sizeof(double)));
  int64_t n_chunk = max_parts * 64;
  if (n_chunk > basis->n) n_chunk = basis->n;
  else n_chunk = n_chunk / 128 * 128;
  rc = ipb_apply_kinv(basis, U, ldu, j, V, stream);
  if (rc) return rc;
  return stream_cost(basis->Kzx, basis->ldkzx, basis->Kxz, basis->ldkxz, basis->m, basis->n, V, j, j, make_costp(cost), y,
                     partial, max_parts, n_chunk, e, 2, V, j, basis->m, nullptr, 0.5 * (double)basis->m, S(stream));
}

int pls_ipb_prior_energy(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, const double *cost,
                         double *e, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(U && e && j >= 0 && ldu >= j, "ipb_prior_energy: bad arguments");
  if (j == 0) return PLS_OK;
  const size_t need = (size_t)basis->m * j * sizeof(double);
  if (!workspace || workspace_bytes < need)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_prior_energy: workspace %zu < %zu bytes", workspace_bytes, need);
  double *V = static_cast<double *>(workspace);
  rc = ipb_apply_kinv(basis, U, ldu, j, V, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(column_reduce_kernel, dim3((unsigned)cdiv(j, 256)), dim3(256), 0, S(stream), cost, j,
                     (int64_t)(cost ? 1 : 0), j, e, 0, 2, (const double *)V, j, basis->m, (const double *)nullptr,
                     0.5 * (double)basis->m, 1.0, (const double *)nullptr);
  return check_launch("column_reduce");
}

// ---- whitened coordinates of the inducing-point basis (see ipb_whitened_ok above) ------------------------------------
size_t pls_ipb_build_whitened_workspace_bytes(int64_t m) {
  if (m <= 0) return 0;
  const int64_t ldt = (m + 1) & ~(int64_t)1;
  return 2 * align_up((size_t)m * ldt * sizeof(double), 256);
}

int pls_ipb_build_whitened(const pls_ipb_desc *basis, double inv_noise, double *Q, int64_t ldq, double *ct, void *workspace,
                           size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(basis->B && basis->c, "ipb_build_whitened: the descriptor needs B and c (pls_ipb_build_gaussian)");
  PLS_REQUIRE(basis->LinvT || (basis->Sf && basis->Sb), "ipb_build_whitened: the descriptor needs the Cholesky factor's operators");
  PLS_REQUIRE(Q && ct && ldq >= basis->m && inv_noise > 0.0, "ipb_build_whitened: bad arguments");
  PLS_REQUIRE((ldq & 1) == 0 && (reinterpret_cast<uintptr_t>(Q) & 15) == 0, "ipb_build_whitened: Q must be 16-byte aligned, ldq even");
  const int64_t m = basis->m, ldt = (m + 1) & ~(int64_t)1;
  if (!workspace || workspace_bytes < pls_ipb_build_whitened_workspace_bytes(m))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_build_whitened: workspace %zu < %zu bytes", workspace_bytes,
                pls_ipb_build_whitened_workspace_bytes(m));
  hipStream_t st = S(stream);
  double *T1 = static_cast<double *>(workspace);
  double *T2 = reinterpret_cast<double *>(static_cast<char *>(workspace) + align_up((size_t)m * ldt * sizeof(double), 256));
  const pls_chol_desc f = ipb_factor(basis);
  // Q = Lc^-1 B' Lc^-T with B' = B / sigma2 + M I:  T2 = Lc^-1 B',  Q = Lc^-1 T2^T (B' symmetric)
  rc = launch_scale_add_diag(basis->B, basis->ldb, inv_noise, (double)m, T1, ldt, m, st);
  if (rc) return rc;
  rc = chol_forward_solve(&f, T1, ldt, m, T2, ldt, st);
  if (rc) return rc;
  rc = launch_transpose(T2, ldt, T1, ldt, m, m, st);
  if (rc) return rc;
  rc = chol_forward_solve(&f, T1, ldt, m, Q, ldq, st);
  if (rc) return rc;
  // c~ = Lc^-1 c / sigma2, then y^T y
  hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, st, basis->c, inv_noise, T1, m);
  rc = check_launch("scale_copy");
  if (rc) return rc;
  rc = chol_forward_solve(&f, T1, 1, 1, ct, 1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scale_copy_kernel, dim3(1), dim3(256), 0, st, basis->c + m, 1.0, ct + m, (int64_t)1);
  return check_launch("scale_copy");
}

int pls_ipb_build_step_operator(const pls_ipb_desc *basis, double *Pt, int64_t ldpt, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(basis->Q && basis->Linv, "ipb_build_step_operator: the descriptor needs Q (pls_ipb_build_whitened) and Linv");
  PLS_REQUIRE(Pt && ldpt >= basis->m && (ldpt & 1) == 0 && (reinterpret_cast<uintptr_t>(Pt) & 15) == 0,
              "ipb_build_step_operator: Pt must be 16-byte aligned with an even leading dimension >= M");
  // Pt[i][j] = sum_k Linv[k][i] Q[k][j]  (Linv lower triangular: only k >= i)
  return gemm_tn_ex(basis->Linv, basis->ldlinv, basis->Q, basis->ldq, Pt, ldpt, basis->m, basis->m, basis->m, 1.0, 0.0, 2, S(stream));
}

int pls_ipb_whiten(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, double *Sw, int64_t lds, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(U && Sw && U != Sw && j >= 0 && ldu >= j && lds >= j, "ipb_whiten: bad arguments");
  if (j == 0) return PLS_OK;
  const pls_chol_desc f = ipb_factor(basis);
  return chol_forward_solve(&f, U, ldu, j, Sw, lds, S(stream));
}

int pls_ipb_unwhiten(const pls_ipb_desc *basis, const double *Sw, int64_t lds, int64_t j, double *U, int64_t ldu, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(basis->LcT, "ipb_unwhiten: the descriptor needs LcT");
  PLS_REQUIRE(U && Sw && U != Sw && j >= 0 && ldu >= j && lds >= j, "ipb_unwhiten: bad arguments");
  if (j == 0) return PLS_OK;
  return gemm_tn_ex(basis->LcT, basis->ldlct, Sw, lds, U, ldu, basis->m, j, basis->m, 1.0, 0.0, 1, S(stream), basis->tri_scratch,
                    basis->tri_scratch_bytes);
}

size_t pls_ipb_whitened_workspace_bytes(const pls_ipb_desc *basis, int64_t j) {
  if (!basis || j <= 0) return 0;
  return (size_t)(2 * cdiv(basis->m, 128)) * j * sizeof(double);
}

static int ipb_whitened_step_impl(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *Sw, int64_t lds, int64_t j,
                                  double eta, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                                  int32_t out_mode, double *energy_in, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  rc = validate_noise(noise, basis->m, j);
  if (rc) return rc;
  rc = validate_blocks(blocks, j);
  if (rc) return rc;
  PLS_REQUIRE(cost->cost == PLS_COST_GAUSSIAN && cost->link == PLS_LINK_IDENTITY, "ipb_whitened_step: Gaussian cost with the identity link only");
  PLS_REQUIRE(basis->Q && basis->ct && basis->q_inv_noise == 1.0 / cost->p[0],
              "ipb_whitened_step: the descriptor's Q / ct were not built for this observation noise (pls_ipb_build_whitened)");
  rc = validate_lag(blocks);
  if (rc) return rc;
  if (blocks && blocks->energy_flush) {
    PLS_REQUIRE(Sw && j > 0 && lds >= j, "ipb_whitened_step: energy_flush needs the particle matrix of the step calls");
    return fast_step_launch(ipb_whitened_op(basis), Sw, lds, j, make_etap(eta, blocks), make_noisep(noise, blocks), nullptr, 0, 0,
                            nullptr, nullptr, 0, S(stream), "ipb_whitened_step", nullptr, nullptr, make_lag(blocks));
  }
  PLS_REQUIRE(Sw && out && out != Sw && j >= 0 && lds >= j && ldo >= j && eta >= 0.0, "ipb_whitened_step: bad arguments");
  PLS_REQUIRE(out_mode == 0 || out_mode == 1, "ipb_whitened_step: out_mode must be 0 or 1");
  if (j == 0) return PLS_OK;
  rc = fast_step_launch(ipb_whitened_op(basis), Sw, lds, j, make_etap(eta, blocks), make_noisep(noise, blocks), out, ldo,
                        out_mode, energy_in, workspace, workspace_bytes, S(stream), "ipb_whitened_step",
                        blocks ? blocks->energy_sums : nullptr, blocks ? blocks->energy_sync : nullptr, make_lag(blocks));
  return rc ? rc : finish_sums16(blocks, energy_in, j, S(stream));
}

int pls_ipb_whitened_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *Sw, int64_t lds, int64_t j,
                          double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode, double *energy_in,
                          void *workspace, size_t workspace_bytes, void *stream) {
  return ipb_whitened_step_impl(basis, cost, Sw, lds, j, eta, nullptr, noise, out, ldo, out_mode, energy_in, workspace,
                                workspace_bytes, stream);
}

int pls_ipb_whitened_step_blocks(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *Sw, int64_t lds, int64_t j,
                                 const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                                 int32_t out_mode, double *energy_in, void *workspace, size_t workspace_bytes, void *stream) {
  PLS_REQUIRE(blocks != nullptr, "ipb_whitened_step_blocks: block descriptor is NULL");
  return ipb_whitened_step_impl(basis, cost, Sw, lds, j, 0.0, blocks, noise, out, ldo, out_mode, energy_in, workspace,
                                workspace_bytes, stream);
}

// ---- whitened coordinates for every cost: the prior as rows of the forward operand (pls_ipb_desc.Awa) ------------------------
int pls_ipb_build_whitened_operand(const pls_ipb_desc *basis, double *Awa, int64_t ldawa, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  PLS_REQUIRE(basis->LinvT && basis->ldlinvt >= basis->m, "ipb_build_whitened_operand: the descriptor needs the inverse factor LinvT");
  PLS_REQUIRE(Awa && ldawa >= basis->m && (ldawa & 1) == 0 && (reinterpret_cast<uintptr_t>(Awa) & 15) == 0,
              "ipb_build_whitened_operand: Awa must be 16-byte aligned with an even leading dimension >= m");
  // rows [0, n): (k(X,Z) Lc^-T)[x][i] = sum_k Kzx[k][x] LinvT[k][i]
  rc = pls_gemm_tn(basis->Kzx, basis->ldkzx, basis->LinvT, basis->ldlinvt, Awa, ldawa, basis->n, basis->m, basis->m, 1.0, 0.0, stream);
  if (rc) return rc;
  // rows [n, n + m): sqrt(m) Lc^-T = sqrt(m) LinvT
  hipLaunchKernelGGL(scale_copy_2d_kernel, dim3((unsigned)cdiv(basis->m * basis->m, 256)), dim3(256), 0, S(stream), basis->LinvT,
                     basis->ldlinvt, Awa + basis->n * ldawa, ldawa, basis->m, basis->m, sqrt((double)basis->m));
  return check_launch("scale_copy_2d");
}

static bool ipb_whitened_generic_ok(const pls_ipb_desc *b, const double *y, int64_t j) {
  if (!b->Awa || b->ldawa < b->m || g_small_rank_step.load() == 0) return false;
  if (!small_rank_ok(b->Awa, b->ldawa, b->m) || (reinterpret_cast<uintptr_t>(y) & 15)) return false;
  return j <= 4096 && 4.0 * (double)(b->n + b->m) * (double)b->m * (double)j <= 8e9;  // (sr_step_route_for's launch-bound window)
}

int pls_ipb_whitened_generic_applies(const pls_ipb_desc *basis, const double *y, int64_t j) {
  if (!basis || validate_ipb(basis) != PLS_OK || !y || j <= 0) return 0;
  return ipb_whitened_generic_ok(basis, y, j) ? 1 : 0;
}

size_t pls_ipb_whitened_generic_workspace_bytes(const pls_ipb_desc *basis, int64_t j) {
  if (!basis || j <= 0) return 0;
  return sr_step_workspace_bytes(basis->m, j, basis->n + basis->m);
}

int pls_ipb_whitened_generic_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, const double *Sw,
                                  int64_t lds, int64_t j, double eta, const pls_block_desc *blocks, const pls_noise_desc *noise,
                                  double *out, int64_t ldo, int32_t out_mode, double *energy_in, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  rc = validate_noise(noise, basis->m, j);
  if (rc) return rc;
  rc = validate_blocks(blocks, j);
  if (rc) return rc;
  PLS_REQUIRE(!blocks || (!blocks->energy_partials && !blocks->energy_partials_prev && !blocks->energy_flush),
              "ipb_whitened_generic_step: lagged energies exist on the Gaussian/identity routes only");
  PLS_REQUIRE(y && Sw && out && out != Sw && j >= 0 && lds >= j && ldo >= j && eta >= 0.0, "ipb_whitened_generic_step: bad arguments");
  PLS_REQUIRE(out_mode == 0 || out_mode == 1, "ipb_whitened_generic_step: out_mode must be 0 or 1");
  if (j == 0) return PLS_OK;
  if (!ipb_whitened_generic_ok(basis, y, j))
    return fail(PLS_ERR_INVALID_ARGUMENT, "ipb_whitened_generic_step: needs pls_ipb_desc.Awa (pls_ipb_build_whitened_operand), at most 128 "
                "inducing points, 16-byte aligned targets and a launch-bound problem (pls_ipb_whitened_generic_applies)");
  if (!sr_step_fits(basis->m, basis->n + basis->m, j, blocks, energy_in, workspace ? workspace_bytes : 0))
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_whitened_generic_step: workspace %zu bytes, need %zu", workspace_bytes,
                pls_ipb_whitened_generic_workspace_bytes(basis, j));
  SrStepOperands ops{basis->Awa, basis->ldawa, basis->m, basis->n + basis->m, Sw, lds, nullptr, 0, nullptr, 0.0};
  ops.n_data = basis->n;
  bool taken = false;
  rc = sr_step_launch(ops, make_costp(cost), y, j, make_etap(eta, blocks), make_noisep(noise, blocks), out, ldo, out_mode, energy_in,
                      blocks, workspace, workspace ? workspace_bytes : 0, S(stream), &taken);
  if (rc) return rc;
  return taken ? PLS_OK : fail(PLS_ERR_WORKSPACE_TOO_SMALL, "ipb_whitened_generic_step: the one-launch step refused a workspace it was sized for");
}

int pls_ipb_whitened_energy(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *Sw, int64_t lds, int64_t j,
                            double *e, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = validate_ipb(basis);
  if (rc) return rc;
  rc = validate_cost(cost);
  if (rc) return rc;
  PLS_REQUIRE(cost->cost == PLS_COST_GAUSSIAN && cost->link == PLS_LINK_IDENTITY, "ipb_whitened_energy: Gaussian cost with the identity link only");
  PLS_REQUIRE(basis->Q && basis->ct && basis->q_inv_noise == 1.0 / cost->p[0],
              "ipb_whitened_energy: the descriptor's Q / ct were not built for this observation noise");
  PLS_REQUIRE(Sw && e && j >= 0 && lds >= j, "ipb_whitened_energy: bad arguments");
  if (j == 0) return PLS_OK;
  return fast_energy_launch(ipb_whitened_op(basis), Sw, lds, j, e, workspace, workspace_bytes, S(stream), "ipb_whitened_energy");
}

size_t pls_select_inducing_workspace_bytes(int64_t n, int64_t m) {
  if (n <= 0 || m <= 1) return 0;
  const size_t nparts = 1024;
  return align_up((size_t)(m - 1) * n * sizeof(double), 256) + align_up((size_t)n * sizeof(double), 256) +
         align_up((size_t)n, 256) + align_up(nparts * (2 * sizeof(double) + sizeof(int64_t)), 256) + 256;
}

int pls_select_inducing_conditional_variance(int32_t kernel_kind, const double *x, int64_t n, int64_t d,
                                             const double *lengthscale, double outputscale, int64_t m, double jitter,
                                             double threshold, int64_t *indices, int64_t *count, void *workspace,
                                             size_t workspace_bytes, void *stream) {
  PLS_REQUIRE(kernel_kind == PLS_KERNEL_RBF_ARD || kernel_kind == PLS_KERNEL_LINEAR, "unknown kernel kind %d", kernel_kind);
  PLS_REQUIRE(x && indices && count, "select_inducing: NULL pointer");
  PLS_REQUIRE(m > 1, "select_inducing: Must have at least 2 inducing points");  // conditional_variance.py:57
  PLS_REQUIRE(n >= m && d >= 1 && d <= 64, "select_inducing: need n >= m and 1 <= d <= 64");
  PLS_REQUIRE(kernel_kind != PLS_KERNEL_RBF_ARD || lengthscale, "select_inducing: RBF needs lengthscale");
  const size_t need = pls_select_inducing_workspace_bytes(n, m);
  if (!workspace || workspace_bytes < need)
    return fail(PLS_ERR_WORKSPACE_TOO_SMALL, "select_inducing: workspace %zu < %zu bytes", workspace_bytes, need);
  char *w = static_cast<char *>(workspace);
  double *ci = reinterpret_cast<double *>(w);
  w += align_up((size_t)(m - 1) * n * sizeof(double), 256);
  double *di = reinterpret_cast<double *>(w);
  w += align_up((size_t)n * sizeof(double), 256);
  unsigned char *chosen = reinterpret_cast<unsigned char *>(w);
  w += align_up((size_t)n, 256);
  const int nparts = (int)(cdiv(n, CV_BLOCK) < 1024 ? cdiv(n, CV_BLOCK) : 1024);
  double *pval = reinterpret_cast<double *>(w);
  double *psum = pval + 1024;
  int64_t *pidx = reinterpret_cast<int64_t *>(psum + 1024);
  w += align_up((size_t)1024 * (2 * sizeof(double) + sizeof(int64_t)), 256);
  CvState *st = reinterpret_cast<CvState *>(w);
  hipStream_t s = S(stream);
  const unsigned gn = (unsigned)cdiv(n, CV_BLOCK);
  hipLaunchKernelGGL(cv_init_kernel, dim3(gn), dim3(CV_BLOCK), 0, s, kernel_kind, x, n, (int)d, lengthscale, outputscale, jitter,
                     di, chosen, st);
  int rc = check_launch("cv_init");
  if (rc) return rc;
  for (int64_t it = 0; it < m; ++it) {
    hipLaunchKernelGGL(cv_argmax_partial_kernel, dim3((unsigned)nparts), dim3(CV_BLOCK), 0, s, di, chosen, n, pval, pidx, psum);
    hipLaunchKernelGGL(cv_argmax_final_kernel, dim3(1), dim3(CV_BLOCK), 0, s, pval, pidx, psum, nparts, di, chosen, indices, m,
                       threshold, it == 0 ? 1 : 0, st);
    if (it + 1 < m)
      hipLaunchKernelGGL(cv_update_kernel, dim3((unsigned)cdiv(n, CV_COLS)), dim3(CV_BLOCK), 0, s, kernel_kind, x, n, (int)d, lengthscale, outputscale,
                         jitter, it, ci, di, st);
    rc = check_launch("cv_iteration");
    if (rc) return rc;
  }
  hipError_t e = hipMemcpyAsync(count, &st->count, sizeof(int64_t), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(PLS_ERR_HIP, "select_inducing: %s", hipGetErrorString(e));
  return PLS_OK;
}

}  // extern "C"
