// GEMM launch helpers shared by the translation units that instantiate gemm_tn_f64_kernel.
#pragma once
#include "common.h"
#include "gemm_tn_f64.h"

namespace plship {

#ifdef PLS_STAMP
extern unsigned long long *g_stamp_buffer;  // diagnostic build only (tools/stamp_probe.py); defined in plship.hip
#endif

// ---------------------------------------------------------------------------------------------------------------
// GEMM launcher
// ---------------------------------------------------------------------------------------------------------------
template <int BI, int BJ, int WI, int WJ, class Epi, int MINW = ((BI >= 128) ? 2 : 4)>
static int launch_gemm_cfg(GemmShape g, const Epi &epi, hipStream_t st) {
  constexpr int BK = 16;  // MINW: waves per SIMD the register allocation must leave room for
  constexpr int NT = (BI / WI) * (BJ / WJ) * 64;
  constexpr size_t lds_bytes = (size_t)2 * BK * ((BI + 16) + (BJ + 16)) * sizeof(double);
  auto kern = gemm_tn_f64_kernel<BI, BJ, WI, WJ, BK, MINW, Epi>;
  static std::atomic<uint64_t> lds_ready{0};
  if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes, lds_ready)) return rc;
#ifdef PLS_STAMP
  g.stamps = g_stamp_buffer;
  // diagnostic build only: PLS_STAMP_LDS_PAD=<bytes> of extra dynamic LDS (e.g. 90000: one workgroup per CU)
  static const size_t pad = getenv("PLS_STAMP_LDS_PAD") ? (size_t)atol(getenv("PLS_STAMP_LDS_PAD")) : 0;
  static std::atomic<uint64_t> pad_ready{0};
  if (pad)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_bytes + pad, pad_ready)) return rc;
  const size_t lds_launch = lds_bytes + pad;
#else
  const size_t lds_launch = lds_bytes;
#endif
  g.nti = (int)cdiv(g.I, BI);
  g.ntj = (int)cdiv(g.J, BJ);
  // (a triangular operand pairs tile rows t and nti - 1 - t in one workgroup: gemm_tn_f64_kernel)
  const int64_t nwg = (int64_t)((g.tri && Epi::kTag == 1) ? (g.nti + 1) / 2 : g.nti) * g.ntj;
  if (nwg <= 0) return PLS_OK;
  if (nwg > 0x7fffffff) return fail(PLS_ERR_INVALID_ARGUMENT, "gemm: too many tiles");
  unsigned nsplit = 1;
  if (g.kchunk > 0 && g.kchunk < g.K) nsplit = (unsigned)cdiv(g.K, g.kchunk);
  {
    LaunchScope scope(Epi::kTag, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, nsplit), dim3(NT), lds_launch, st, g, epi);
  }
  return check_launch("gemm_tn_f64");
}

static inline bool use_big_tiles(int64_t I, int64_t J, int64_t nsplit = 1) { return cdiv(I, 128) * cdiv(J, 128) * nsplit >= 256; }

// kchunk > 0 (EpiStore only): split-K into cdiv(K, kchunk) slabs, one grid.y plane each
template <class Epi>
static int launch_gemm(const double *L, int64_t ldl, const double *R, int64_t ldr, int64_t I, int64_t J, int64_t K,
                       const Epi &epi, hipStream_t st, int64_t kchunk = 0, int tri = 0) {
  GemmShape g{L, ldl, R, ldr, I, J, K, 0, 0, kchunk, tri};
  const int64_t nsplit = (kchunk > 0 && kchunk < K) ? cdiv(K, kchunk) : 1;
  if (use_big_tiles(I, J, nsplit)) return launch_gemm_cfg<128, 128, 64, 64>(g, epi, st);
  return launch_gemm_cfg<64, 64, 32, 32>(g, epi, st);
}

}  // namespace plship
