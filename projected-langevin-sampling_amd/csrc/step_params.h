// Per-column step size and the noise source of a Langevin update, shared by the translation units whose kernels apply it
// (plship.hip: the update kernel and the fused Gaussian epilogue; small_rank_step.hip: the one-launch small-rank step).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/plship.h"

namespace plship {

struct NoiseP {
  int kind;
  const double *xi;
  int64_t ldxi;
  uint64_t seed, step;
  int64_t j_offset;
  const uint64_t *step_base;  // optional device counter added to `step` at run time (graph replays)
  int64_t block_cols;         // > 0: the columns are blocks of independent runs (pls_block_desc); Philox column = column inside the block
  __device__ uint64_t live_step() const { return step_base ? step + *step_base : step; }
  __device__ int64_t global_column(int64_t col) const { return j_offset + (block_cols > 0 ? col % block_cols : col); }
};

// step size of a column: one scalar, or one per column block (pls_block_desc; the batched step-size search)
struct EtaP {
  double eta;
  const double *blocks;  // device array, NULL = the scalar
  int64_t block_cols;
  __device__ double at(int64_t col) const { return blocks ? blocks[col / block_cols] : eta; }
};

inline EtaP make_etap(double eta, const pls_block_desc *b) {
  EtaP e{eta, nullptr, 0};
  if (b) {
    e.blocks = b->eta;
    e.block_cols = b->block_cols;
  }
  return e;
}

inline NoiseP make_noisep(const pls_noise_desc *n, const pls_block_desc *blocks = nullptr) {
  NoiseP p;
  p.block_cols = blocks ? blocks->block_cols : 0;
  if (!n) {
    p.kind = PLS_NOISE_NONE;
    p.xi = nullptr;
    p.ldxi = 0;
    p.seed = p.step = 0;
    p.j_offset = 0;
    p.step_base = nullptr;
    return p;
  }
  p.kind = n->kind;
  p.xi = n->xi;
  p.ldxi = n->ldxi;
  p.seed = n->seed;
  p.step = n->step;
  p.j_offset = n->j_offset;
  p.step_base = n->step_base;
  return p;
}

}  // namespace plship
