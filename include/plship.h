/*
 * plship.h -- C ABI of libplship.so: the MI355X (gfx950) projected-Langevin-sampling hot path.
 *
 * The reference (jswu18/projected-langevin-sampling) is pure Python/torch and has no FFI; its
 * boundary for this path is the Python plugin API (PLS / PLSBasis / PLSCost / PLSLinkFunction).
 * The entry points below are what a ctypes binding for that API calls.  Each one cites the
 * reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every array pointer is a DEVICE pointer to float64 unless the name ends in _host;
 *   - matrices are row-major with an explicit leading dimension (elements, not bytes);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued asynchronously, nothing synchronises, nothing allocates (callers hand in
 *     workspaces), so every call may be captured into a hipGraph;
 *   - return value: 0 = PLS_OK, otherwise a pls_status; pls_last_error() returns a
 *     thread-local human readable message for the last failing call on this thread;
 *   - no global mutable state: descriptor structs are plain data owned by the caller, the
 *     library is thread-compatible (different threads may call with different streams).
 *
 * N = training points, M = inducing points, Mk = kept eigen-directions (<= M), J = particles
 * (columns; a rank may own a J-shard and pass its global column offset), D = input dim.
 */
#ifndef PLSHIP_H
#define PLSHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLSHIP_ABI_VERSION 5

typedef enum {
  PLS_OK = 0,
  PLS_ERR_INVALID_ARGUMENT = 1,
  PLS_ERR_HIP = 2,
  PLS_ERR_WORKSPACE_TOO_SMALL = 3,
  PLS_ERR_UNSUPPORTED = 4
} pls_status;

/* base kernel k (third party gpytorch ScaleKernel(RBFKernel(ard)) in the reference,
 * constructed at experiments/uci/regression/main.py:171-173; MockKernel = mockers/kernel.py:13-23) */
typedef enum { PLS_KERNEL_RBF_ARD = 0, PLS_KERNEL_LINEAR = 1 } pls_kernel_kind;

/* src/projected_langevin_sampling/costs/{gaussian,poisson,bernoulli,student_t,multimodal}.py */
typedef enum {
  PLS_COST_GAUSSIAN = 0,
  PLS_COST_POISSON = 1,
  PLS_COST_BERNOULLI = 2,
  PLS_COST_STUDENT_T = 3,
  PLS_COST_MULTIMODAL = 4
} pls_cost_kind;

/* src/projected_langevin_sampling/link_functions.py:30-80 */
typedef enum {
  PLS_LINK_IDENTITY = 0,
  PLS_LINK_SQUARE = 1,
  PLS_LINK_SIGMOID = 2,
  PLS_LINK_PROBIT = 3
} pls_link_kind;

/* How d cost / d f is evaluated.
 *   PLS_DERIV_REFERENCE: what the reference's calculate_cost_derivative dispatch does -- the closed
 *     form when (cost, link) is one of its matched pairs (gaussian.py:103-106, poisson.py:97-100,
 *     bernoulli.py:92-95, student_t.py:103-106), otherwise the autograd value.
 *   PLS_DERIV_AUTOGRAD: the value costs/base.py:68-84 (vmap(jacfwd)) returns, evaluated analytically
 *     by the chain rule (clip has zero slope outside its range), for every (cost, link) pair. */
typedef enum { PLS_DERIV_REFERENCE = 0, PLS_DERIV_AUTOGRAD = 1 } pls_deriv_mode;

typedef struct {
  int32_t cost;       /* pls_cost_kind */
  int32_t link;       /* pls_link_kind */
  int32_t deriv_mode; /* pls_deriv_mode */
  int32_t reserved;
  /* p[0..3]: gaussian {observation_noise (a VARIANCE, gaussian.py:71,86)};
   *          student_t {degrees_of_freedom, scale};
   *          multimodal {observation_noise (a STD, multimodal.py:56), shift, bernoulli_noise};
   *          poisson / bernoulli: unused */
  double p[4];
  double jitter; /* clip of sigmoid / probit links (link_functions.py:36, :64), default 1e-10 */
} pls_cost_desc;

/* Langevin noise source for one step. */
typedef enum {
  PLS_NOISE_NONE = 0,     /* drift only (tests) */
  PLS_NOISE_INJECTED = 1, /* xi is read from memory: parity runs inject the oracle's noise */
  PLS_NOISE_PHILOX = 2    /* in-kernel Philox4x32-10 + Box-Muller, counter = (row, global column, step) */
} pls_noise_kind;

typedef struct {
  int32_t kind; /* pls_noise_kind */
  int32_t reserved;
  const double *xi; /* PLS_NOISE_INJECTED: (rows x J) standard-normal matrix */
  int64_t ldxi;
  uint64_t seed;    /* PLS_NOISE_PHILOX */
  uint64_t step;    /* PLS_NOISE_PHILOX: step counter, so every step draws fresh noise */
  int64_t j_offset; /* global index of local column 0 (J-sharding: results do not depend on the GPU count) */
  /* PLS_NOISE_PHILOX, optional: a DEVICE counter read at run time; the step used is *step_base + step.  Launch arguments
   * are frozen into a captured hipGraph, this word is not: a graph of K steps (step = 0..K-1) followed by
   * pls_counter_add(step_base, K) draws fresh noise on every replay. */
  const uint64_t *step_base;
} pls_noise_desc;

/* Orthonormal basis state (reference: basis/orthonormal.py:22-68), produced by the setup calls below.
 *   A  = V~^T k(Z,X)      (Mk x N), lda   -- V~ = V diag(1/sqrt(Mk*lambda)) (orthonormal.py:63-68)
 *   At = k(X,Z) V~        (N x Mk), ldat  -- the same matrix transposed, so both contractions stream
 *                                            k-major operands (DESIGN.md "data layout")
 *   lam = kept eigenvalues of k(Z,Z)/M (Mk)
 * Optional Gaussian/identity fast path (paper's O(M^3 + J M^2) step): B = A A^T (Mk x Mk), c = A y (Mk entries,
 * followed by ONE more entry c[Mk] = y^T y used by the fast energy). */
typedef struct {
  int64_t mk, n;
  const double *A;
  int64_t lda;
  const double *At;
  int64_t ldat;
  const double *lam;
  const double *B; /* may be NULL */
  int64_t ldb;
  const double *c; /* may be NULL */
} pls_onb_desc;

/* Cholesky factor of an SPD matrix K = Lc Lc^T as pls_chol_factor leaves it on the device:
 *   Lc  (M x M, lower, row-major) and LcT = Lc^T (upper): the same factor stored both ways, so that either is a k-major
 *       operand of the MFMA contraction;
 *   Sf, Sb (M x M): the block forward / backward substitution operators.  With D_b the inverse of the b-th 128 x 128
 *       DIAGONAL block of Lc:  Sf[k][i] = -(D_b Lc[b, k-block])^T for k-blocks left of block b (i in block b), D_b^T on the
 *       diagonal block;  Sb[k][i] = -(Lc[k-block, b] D_b) for k-blocks below, D_b on the diagonal.  One block row of a
 *       solve is then a single MFMA k-loop; K^-1 itself is never formed.  NULL if the factor is only used for L xi. */
typedef struct {
  int64_t m;
  const double *Lc;
  int64_t ldlc;
  const double *LcT;
  int64_t ldlct;
  const double *Sf;
  int64_t ldsf;
  const double *Sb;
  int64_t ldsb;
  /* Optional (ABI 3): the inverse factor Linv = Lc^-1 (M x M, lower, row-major) and its transpose LinvT, as
   * pls_chol_build_inverse leaves them (column c of Lc^-1 by forward substitution of e_c).  With them a solve is one
   * (forward) or two (forward + backward) TRIANGULAR PRODUCTS on the MFMA contraction instead of a block substitution:
   * any number of right-hand sides fills the chip, which the substitution -- a workgroup per 32 columns, serial over the
   * block rows -- does not on the narrow J-shard of an 8-GPU run.  NULL: block substitution only. */
  const double *Linv;
  int64_t ldlinv;
  const double *LinvT;
  int64_t ldlinvt;
  /* Optional (ABI 4): scratch for BALANCED triangular products on few output tiles (pls_tri_scratch_bytes(M, J) bytes,
   * 16-byte aligned).  A triangular operand gives tile row t of a product a contraction of (t + 1) * 64 rows; when every
   * workgroup is resident at once (M = 1024, J = 1024: 256 tiles on 256 CUs) the launch lasts as long as its heaviest tile,
   * i.e. as long as the full product.  With the scratch, tile rows t and nti - 1 - t are shared by two workgroups with equal
   * loads; the heavy tile's two partial sums meet through a scratch slot and are added by whichever workgroup arrives
   * second (deterministic: a + b == b + a; nobody waits).  Contract: the first 16 KB (flag words) are ZERO when the scratch
   * is first handed in; every call leaves them zero; calls that may run concurrently (different streams) need different
   * scratches.  NULL: one tile per workgroup. */
  void *tri_scratch;
  size_t tri_scratch_bytes;
} pls_chol_desc;

/* Inducing-point basis state (reference: basis/inducing_point.py:23-50).
 *   Kzx = k(Z,X) (M x N), Kxz = its transpose (N x M);
 *   k(Z,Z) enters through its Cholesky factor (pls_chol_factor): LcT = Lc^T for the noise colouring e = Lc xi, and the
 *   substitution operators Sf / Sb for the two solves per step, V = k(Z,Z)^-1 U = Lc^-T Lc^-1 U (the reference's
 *   gpytorch.solve, inducing_point.py:89-93, :130-132).
 *   W = k(Z,Z)^-1 (M x M, symmetric) is OPTIONAL and only used for A/B runs: with Sf/Sb == NULL, or after
 *   pls_set_option(PLS_OPT_IPB_EXPLICIT_INVERSE, 1), the solves are replaced by the contraction W U. */
typedef struct {
  int64_t m, n;
  const double *Kzx;
  int64_t ldkzx;
  const double *Kxz;
  int64_t ldkxz;
  const double *W; /* may be NULL when Sf / Sb are set */
  int64_t ldw;
  const double *LcT; /* may be NULL when noise is injected already coloured */
  int64_t ldlct;
  const double *B; /* optional Gaussian fast path: Kzx Kxz (M x M), see pls_ipb_build_gaussian */
  int64_t ldb;
  const double *c; /* optional: Kzx y (M), then y^T y */
  const double *Sf; /* substitution operators of pls_chol_factor (see pls_chol_desc) */
  int64_t ldsf;
  const double *Sb;
  int64_t ldsb;
  /* Optional (ABI 3): the inverse factor of k(Z,Z) (see pls_chol_desc) ... */
  const double *Linv;
  int64_t ldlinv;
  const double *LinvT;
  int64_t ldlinvt;
  /* ... and the Gaussian/identity operator in WHITENED coordinates S = Lc^-1 U (pls_ipb_build_whitened, built for ONE
   * observation noise: q_inv_noise = 1 / sigma2):  Q = Lc^-1 (B / sigma2 + M I) Lc^-T (M x M), ct = Lc^-1 c / sigma2 (M entries,
   * then y^T y).  With them pls_ipb_step's Gaussian path is forward solve + Q S (fused update kernel) + Lc dS, and
   * pls_ipb_whitened_step advances S itself with ONE M x M x J contraction per step. */
  const double *Q;
  int64_t ldq;
  const double *ct;
  double q_inv_noise;
  /* Optional (ABI 4): scratch for balanced triangular products (see pls_chol_desc.tri_scratch): the forward solve and the
   * product with Lc of pls_ipb_step's whitened route, pls_ipb_whiten / _unwhiten, the solves of the other routes. */
  void *tri_scratch;
  size_t tri_scratch_bytes;
  /* Optional (ABI 4): Pt = Lc^-T Q (M x M; pls_ipb_build_step_operator), the whitened operator with the forward solve folded
   * in: Q S = Q Lc^-1 U = P U, so pls_ipb_step computes dS from U itself and a call is TWO launches (P U with the fused
   * update + noise, then Lc dS) and 3 M^2 J flop instead of three and 4 M^2 J.  Built for the same observation noise as
   * Q (q_inv_noise).  A call that asks for energy_in keeps the three-launch route (the energy is a quadratic form in S). */
  const double *Pt;
  int64_t ldpt;
  /* Optional (ABI 5): Awa ((n + m) x m, k-major like Kxz, 16-byte aligned rows, even ldawa; pls_ipb_build_whitened_operand),
   * the forward operand of WHITENED particles with the prior as rows: rows [0, n) = k(X,Z) Lc^-T (F = Awa S for S = Lc^-1 U),
   * rows [n, n + m) = sqrt(m) Lc^-T, whose "cost" is f^2 / 2 -- their back-projection is the prior drift m (Lc^T Lc)^-1 S and
   * their cost the prior energy (m / 2) |k(Z,Z)^-1 U|^2.  With it the step of a cost without the Gaussian algebra in whitened
   * coordinates is the one-launch small-rank step and nothing else (pls_ipb_whitened_generic_step). */
  const double *Awa;
  int64_t ldawa;
} pls_ipb_desc;

/* Step-size search (experiments/runners.py:331-446): the S candidate step sizes run as S column blocks of ONE particle
 * matrix, block b = columns [b * block_cols, (b + 1) * block_cols).  Every block is an independent copy of the same
 * sampler: its step size is eta[b], and its noise stream is the one a stand-alone run of block_cols particles would
 * draw (Philox column = j_offset + column inside the block), so a batched search reproduces the sequential one.
 * eta is a DEVICE array: a block whose search has ended is frozen by writing 0 (update and noise both vanish). */
typedef struct {
  int64_t block_cols;
  const double *eta; /* device, cdiv(j, block_cols) entries */
  /* Optional output (ABI 3), steps with energy_in != NULL only: cdiv(j, 256) doubles, entry i = the sum of the per-particle
   * energies of columns [256 i, 256 (i + 1)) in the library's fixed order (pls_chunk_sums).  On the Gaussian/identity fast
   * paths the launch that finishes the energy by-product writes them (no second launch per training iteration); every
   * other route appends one pls_chunk_sums launch, so the field is honoured whatever the descriptor and the options
   * select.  The mean energy of a block of columns that starts at a multiple of 256 is then a few host additions over
   * these entries (ascending order: the value pls_block_means returns, bit for bit).  May point to pinned host memory
   * mapped into the device.  NULL: not written. */
  double *energy_sums;
  /* Optional (ABI 4), Gaussian/identity fast paths with energy_in != NULL: cdiv(j, 256) 32-bit counters (device), ZERO when
   * first handed in; every call leaves them zero.  With them the step launch FINISHES the energies itself -- the workgroup
   * that arrives last at a 256-column chunk adds the partial rows of that chunk in their fixed order, writes energy_in and
   * energy_sums (the values the finishing launch computes, bit for bit) -- so a training iteration is ONE launch.  Calls
   * that may run concurrently need different counters.  NULL (or any other route): a finishing launch follows. */
  uint32_t *energy_sync;
  /* Optional (ABI 4), LAGGED energies for training loops on the Gaussian/identity fast paths (pls_onb_step_blocks,
   * pls_ipb_whitened_step_blocks).  The reduction of a step's energy by-product over the tile rows is a global dependency
   * behind its last MFMA: finished inside the launch it adds 4.4-5 us of serial tail, as a launch of its own 6-9 us.  Instead:
   *   energy_partials       (out) this launch leaves ONLY its partial rows here (pls_energy_partials_bytes(rows, j) bytes;
   *                         energy_in / energy_sums / energy_sync are then not used);
   *   energy_partials_prev  (in)  the partial rows the PREVIOUS launch of the loop left (another buffer: alternate two); this
   *                         launch finishes them at its START, under the landing of its first operand rows, into
   *   energy_prev           (out) the J energies of the previous launch's input particles, and
   *   energy_sums_prev      (out, optional) their 256-column chunk sums (may be pinned host memory);
   *   energy_flush          1: no step at all -- the last launch's partial rows are finished by a small launch of their own
   *                         (pass the particle matrix and step arguments of the step calls; out may be NULL).
   * The values are the ones the other forms compute, bit for bit; they arrive one launch later. */
  double *energy_partials;
  const double *energy_partials_prev;
  double *energy_prev;
  double *energy_sums_prev;
  int32_t energy_flush;
  int32_t reserved;
  /* Optional (ABI 5), orthonormal basis with at most 128 functions and a cost without the Gaussian algebra, problems in the
   * launch-bound regime (the sizes of the reference's own experiments: N, J in the hundreds to thousands): the whole
   * step -- projection, cost derivative, back-projection, the fixed-order sum over the row slabs, prior drift, noise, and with
   * energy_in the energies of the input particles and their energy_sums -- is ONE launch (csrc/small_rank_step.h; option
   * PLS_OPT_SMALL_RANK_STEP).  Its workgroups meet through pls_step_sync_words(j) 32-bit counters: handed in here they must
   * be ZERO when first handed in and every call leaves them zero (calls that may run concurrently need different
   * counters); NULL: the launch is preceded by a memset node over counters carved from the workspace. */
  uint32_t *step_sync;
  /* Optional output (ABI 5), steps with energy_in != NULL only: cdiv(j, 16) doubles, entry b = the sum of the per-particle
   * energies of columns [16 b, 16 (b + 1)) added in ascending column order.  The one-launch step writes them for free (the
   * workgroup that finishes a column block holds its sixteen energies), where energy_sums costs it a second hand-over
   * between workgroups; every other route appends one small launch.  A training loop adds the entries in ascending order
   * on the host.  May point to pinned host memory mapped into the device.  NULL: not written. */
  double *energy_sums16;
} pls_block_desc;

const char *pls_last_error(void);
int pls_abi_version(void);

/* Route options (the defaults are what bench.py measures; tests flip them to reach both code paths).  NOTHING here is
 * process-wide: every option is a value of the CALLING THREAD (each thread starts from the defaults), and a launch takes the
 * routes of the thread that makes the call -- two bases driven from two host threads can choose differently, and a test
 * that flips an option disturbs no other thread.  (The per-launch timeline of pls_timeline_begin / _end and the text behind
 * pls_last_error are per thread as well.)
 *   PLS_OPT_SMALL_RANK_MAX: bases with at most this many functions (0..128, default 128) take the fused small-rank
 *   kernels (F, d cost / d f and the back-projection in ONE pass: the N x J matrices F and G are never written);
 *   larger ranks, or 0, take the two-GEMM path.  Results agree to rounding, not bit for bit. */
typedef enum pls_option {
  PLS_OPT_SMALL_RANK_MAX = 1,
  /* 1: V = k(Z,Z)^-1 U of the inducing-point basis as the contraction W U with the explicit inverse (needs
   * pls_ipb_desc.W); 0 (default): two blocked triangular solves with the Cholesky factor (pls_chol_solve). */
  PLS_OPT_IPB_EXPLICIT_INVERSE = 2,
  /* Contractions with few output tiles (a narrow J-shard of an 8-GPU run: Mk x Mk x J/8) take 64 x 64 tiles whose k range
   * is split over two wave groups INSIDE the workgroup (csrc/gemm_tn_f64_kg.h), so that 256..1023 tiles still put 2..4
   * waves on every SIMD; the epilogue runs once on the fixed-order sum.  MODE: 0 off, 1 automatic (default), 2 / 3 force the
   * two- / one-group kernel wherever the operands are 16-byte aligned (A/B runs, tests).  MAX_TILES: the number of
   * 128 x 128 output tiles below which the automatic mode takes it (default 256: one workgroup per CU). */
  PLS_OPT_KSPLIT_MODE = 3,
  PLS_OPT_KSPLIT_MAX_TILES = 4,
  /* Solves with the Cholesky factor of k(Z,Z): 1 (default) = triangular products with the inverse factor wherever the
   * descriptor carries Linv / LinvT, 0 = block substitution (tri_solve_strip_kernel) always. */
  PLS_OPT_SOLVE_MODE = 5,
  /* (6, 7: the wave-pair fused kernel for 129 .. 256 functions of ABI 3; slower than the row-block back-projection at every
   * rank and removed in ABI 4) */
  /* Back-projection D = A G of a basis whose function count is above 128 and not a multiple of 128 (two-GEMM path):
   * 1 (default) = one launch of csrc/gemm_tn_f64_rows.h (equal-height tiles, the MFMA count follows the rank in steps of
   * 16, G read once); 0 = 128-row tiles plus 64- / 32- / 16-row remainder launches (round 2). */
  PLS_OPT_ROW_BLOCKS = 8,
  /* Triangular products on few output tiles with a scratch in the descriptor (pls_chol_desc.tri_scratch): 1 (default) =
   * balanced (csrc/gemm_tn_f64_kg.h, gemm_tn_f64_kg_tri_kernel), 0 = one tile per workgroup (A/B runs, tests). */
  PLS_OPT_TRI_BALANCE = 9,
  /* pls_ipb_step, Gaussian/identity without energy_in: 1 (default) = dS = -eta (P U - ct) + noise straight from U when the
   * descriptor carries Pt, 0 = forward solve, then Q S (A/B runs, tests). */
  PLS_OPT_IPB_STEP_OPERATOR = 10,
  /* 1 (default): pls_block_desc.energy_sync is honoured (the step launch finishes the energies); 0: always a finishing
   * launch (A/B runs, tests). */
  PLS_OPT_ENERGY_FUSED_FINISH = 11,
  /* 1 (default): the k-split kernel of narrow particle shards draws the Philox noise of its output block in front of its
   * k-loop, while the first operand rows travel; 0: in the epilogue, like the other tilings.  Same bits either way. */
  PLS_OPT_KG_NOISE_PREGEN = 12,
  /* Steps of the orthonormal basis with at most 128 functions and a cost without the Gaussian algebra: 1 (default) = ONE
   * launch (csrc/small_rank_step.h) while the problem is launch-bound (at most 4096 particles and 8 GFLOP per step), the
   * slab kernels of csrc/small_rank.h + update launch beyond; 0 = never; 2 = wherever the kernel applies (A/B runs, tests).
   * Results agree to rounding (another summation order over the data rows), not bit for bit. */
  PLS_OPT_SMALL_RANK_STEP = 13,
  /* pls_ipb_step on at most 128 inducing points, in front of the one-launch step above: 1 (default) = V = k(Z,Z)^-1 U and the
   * coloured Philox noise Lc xi in ONE launch (csrc/ipb_prep.h; needs the inverse factor Linv / LinvT in the descriptor and
   * PLS_OPT_SOLVE_MODE 1), 0 = the two triangular products, the fill and the third product as launches of their own (A/B
   * runs, tests).  Same draws either way; results agree to rounding. */
  PLS_OPT_IPB_PREP = 14
} pls_option;
/* Diagnostic: out[i] = op(x[i]) with the device exp (op 0) / log (op 1) the per-element kernels use (csrc/fmath.h), or
 * out[i] = x[i] / x[n + i] with their division (op 2: fast_div, IEEE special cases restored; op 3: fast_div_normal), so
 * that their accuracy can be pinned against libm.  Not on the step path. */
int pls_debug_math(int32_t op, const double *x, double *out, int64_t n, void *stream);
int pls_set_option(int32_t option, int64_t value);
int64_t pls_get_option(int32_t option); /* -1 for an unknown option */

/* Per-launch timeline (measurement only; off by default, zero cost when off).  Between begin and end, every
 * kernel the calling thread launches through this library is bracketed by two HIP events recorded on the
 * launch's own stream.  pls_timeline_end synchronises on them and returns, per launch, its duration in
 * milliseconds and a pls_kernel_tag; it returns the number of launches seen (<= capacity recorded). */
typedef enum {
  PLS_TAG_GEMM_STORE = 1,            /* gemm_tn_f64, plain / accumulate epilogue (back-projection A G, setup GEMMs) */
  PLS_TAG_GEMM_COST_DERIV = 2,       /* gemm_tn_f64, F tile -> d cost / d f epilogue */
  PLS_TAG_GEMM_COST_VALUE = 3,       /* gemm_tn_f64, F tile -> per-column cost partial sums */
  PLS_TAG_GEMM_LANGEVIN_GAUSSIAN = 4,/* gemm_tn_f64, B U with the whole Langevin update in the epilogue */
  PLS_TAG_LANGEVIN_UPDATE = 5,
  PLS_TAG_KERNEL_GRAM = 6,
  PLS_TAG_OTHER = 7,
  PLS_TAG_SMALL_RANK_DRIFT = 8,      /* small_rank_kernel: F, d cost / d f and the back-projection in one pass (rank <= 128) */
  PLS_TAG_SMALL_RANK_VALUE = 9,      /* small_rank_kernel: F and the per-column cost sums in one pass */
  PLS_TAG_TRI_SOLVE = 10,            /* tri_solve_strip_kernel: V = Lc^-T Lc^-1 U, forward + backward substitution in one launch */
  PLS_TAG_SMALL_RANK_STEP = 11,      /* small_rank_step_kernel: the whole step (and its energies) of a small-rank basis in one launch */
  PLS_TAG_IPB_PREP = 12              /* ipb_prep_kernel: V = k(Z,Z)^-1 U and the coloured noise Lc xi of a step, <= 128 inducing points */
} pls_kernel_tag;
int pls_timeline_begin(int32_t capacity);
int pls_timeline_end(float *ms, int32_t *tags, int32_t capacity, int32_t *count);

/* ---------------------------------------------------------------------------------------------
 * Primitive operators (un-fused entry points; a user-defined Python cost or basis composes these)
 * ------------------------------------------------------------------------------------------- */

/* out(n1 x n2) = k(x1, x2); x1 (n1 x d), x2 (n2 x d) row-major contiguous.
 * Replaces kernel.base_kernel(x1=.., x2=..) at orthonormal.py:36-41, inducing_point.py:41-46.
 * lengthscale: d values (RBF_ARD; ignored for LINEAR). */
int pls_kernel_gram(int32_t kernel_kind, const double *x1, int64_t n1, const double *x2, int64_t n2,
                    int64_t d, const double *lengthscale, double outputscale, double *out, int64_t ldout,
                    void *stream);

/* C(I x J) = alpha * L^T R + beta * C with L (K x I, ldl), R (K x J, ldr): the fp64 MFMA contraction.
 * Replaces every `@` on the path: orthonormal.py:106-108, :152-155; inducing_point.py:89-93, :144. */
int pls_gemm_tn(const double *L, int64_t ldl, const double *R, int64_t ldr, double *C, int64_t ldc, int64_t I,
                int64_t J, int64_t K, double alpha, double beta, void *stream);

/* G(N x J) = d cost / d f evaluated at F(N x J), y(N).  Replaces PLSCost.calculate_cost_derivative
 * (costs/gaussian.py:86-88, poisson.py:76-82, bernoulli.py:64-77, student_t.py:82-88, base.py:68-84). */
int pls_cost_derivative(const pls_cost_desc *cost, const double *F, int64_t ldf, const double *y, int64_t n,
                        int64_t j, double *G, int64_t ldg, void *stream);

/* c(J) = sum_n cost(y_n, F_nj).  Replaces PLSCost.calculate_cost (gaussian.py:63-73, poisson.py:59-66,
 * bernoulli.py:57-62, student_t.py:57-72, multimodal.py:37-77).  Deterministic summation order.
 * workspace: pls_cost_value_workspace_bytes(n, j) bytes. */
size_t pls_cost_value_workspace_bytes(int64_t n, int64_t j);
int pls_cost_value(const pls_cost_desc *cost, const double *F, int64_t ldf, const double *y, int64_t n, int64_t j,
                   double *c, void *workspace, size_t workspace_bytes, void *stream);

/* out = link(in + col_offset[col]) element-wise (rows x cols); col_offset may be NULL.
 * Replaces PLSLinkFunction.transform (link_functions.py:30-80) and, with the per-particle observation noise as
 * col_offset, PLSCost.predict_samples (costs/base.py:117-133). */
int pls_link_transform(int32_t link, double jitter, const double *in, int64_t ldin, int64_t rows, int64_t cols,
                       const double *col_offset, double *out, int64_t ldout, void *stream);

/* out[r] = sum_j (S[r][j] - shift[r])^power, power in {1, 2}, shift may be NULL; fixed summation order.
 * The two passes (power 1 -> mean, power 2 about the mean) replace prediction_samples.mean(dim=1) / .var(axis=1) at
 * costs/gaussian.py:49-52; on a J-sharded run each pass is followed by one all-reduce of `rows` doubles. */
int pls_row_power_sums(const double *S, int64_t lds, int64_t rows, int64_t cols, const double *shift, int32_t power,
                       double *out, void *stream);

/* out[r][k] = q[k]-quantile of row r of S (rows x cols), linear interpolation between order statistics at position
 * q * (cols - 1): the value torch.quantile(S, q, dim=1) returns (conformalise/pls.py:36-45, :57-62).  Up to 16384
 * samples per row one workgroup sorts the row in LDS (bitonic); longer rows (a large calibration split, the gathered
 * samples of a J-sharded run) take a radix selection of the two order statistics instead (nine passes over the row, no
 * workspace).  q: nq values in [0, 1] (device).  A row that contains a NaN yields NaN, like torch. */
int pls_row_quantiles(const double *S, int64_t lds, int64_t rows, int64_t cols, const double *q, int32_t nq, double *out,
                      int64_t ldout, void *stream);

/* *counter += increment on the stream (device word; see pls_noise_desc.step_base). */
int pls_counter_add(uint64_t *counter, uint64_t increment, void *stream);

/* out(rows x J) standard normals from the library's counter-based generator (same stream the fused
 * step uses).  Replaces torch.normal at basis/base.py:55-63 and samplers.py:30-35 for on-device runs. */
int pls_normal_fill(double *out, int64_t ldout, int64_t rows, int64_t j, uint64_t seed, uint64_t step,
                    int64_t j_offset, void *stream);

/* out[b] = mean of e[b * block_cols .. (b + 1) * block_cols) (the last block may be shorter), b < nblocks; fixed
 * summation order.  Replaces `.mean()` of the per-particle energies (orthonormal.py:126, inducing_point.py:115), per
 * step-size candidate when the particle matrix holds several (pls_block_desc).  `out` may be device memory or pinned
 * host memory mapped into the device (the training loop reads it after an event, without a copy kernel). */
int pls_block_means(const double *e, int64_t j, int64_t block_cols, double *out, void *stream);

/* out[i] = sum of e[256 i .. min(j, 256 (i + 1))): the chunk sums every energy mean of the library is built from (xor
 * butterfly inside each wave, then (w0 + w1) + (w2 + w3); pls_block_means adds the chunks of a block in ascending order). */
int pls_chunk_sums(const double *e, int64_t j, double *out, void *stream);

/* Cholesky factorisation on the device: K (M x M, SPD, row-major; only read) + jitter * I = Lc Lc^T.
 * Replaces the factorisation inside gpytorch.solve(lhs = k(Z,Z), ...) (inducing_point.py:89-93, :104-106, :130-132,
 * :235-239) and the per-step eigh(k(Z,Z)) of the noise sampler (samplers.py:27 via inducing_point.py:133-137).
 * Right-looking blocked algorithm, 64-column panels; the trailing updates run on the fp64 MFMA contraction.
 * Outputs (caller-allocated, M x M each, 16-byte aligned, even leading dimensions): Lc, LcT and -- unless NULL -- the
 * substitution operators Sf, Sb (see pls_chol_desc).  info (DEVICE int32): 0, or the 1-based index of the first
 * pivot that was not positive (K + jitter I is not numerically positive definite; the outputs are then garbage and the
 * caller retries with a larger jitter, as gpytorch's psd_safe_cholesky does).  Nothing synchronises. */
int pls_chol_factor(const double *K, int64_t ldk, int64_t m, double jitter, double *Lc, int64_t ldlc, double *LcT,
                    int64_t ldlct, double *Sf, int64_t ldsf, double *Sb, int64_t ldsb, int32_t *info, void *stream);

/* The substitution operators Sf, Sb (see pls_chol_desc) of a factor that is already on the device -- pls_chol_factor
 * ends with this call; parity runs upload the oracle's own LAPACK factor (Lc and its transpose, zeros in the other
 * triangle) and build the operators from it, so that both sides solve with the SAME factor.  Sf and Sb must be zero
 * outside the blocks this call writes (block upper / lower triangle). */
int pls_chol_build_operators(const double *Lc, int64_t ldlc, const double *LcT, int64_t ldlct, int64_t m, double *Sf,
                             int64_t ldsf, double *Sb, int64_t ldsb, void *stream);

/* V (M x J) = K^-1 U = Lc^-T Lc^-1 U: block forward then backward substitution in ONE launch (a workgroup owns 32
 * columns for the whole solve).  Replaces gpytorch.solve(lhs = k(Z,Z), input = k(Z,Z), rhs = U).  V must not alias U. */
int pls_chol_solve(const pls_chol_desc *factor, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv,
                   void *stream);

/* Linv = Lc^-1 and LinvT = Lc^-T (M x M each, 16-byte aligned, even leading dimensions): the identity pushed through the
 * block forward substitution of pls_chol_solve, column by column the backward-stable solve Lc x = e_c.  Needs Sf / Sb. */
int pls_chol_build_inverse(const pls_chol_desc *factor, double *Linv, int64_t ldlinv, double *LinvT, int64_t ldlinvt,
                           void *stream);

/* Y (M x J) = Lc^-1 U: the forward half of pls_chol_solve (one triangular product with LinvT when the descriptor has it
 * and PLS_OPT_SOLVE_MODE is 1, block forward substitution otherwise).  Y must not alias U. */
int pls_chol_forward_solve(const pls_chol_desc *factor, const double *U, int64_t ldu, int64_t j, double *Y, int64_t ldy,
                           void *stream);

/* Bytes of pls_chol_desc.tri_scratch / pls_ipb_desc.tri_scratch for products with M rows and up to J columns (16 KB of
 * flag words, then two 32 KB partial-sum slots per pair of 64-row tile rows and 64-column tile). */
size_t pls_tri_scratch_bytes(int64_t m, int64_t j);

/* pls_chol_solve with a workspace of pls_chol_solve_workspace_bytes(M, J): with Linv / LinvT in the descriptor (and
 * PLS_OPT_SOLVE_MODE 1) the solve is two triangular products, Lc^-1 U into the workspace and Lc^-T of that into V; a
 * narrow J-shard then uses every CU (M = 1024, J = 1024: 32 workgroups of the substitution kernel on 256 CUs). */
size_t pls_chol_solve_workspace_bytes(int64_t m, int64_t j);
int pls_chol_solve_ws(const pls_chol_desc *factor, const double *U, int64_t ldu, int64_t j, double *V, int64_t ldv,
                      void *workspace, size_t workspace_bytes, void *stream);

/* out (M x J) = Lc X with the TRANSPOSED factor LcT (upper) as the k-major operand; only k <= row is contracted.
 * Colours standard normals: e = Lc xi ~ N(0, k(Z,Z)) (replaces sample_multivariate_normal's Q sqrt(Lambda) xi,
 * samplers.py:37-44, same law). */
int pls_tri_multiply(const double *LcT, int64_t ldlct, int64_t m, const double *X, int64_t ldx, int64_t j, double *out,
                     int64_t ldo, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Orthonormal basis: setup + step
 * ------------------------------------------------------------------------------------------- */

/* A = Vs^T Kzx and At = Kzx^T Vs from the scaled eigenvectors Vs (M x Mk) and Kzx (M x N).
 * Replaces the per-step re-association `base_gram_induce_train.T @ scaled_eigenvectors`
 * (orthonormal.py:106-108) and `scaled_eigenvectors.T @ base_gram_induce_train` (:152-154) by a one-time build. */
int pls_onb_build_projection(const double *Vs, int64_t ldvs, const double *Kzx, int64_t ldkzx, int64_t m, int64_t mk,
                             int64_t n, double *A, int64_t lda, double *At, int64_t ldat, void *stream);

/* Gaussian/identity fast path constants: B = A A^T (Mk x Mk), c[0..Mk) = A y and c[Mk] = y^T y (c holds Mk + 1 doubles). */
int pls_onb_build_gaussian(const pls_onb_desc *basis, const double *y, double *B, int64_t ldb, double *c,
                           void *stream);

/* F(N x J) = A^T U.  Replaces OrthonormalBasis.calculate_untransformed_train_prediction_samples
 * (orthonormal.py:98-108). */
int pls_onb_forward(const pls_onb_desc *basis, const double *U, int64_t ldu, int64_t j, double *F, int64_t ldf,
                    void *stream);

/* dU(Mk x J) = -eta * A G - eta * diag(1/lam) U + sqrt(2 eta) * xi.
 * Replaces OrthonormalBasis._calculate_particle_update (orthonormal.py:128-159). */
int pls_onb_particle_update(const pls_onb_desc *basis, const double *U, int64_t ldu, const double *G, int64_t ldg,
                            int64_t j, double eta, const pls_noise_desc *noise, double *dU, int64_t lddu,
                            void *stream);

/* One fused Langevin step: PLS.calculate_particle_update(U, eta)
 * (projected_langevin_sampling.py:107-123 -> orthonormal.py:98-108 -> costs/{*}.py -> orthonormal.py:128-159).
 * Streams N in chunks: F chunk -> cost derivative in the GEMM epilogue -> back-projection; F and G are
 * never materialised beyond one chunk.  If basis->B/c are set and the cost is Gaussian/identity the
 * Mk x Mk x J fast path is taken unless force_generic != 0.
 * out_mode 0: out = dU (the reference's return value);  out_mode 1: out = U + dU (the caller's
 * `particles += update`, trainers.py:157, fused).  out must not alias U: other workgroups still read U as
 * the GEMM operand, so callers ping-pong two particle buffers.
 * energy_in (may be NULL): receives e_j(U) of the INPUT particles -- the value pls_onb_energy returns -- as a
 * by-product: on the fast path from the same B U product (it then needs 2 * cdiv(mk, 128) * j workspace doubles), otherwise
 * from the same F tile the cost derivative is taken of, so a train loop (trainers.py:153-159, which recomputes F for the
 * energy) pays no separate energy pass.
 * workspace: pls_onb_step_workspace_bytes(...) bytes (any larger size lets it use bigger N chunks). */
size_t pls_onb_step_workspace_bytes(const pls_onb_desc *basis, int64_t j, int64_t n_chunk);
int pls_onb_step(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                 int64_t j, double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode,
                 int32_t force_generic, double *energy_in, void *workspace, size_t workspace_bytes, void *stream);

/* out[b] = the sum of e[16 b .. min(j, 16 (b + 1))) in ascending order (pls_block_desc.energy_sums16 as a stand-alone launch). */
int pls_sums16(const double *e, int64_t j, double *out, void *stream);
/* Number of 32-bit counters behind pls_block_desc.step_sync for j particle columns. */
size_t pls_step_sync_words(int64_t j);
/* Bytes of pls_block_desc.energy_partials for a basis with `rows` functions (Mk, or M of the inducing-point basis) and j
 * particle columns. */
size_t pls_energy_partials_bytes(int64_t rows, int64_t j);

/* pls_onb_step with one step size PER COLUMN BLOCK (pls_block_desc): the batched step-size search. */
int pls_onb_step_blocks(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                        int64_t j, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                        int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace,
                        size_t workspace_bytes, void *stream);

/* e(J) = cost_j + 0.5 * sum_m U_mj^2 / lam_m  (per-particle energy; the caller takes the mean over all
 * particles of all ranks).  Replaces PLS.calculate_energy_potential -> OrthonormalBasis.calculate_energy_potential
 * (projected_langevin_sampling.py:125-138, orthonormal.py:110-126).
 * Gaussian/identity with basis->B/c set (and force_generic == 0): cost_j = (u_j^T B u_j - 2 c^T u_j + y^T y) / (2 sigma2)
 * from ONE Mk x Mk x J contraction instead of the N x Mk x J one. */
size_t pls_onb_energy_workspace_bytes(const pls_onb_desc *basis, int64_t j, int64_t n_chunk);
int pls_onb_energy(const pls_onb_desc *basis, const pls_cost_desc *cost, const double *y, const double *U,
                   int64_t ldu, int64_t j, double *e, int32_t force_generic, void *workspace, size_t workspace_bytes,
                   void *stream);

/* e(J) = cost_j + 0.5 * sum_m U_mj^2 / lam_m with the cost vector handed in (cost may be NULL = zeros).
 * Replaces OrthonormalBasis.calculate_energy_potential(particles, cost) (orthonormal.py:110-126) before its
 * .mean().item(). */
int pls_onb_prior_energy(const pls_onb_desc *basis, const double *U, int64_t ldu, int64_t j, const double *cost,
                         double *e, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Inducing-point basis: step (setup = pls_kernel_gram + pls_chol_factor)
 * ------------------------------------------------------------------------------------------- */

/* F = Kxz k(Z,Z)^-1 U (inducing_point.py:81-93). workspace: m*j doubles. */
int pls_ipb_forward(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, double *F, int64_t ldf,
                    void *workspace, size_t workspace_bytes, void *stream);

/* dU = -eta Kzx G - eta M k(Z,Z)^-1 U + sqrt(2 eta) e, e = Lc xi (inducing_point.py:117-150).
 * PLS_NOISE_INJECTED: noise->xi is used AS e (already coloured).
 * workspace: 4 * align256(m*j*8) bytes. */
int pls_ipb_particle_update(const pls_ipb_desc *basis, const double *U, int64_t ldu, const double *G, int64_t ldg,
                            int64_t j, double eta, const pls_noise_desc *noise, double *dU, int64_t lddu,
                            void *workspace, size_t workspace_bytes, void *stream);

size_t pls_ipb_step_workspace_bytes(const pls_ipb_desc *basis, int64_t j, int64_t n_chunk);
/* energy_in (optional, J doubles): receives the energy of the INPUT particles (cost of the same F the drift uses +
 * the prior term) as a by-product, like pls_onb_step.  If basis->B/c are set and the cost is Gaussian/identity the
 * M x M x J fast path is taken unless force_generic != 0.  Other costs on at most 128 inducing points take, while the
 * problem is launch-bound (PLS_OPT_SMALL_RANK_STEP), two launches: V = k(Z,Z)^-1 U with the coloured noise (csrc/ipb_prep.h,
 * PLS_OPT_IPB_PREP), then projection, cost, back-projection, prior drift, update and energies in one
 * (csrc/small_rank_step.h; pls_block_desc.step_sync / energy_sums16 apply as for pls_onb_step_blocks). */
int pls_ipb_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                 int64_t j, double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode,
                 int32_t force_generic, double *energy_in, void *workspace, size_t workspace_bytes, void *stream);

/* pls_ipb_step with one step size PER COLUMN BLOCK (pls_block_desc): the batched step-size search. */
int pls_ipb_step_blocks(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, double *U, int64_t ldu,
                        int64_t j, const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                        int32_t out_mode, int32_t force_generic, double *energy_in, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Gaussian/identity fast path constants of the inducing-point basis: B = Kzx Kxz (M x M), c[0..M) = Kzx y, c[M] = y^T y.
 * With V = K^-1 U the data drift Kzx (Kxz V - y) / sigma2 (inducing_point.py:117-150 with gaussian.py:86-88) is
 * (B V - c) / sigma2 and the cost (gaussian.py:63-73) the quadratic form (v^T B v - 2 c^T v + y^T y) / (2 sigma2). */
int pls_ipb_build_gaussian(const pls_ipb_desc *basis, const double *y, double *B, int64_t ldb, double *c, void *stream);

/* e(J) = cost_j + (M/2) * ||k(Z,Z)^-1 U_j||^2 (inducing_point.py:95-115). */
size_t pls_ipb_energy_workspace_bytes(const pls_ipb_desc *basis, int64_t j, int64_t n_chunk);
int pls_ipb_energy(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, const double *U,
                   int64_t ldu, int64_t j, double *e, int32_t force_generic, void *workspace, size_t workspace_bytes, void *stream);

/* Whitened coordinates S = Lc^-1 U of the inducing-point basis, Gaussian cost with the identity link.
 * With V = k(Z,Z)^-1 U = Lc^-T S the update of inducing_point.py:117-150 under gaussian.py:86-88,
 *     dU = -eta ((B V - c) / sigma2 + M V) + sqrt(2 eta) Lc xi,
 * is  dU = Lc dS,  dS = -eta (Q S - ct) + sqrt(2 eta) xi,  and the energy of inducing_point.py:95-115 is
 * S^T Q S / 2 - ct^T S + y^T y / (2 sigma2): the same law, the same noise xi, one M x M x J contraction per step.
 * pls_ipb_build_whitened: Q (M x M, 16-byte aligned, ldq even) and ct (M + 1 doubles) from the descriptor's B, c and
 * Cholesky operators for inv_noise = 1 / sigma2; the caller then stores Q, ldq, ct and q_inv_noise = inv_noise in the
 * descriptor.  workspace: pls_ipb_build_whitened_workspace_bytes(M). */
size_t pls_ipb_build_whitened_workspace_bytes(int64_t m);
int pls_ipb_build_whitened(const pls_ipb_desc *basis, double inv_noise, double *Q, int64_t ldq, double *ct, void *workspace,
                           size_t workspace_bytes, void *stream);
/* Pt (M x M, 16-byte aligned, ldpt even) = Lc^-T Q from the descriptor's Linv and Q (see pls_ipb_desc.Pt); rebuilt whenever
 * Q is. */
int pls_ipb_build_step_operator(const pls_ipb_desc *basis, double *Pt, int64_t ldpt, void *stream);
/* S = Lc^-1 U and back, U = Lc S (neither may alias its input). */
int pls_ipb_whiten(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, double *S, int64_t lds, void *stream);
int pls_ipb_unwhiten(const pls_ipb_desc *basis, const double *S, int64_t lds, int64_t j, double *U, int64_t ldu, void *stream);
/* One Langevin step of the whitened particles: out = dS (out_mode 0) or S + dS (out_mode 1); Philox noise draws the xi
 * of the reference's e = Lc xi (same counters as pls_ipb_step), injected noise is used as xi (NOT coloured).  energy_in
 * (optional): energy of the INPUT particles; it needs pls_ipb_whitened_workspace_bytes(basis, J) workspace bytes. */
size_t pls_ipb_whitened_workspace_bytes(const pls_ipb_desc *basis, int64_t j);
int pls_ipb_whitened_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *S, int64_t lds, int64_t j,
                          double eta, const pls_noise_desc *noise, double *out, int64_t ldo, int32_t out_mode, double *energy_in,
                          void *workspace, size_t workspace_bytes, void *stream);
/* The same for ANY cost, on at most 128 inducing points, in ONE launch (csrc/small_rank_step.h over pls_ipb_desc.Awa): dS =
 * -eta Awa^T g + sqrt(2 eta) xi with g = d cost / d f on the data rows and f on the prior rows -- inducing_point.py:117-150 in
 * whitened coordinates, no solve and no coloured noise per step (a training loop whitens once, unwhitens once).  y: the n targets.
 * blocks: step sizes, and step_sync / energy_sums / energy_sums16 as for pls_onb_step_blocks.  _applies: 1 if the descriptor,
 * the targets' alignment and the sizes allow it (else the caller stays in the original coordinates: pls_ipb_step). */
int pls_ipb_build_whitened_operand(const pls_ipb_desc *basis, double *Awa, int64_t ldawa, void *stream);
int pls_ipb_whitened_generic_applies(const pls_ipb_desc *basis, const double *y, int64_t j);
size_t pls_ipb_whitened_generic_workspace_bytes(const pls_ipb_desc *basis, int64_t j);
int pls_ipb_whitened_generic_step(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *y, const double *S,
                                  int64_t lds, int64_t j, double eta, const pls_block_desc *blocks, const pls_noise_desc *noise,
                                  double *out, int64_t ldo, int32_t out_mode, double *energy_in, void *workspace,
                                  size_t workspace_bytes, void *stream);
int pls_ipb_whitened_step_blocks(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *S, int64_t lds, int64_t j,
                                 const pls_block_desc *blocks, const pls_noise_desc *noise, double *out, int64_t ldo,
                                 int32_t out_mode, double *energy_in, void *workspace, size_t workspace_bytes, void *stream);
/* e(J) of whitened particles; workspace: pls_ipb_whitened_workspace_bytes(basis, J). */
int pls_ipb_whitened_energy(const pls_ipb_desc *basis, const pls_cost_desc *cost, const double *S, int64_t lds, int64_t j,
                            double *e, void *workspace, size_t workspace_bytes, void *stream);

/* e(J) = cost_j + (M/2) * ||k(Z,Z)^-1 U_j||^2 with the cost vector handed in (inducing_point.py:95-115).
 * workspace: m*j doubles. */
int pls_ipb_prior_energy(const pls_ipb_desc *basis, const double *U, int64_t ldu, int64_t j, const double *cost,
                         double *e, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Setup before the path: greedy conditional-variance inducing-point selection
 * ------------------------------------------------------------------------------------------- */

/* Picks m rows of x (N x D, row-major) by the partial pivoted Cholesky / greedy DPP-MAP rule of
 * src/inducing_point_selectors/conditional_variance.py:27-120 (the caller has already applied the reference's random
 * permutation, :58-61): start from argmax of the jittered kernel diagonal, then repeatedly
 *   e = (round20(k(X, x_j)) + jitter * [n == j] - c[:i, j] . c[:i, :]) / sqrt(d_j);  c[i, :] = e;  d = max(d - e^2, 0)
 * and take the largest remaining d among the points not chosen yet (:101-106); stop early once sum(d) < threshold
 * (:108-113).  No N x N Gram matrix is formed (the reference builds one just to read its diagonal, :64-69).
 * indices: m int64 (device), count: 1 int64 (device) = number of points selected (m unless the threshold stopped it).
 * Nothing synchronises: the pivot of every iteration stays on the device.  Ties are broken towards the smaller index.
 * workspace: pls_select_inducing_workspace_bytes(n, m) bytes. */
size_t pls_select_inducing_workspace_bytes(int64_t n, int64_t m);
int pls_select_inducing_conditional_variance(int32_t kernel_kind, const double *x, int64_t n, int64_t d,
                                             const double *lengthscale, double outputscale, int64_t m, double jitter,
                                             double threshold, int64_t *indices, int64_t *count, void *workspace,
                                             size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PLSHIP_H */
