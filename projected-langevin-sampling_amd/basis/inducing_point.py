"""Inducing-point basis (drop-in for src/projected_langevin_sampling/basis/inducing_point.py:11-240)."""
from __future__ import annotations

import torch

from .. import _chol
from .. import _lib as L
from .. import _ops
from ..kernel import PLSKernel, _dev
from ..samplers import sample_multivariate_normal
from .base import BlockSpec, NoiseSpec, PLSBasis, alloc_matrix
from .orthonormal import _rows_contiguous


class InducingPointBasis(PLSBasis):
    """Particles are function values at the M inducing points (inducing_point.py:23-50).

    The reference calls gpytorch.solve(k(Z,Z), .) twice per step and eigh(k(Z,Z)) once per step for the noise
    (:130-137).  Here k(Z,Z) is factorised ONCE on the device (pls_chol_factor: blocked fp64 Cholesky with gpytorch's
    psd_safe_cholesky jitter schedule, _chol.py); every per-step solve is a block forward + backward substitution with that
    factor (pls_chol_solve, one launch), and the noise is e = L_c xi (same distribution N(0, k(Z,Z)) as the reference's
    Q sqrt(Lambda) xi; injected noise is used as-is).

    ``cholesky_factor`` (extension, parity runs): a lower Cholesky factor of k(Z,Z) computed elsewhere -- the oracle's
    LAPACK factor -- so that both sides solve with the same factor (the analogue of ``spectrum=`` on the orthonormal
    basis).  ``explicit_inverse`` (extension, A/B runs): also form W = k(Z,Z)^-1 on the host, for
    pls_set_option(PLS_OPT_IPB_EXPLICIT_INVERSE, 1)."""

    def __init__(
        self,
        kernel: PLSKernel,
        x_induce: torch.Tensor,
        y_induce: torch.Tensor,
        x_train: torch.Tensor,
        additional_predictive_noise_distribution: torch.distributions.Distribution | None = None,
        cholesky_factor: torch.Tensor | None = None,
        explicit_inverse: bool = False,
    ):
        super().__init__(additional_predictive_noise_distribution=additional_predictive_noise_distribution)
        self.kernel = kernel
        self.x_induce = x_induce  # (M, D)
        self.y_induce = y_induce  # (M,)
        self.gram_induce = self.kernel.forward(x1=x_induce, x2=x_induce)  # r(Z,Z)  :38-40
        self.base_gram_induce = self.kernel.base_kernel(x1=x_induce, x2=x_induce)  # k(Z,Z)  :41-43
        self.base_gram_induce_train = self.kernel.base_kernel(x1=x_induce, x2=x_train)  # k(Z,X) (M,N) :44-46
        dev = self.base_gram_induce.device
        m, n = self.base_gram_induce_train.shape
        self._n = n
        if cholesky_factor is None:
            self._chol = _chol.cholesky_factor(self.base_gram_induce)  # what gpytorch.solve does, on the device
        else:
            self._chol = _chol.factor_from_host(cholesky_factor)
        # the inverse factor Lc^-1: every solve with k(Z,Z) becomes triangular products on the MFMA contraction, which
        # fill the chip for any number of particle columns (a rank of an 8-GPU run holds J / 8 of them)
        self._chol.build_inverse()
        self._W = None
        if explicit_inverse:  # A/B only: the contraction W U instead of the two triangular solves
            self._W = _dev(torch.cholesky_inverse(self._chol.Lc.cpu()))
        # k(X,Z) as its own k-major operand for the back-projection k(Z,X) G
        self._Kxz = alloc_matrix(n, m, dev)
        self._Kxz.copy_(self.base_gram_induce_train.T)
        self._B = None  # Gaussian fast path constants, keyed by the y they were built from
        self._c = None
        self._gauss_key = None
        self._Q = None  # ... and the same operator in whitened coordinates, keyed by (y, observation noise)
        self._Pt = None
        self._ct = None
        self._q_inv_noise = 0.0
        self._white_key = None
        self._Awa = None  # k(X,Z) Lc^-T over sqrt(M) Lc^-T: the forward operand of whitened particles with the prior as rows

    @property
    def approximation_dimension(self) -> int:
        return self.x_induce.shape[0]  # :52-58

    def _desc(self, with_gaussian: bool = False) -> L.IpbDesc:
        d = L.IpbDesc()
        d.m, d.n = self.approximation_dimension, self._n
        d.Kzx, d.ldkzx = self.base_gram_induce_train.data_ptr(), L.ld(self.base_gram_induce_train)
        d.Kxz, d.ldkxz = self._Kxz.data_ptr(), L.ld(self._Kxz)
        if self._W is not None:
            d.W, d.ldw = self._W.data_ptr(), L.ld(self._W)
        f = self._chol
        d.LcT, d.ldlct = f.LcT.data_ptr(), L.ld(f.LcT)
        d.Sf, d.ldsf = f.Sf.data_ptr(), L.ld(f.Sf)
        d.Sb, d.ldsb = f.Sb.data_ptr(), L.ld(f.Sb)
        if f.Linv is not None:
            d.Linv, d.ldlinv = f.Linv.data_ptr(), L.ld(f.Linv)
            d.LinvT, d.ldlinvt = f.LinvT.data_ptr(), L.ld(f.LinvT)
        sc = f.tri_scratch()  # balanced triangular products on narrow particle shards (pls_ipb_desc.tri_scratch)
        if sc is not None:
            d.tri_scratch, d.tri_scratch_bytes = sc.data_ptr(), sc.numel() * 8
        if self._Awa is not None:
            d.Awa, d.ldawa = self._Awa.data_ptr(), L.ld(self._Awa)
        if with_gaussian and self._B is not None:
            d.B, d.ldb, d.c = self._B.data_ptr(), L.ld(self._B), self._c.data_ptr()
            if self._Q is not None and self.whitened:
                d.Q, d.ldq, d.ct, d.q_inv_noise = self._Q.data_ptr(), L.ld(self._Q), self._ct.data_ptr(), self._q_inv_noise
                if self._Pt is not None:
                    d.Pt, d.ldpt = self._Pt.data_ptr(), L.ld(self._Pt)
        return d

    #: take the Gaussian/identity step in whitened coordinates (pls_ipb_build_whitened); False: the round-2 route
    #: (solve, B V, Lc xi, update) -- kept for A/B runs and tests
    whitened = True

    def prepare_gaussian(self, y_dev: torch.Tensor, observation_noise: float | None = None) -> None:
        """B = k(Z,X) k(X,Z), c = k(Z,X) y (pls_ipb_build_gaussian) and, for the cost's observation noise, the same
        operator in whitened coordinates: Q = Lc^-1 (B / sigma2 + M I) Lc^-T, c~ = Lc^-1 c / sigma2 (pls_ipb_build_whitened)."""
        key = (y_dev.data_ptr(), y_dev._version)
        lib = L.load()
        m = self.approximation_dimension
        if self._gauss_key != key:
            self._B = alloc_matrix(m, m, y_dev.device)
            self._c = torch.empty(m + 1, dtype=torch.float64, device=y_dev.device)
            L.check(
                lib.pls_ipb_build_gaussian(self._desc(), y_dev.data_ptr(), self._B.data_ptr(), L.ld(self._B), self._c.data_ptr(),
                                           L.stream_ptr()),
                "pls_ipb_build_gaussian",
            )
            self._gauss_key = key
            # the whitened operator was built from the previous y: drop it, or a later call without observation_noise would
            # leave a stale Q in the descriptor (the C side only compares q_inv_noise with the cost's 1 / sigma2)
            self._white_key = None
            self._Q = self._ct = self._Pt = None
            self._q_inv_noise = 0.0
        if observation_noise is None or not self.whitened:
            return
        inv_noise = 1.0 / float(observation_noise)
        wkey = (key, inv_noise)
        if self._white_key == wkey:
            return
        q = alloc_matrix(m, m, y_dev.device)
        ct = torch.empty(m + 1, dtype=torch.float64, device=y_dev.device)
        ws_bytes = lib.pls_ipb_build_whitened_workspace_bytes(m)
        ws = torch.empty(ws_bytes // 8 + 1, dtype=torch.float64, device=y_dev.device)
        self._Q = self._Pt = None  # (the descriptor of the build call must not carry a stale operator)
        L.check(
            lib.pls_ipb_build_whitened(self._desc(with_gaussian=True), inv_noise, q.data_ptr(), L.ld(q), ct.data_ptr(),
                                       ws.data_ptr(), ws_bytes, L.stream_ptr()),
            "pls_ipb_build_whitened",
        )
        self._Q, self._ct, self._q_inv_noise, self._white_key = q, ct, inv_noise, wkey
        # ... and with the forward solve folded in: P^T = Lc^-T Q, so the per-call step computes dS from U itself
        self._Pt = None
        if self._chol.Linv is not None:
            pt = alloc_matrix(m, m, y_dev.device)
            L.check(lib.pls_ipb_build_step_operator(self._desc(with_gaussian=True), pt.data_ptr(), L.ld(pt), L.stream_ptr()),
                    "pls_ipb_build_step_operator")
            self._Pt = pt

    def _prepare_for(self, cost) -> None:
        self.prepare_gaussian(cost.y_device(), float(cost.desc().p[0]))

    # ---- whitened coordinates S = Lc^-1 U (Gaussian cost, identity link) ------------------------------------------------
    def whiten(self, particles: torch.Tensor) -> torch.Tensor:
        """S = Lc^-1 U (pls_ipb_whiten)."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        s = torch.empty_like(u, memory_format=torch.contiguous_format)
        if u.shape[1]:
            L.check(L.load().pls_ipb_whiten(self._desc(), u.data_ptr(), L.ld(u), u.shape[1], s.data_ptr(), L.ld(s), L.stream_ptr()),
                    "pls_ipb_whiten")
        return s

    def unwhiten(self, whitened: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """U = Lc S (pls_ipb_unwhiten)."""
        s = _rows_contiguous(L.require_gpu_tensor(whitened, "whitened particles"))
        u = torch.empty_like(s, memory_format=torch.contiguous_format) if out is None else L.require_gpu_tensor(out, "out")
        assert u.shape == s.shape, f"out has shape {tuple(u.shape)}, the whitened particles {tuple(s.shape)}"
        if s.shape[1]:
            L.check(L.load().pls_ipb_unwhiten(self._desc(), s.data_ptr(), L.ld(s), s.shape[1], u.data_ptr(), L.ld(u), L.stream_ptr()),
                    "pls_ipb_unwhiten")
        return u

    def whitened_step(self, cost, whitened: torch.Tensor, step_size: float, out: torch.Tensor | None = None,
                      new_state: bool = False, noise: NoiseSpec | None = None, input_energy: torch.Tensor | None = None,
                      blocks: BlockSpec | None = None, workspace: torch.Tensor | None = None) -> torch.Tensor:
        """One Langevin step of whitened particles (pls_ipb_whitened_step): ONE M x M x J contraction with the update,
        the noise and -- optionally -- the energy of the input particles in its epilogue.  ``noise`` injected = xi itself
        (standard normal, not coloured); Philox noise draws the xi of fused_step's e = Lc xi."""
        s = _rows_contiguous(L.require_gpu_tensor(whitened, "whitened particles"))
        j = s.shape[1]
        if out is None:
            out = torch.empty_like(s, memory_format=torch.contiguous_format)
        else:  # (written as float64 through a raw pointer: a buffer of another dtype or shape must never get this far)
            L.require_gpu_tensor(out, "out")
            assert out.shape == s.shape, f"out has shape {tuple(out.shape)}, the particles {tuple(s.shape)}"
        if j == 0:
            return out
        assert out.data_ptr() != s.data_ptr(), "whitened_step: out must not alias its input"
        assert self.whitened, "whitened coordinates are switched off on this basis"
        if not self._is_gaussian(cost, False):  # any other cost: the one-launch small-rank step over Awa (at most 128 points)
            return self._whitened_generic_step(cost, s, step_size, out, new_state, noise, input_energy, blocks, workspace)
        self._prepare_for(cost)
        lib = L.load()
        desc = self._desc(with_gaussian=True)
        ws, ws_bytes = None, 0
        if input_energy is not None:
            ws_bytes = lib.pls_ipb_whitened_workspace_bytes(desc, j)
            ws = self._pick_workspace(workspace, ws_bytes, s.device)
        nd = (noise if noise is not None else self._draw_noise_spec(None)).desc()
        mode = L.OUT_NEW_STATE if new_state else L.OUT_DELTA
        if blocks is None:
            L.check(lib.pls_ipb_whitened_step(desc, cost.desc(), s.data_ptr(), L.ld(s), j, float(step_size), nd, out.data_ptr(),
                                              L.ld(out), mode, L.ptr(input_energy), L.ptr(ws), ws_bytes, L.stream_ptr()),
                    "pls_ipb_whitened_step")
        else:
            L.check(lib.pls_ipb_whitened_step_blocks(desc, cost.desc(), s.data_ptr(), L.ld(s), j, blocks.desc(), nd,
                                                     out.data_ptr(), L.ld(out), mode, L.ptr(input_energy), L.ptr(ws), ws_bytes,
                                                     L.stream_ptr()), "pls_ipb_whitened_step_blocks")
        return out

    # ---- whitened coordinates for the costs WITHOUT the Gaussian algebra (at most 128 inducing points, launch-bound sizes) ----
    def whitened_generic_applies(self, cost, j: int) -> bool:
        """True if a loop over `j` particle columns may keep this cost's particles whitened: every step is then ONE launch
        (pls_ipb_whitened_generic_step: the one-launch small-rank step over k(X,Z) Lc^-T with the prior as rows) instead of the
        solve, the coloured noise and the step.  The operand is built on first use."""
        if not (self.whitened and cost.is_native()) or self._is_gaussian(cost, False):
            return False
        if not (1 <= self.approximation_dimension <= self.SMALL_RANK_MAX) or self._chol.Linv is None or j <= 0:
            return False
        m, n = self.approximation_dimension, self._n
        if j > 4096 or 4.0 * (n + m) * m * j > 8e9:  # (the launch-bound window of the C side: do not build the operand for nothing)
            return False
        if self._Awa is None:
            awa = alloc_matrix(n + m, m, self._Kxz.device)
            L.check(L.load().pls_ipb_build_whitened_operand(self._desc(), awa.data_ptr(), L.ld(awa), L.stream_ptr()),
                    "pls_ipb_build_whitened_operand")
            self._Awa = awa
        return bool(L.load().pls_ipb_whitened_generic_applies(self._desc(), cost.y_device().data_ptr(), int(j)))

    def _whitened_generic_call(self, cost, j: int, device):
        lib = L.load()
        desc = self._desc()
        ws_bytes = int(lib.pls_ipb_whitened_generic_workspace_bytes(desc, j))
        return lib, desc, ws_bytes

    def whitened_generic_sums_step_launcher(self, cost, state: torch.Tensor, eta: torch.Tensor):
        """(see OrthonormalBasis.sums_step_launcher) -- pls_ipb_whitened_generic_step on WHITENED particles, pre-bound"""
        s = _rows_contiguous(L.require_gpu_tensor(state, "whitened particles"))
        j = s.shape[1]
        assert self.whitened_generic_applies(cost, j)
        lib, desc, ws_bytes = self._whitened_generic_call(cost, j, s.device)
        cd, y = cost.desc(), cost.y_device()
        ws = torch.empty((ws_bytes + 7) // 8 + 1, dtype=torch.float64, device=s.device)
        sync = torch.zeros(max(int(lib.pls_step_sync_words(j)), 1), dtype=torch.int32, device=s.device)
        blocks, nd = L.BlockDesc(), L.NoiseDesc()
        blocks.block_cols, blocks.eta = j, L.require_gpu_tensor(eta, "eta").data_ptr()
        blocks.step_sync = sync.data_ptr()
        nd.kind, nd.step, nd.j_offset = L.NOISE_PHILOX, 0, int(self.j_offset)
        fn = lib.pls_ipb_whitened_generic_step
        y_ptr, ws_ptr, stream, mode = y.data_ptr(), ws.data_ptr(), L.stream_ptr(), L.OUT_NEW_STATE

        def launch(s_ptr, lds, out_ptr, ldo, seed, energy_ptr, sums_ptr):
            nd.seed = seed
            blocks.energy_sums16 = sums_ptr
            rc = fn(desc, cd, y_ptr, s_ptr, lds, j, 0.0, blocks, nd, out_ptr, ldo, mode, energy_ptr, ws_ptr, ws_bytes, stream)
            if rc:
                L.check(rc, "pls_ipb_whitened_generic_step")

        launch.keep_alive = (desc, cd, y, ws, sync, eta, self)
        return launch

    def _whitened_generic_step(self, cost, s, step_size, out, new_state, noise, input_energy, blocks, workspace):
        j = s.shape[1]
        assert self.whitened_generic_applies(cost, j), "this cost / size has no whitened step: stay in the original coordinates"
        lib, desc, ws_bytes = self._whitened_generic_call(cost, j, s.device)
        ws = self._pick_workspace(workspace, ws_bytes, s.device)
        nd = (noise if noise is not None else self._draw_noise_spec(None)).desc()
        bd = None if blocks is None else blocks.desc()
        if bd is None:
            bd = L.BlockDesc()
            bd.block_cols, bd.eta = j, self._eta_word(step_size, s.device).data_ptr()
        if not bd.step_sync:
            bd.step_sync = self._step_sync(j, s.device).data_ptr()
        try:
            L.check(lib.pls_ipb_whitened_generic_step(desc, cost.desc(), cost.y_device().data_ptr(), s.data_ptr(), L.ld(s), j, 0.0, bd,
                                                      nd, out.data_ptr(), L.ld(out), L.OUT_NEW_STATE if new_state else L.OUT_DELTA,
                                                      L.ptr(input_energy), ws.data_ptr(), ws_bytes, L.stream_ptr()),
                    "pls_ipb_whitened_generic_step")
        except L.PlsHipError:
            self.zero_step_sync()
            raise
        return out

    def supports_lagged_energies(self, cost) -> bool:
        """(see OrthonormalBasis.supports_lagged_energies) -- for loops that stay in whitened coordinates: the Gaussian/identity
        route only (the one-launch small-rank step finishes its energies itself)"""
        return bool(cost.is_native()) and self.whitened and self._is_gaussian(cost, False)

    def energy_partial_rows_bytes(self, j: int) -> int:
        return int(L.load().pls_energy_partials_bytes(self.approximation_dimension, j))

    def lagged_step_launcher(self, cost, state: torch.Tensor, eta: torch.Tensor):
        """(see OrthonormalBasis.lagged_step_launcher) -- the whitened step, pls_ipb_whitened_step_blocks, pre-bound"""
        s = _rows_contiguous(L.require_gpu_tensor(state, "whitened particles"))
        assert self.whitened and self._is_gaussian(cost, False)
        self._prepare_for(cost)
        fn = L.load().pls_ipb_whitened_step_blocks
        desc, cd = self._desc(with_gaussian=True), cost.desc()
        blocks, nd = L.BlockDesc(), L.NoiseDesc()
        j = s.shape[1]
        blocks.block_cols, blocks.eta = j, L.require_gpu_tensor(eta, "eta").data_ptr()
        nd.kind, nd.step, nd.j_offset = L.NOISE_PHILOX, 0, int(self.j_offset)
        stream, mode = L.stream_ptr(), L.OUT_NEW_STATE

        def launch(s_ptr, lds, out_ptr, ldo, seed, partials_out, partials_prev, energy_prev, sums_prev):
            nd.seed = seed
            blocks.energy_partials, blocks.energy_partials_prev = partials_out, partials_prev
            blocks.energy_prev, blocks.energy_sums_prev = energy_prev, sums_prev
            rc = fn(desc, cd, s_ptr, lds, j, blocks, nd, out_ptr, ldo, mode, None, None, 0, stream)
            if rc:
                L.check(rc, "pls_ipb_whitened_step_blocks")

        launch.keep_alive = (desc, cd, eta, self)
        return launch

    def flush_energies(self, cost, state: torch.Tensor, blocks: BlockSpec) -> None:
        """(see OrthonormalBasis.flush_energies); ``state`` = the WHITENED particle matrix of the whitened_step calls"""
        s = _rows_contiguous(L.require_gpu_tensor(state, "whitened particles"))
        self._prepare_for(cost)
        nd = NoiseSpec(none=True).desc()
        L.check(
            L.load().pls_ipb_whitened_step_blocks(self._desc(with_gaussian=True), cost.desc(), s.data_ptr(), L.ld(s), s.shape[1],
                                                  blocks.desc(), nd, None, 0, L.OUT_DELTA, None, None, 0, L.stream_ptr()),
            "pls_ipb_whitened_step_blocks",
        )

    def whitened_particle_energy(self, cost, whitened: torch.Tensor) -> torch.Tensor:
        """e_j of whitened particles: S^T Q S / 2 - c~^T S + y^T y / (2 sigma2) (pls_ipb_whitened_energy)."""
        s = _rows_contiguous(L.require_gpu_tensor(whitened, "whitened particles"))
        j = s.shape[1]
        if not self._is_gaussian(cost, False):
            # (plain loops and final values only: a step with step size zero and no noise, for its energy by-product)
            e = torch.empty(j, dtype=torch.float64, device=s.device)
            if j:
                scratch = torch.empty_like(s, memory_format=torch.contiguous_format)
                self._whitened_generic_step(cost, s, 0.0, scratch, False, NoiseSpec(none=True), e, None, None)
            return e
        self._prepare_for(cost)
        lib = L.load()
        desc = self._desc(with_gaussian=True)
        ws_bytes = lib.pls_ipb_whitened_workspace_bytes(desc, j)
        ws = self._workspace(ws_bytes, s.device)
        e = torch.empty(j, dtype=torch.float64, device=s.device)
        if j:
            L.check(lib.pls_ipb_whitened_energy(desc, cost.desc(), s.data_ptr(), L.ld(s), j, e.data_ptr(), ws.data_ptr(), ws_bytes,
                                                L.stream_ptr()), "pls_ipb_whitened_energy")
        return e

    @staticmethod
    def _is_gaussian(cost, force_generic: bool) -> bool:
        cd = cost.desc()
        return cd.cost == L.COST_GAUSSIAN and cd.link == L.LINK_IDENTITY and not force_generic

    def _initialise_particles(self, number_of_particles: int, noise_only: bool = True, seed: int | None = None) -> torch.Tensor:
        particle_noise = self._initialise_particles_noise(number_of_particles=number_of_particles, seed=seed)
        return particle_noise if noise_only else (self.y_induce.cpu()[:, None] + particle_noise)  # :77-79

    def calculate_untransformed_train_prediction_samples(self, particles: torch.Tensor) -> torch.Tensor:
        """F = k(X,Z) k(Z,Z)^-1 U  (:81-93)."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        j = u.shape[1]
        m = self.approximation_dimension
        f = torch.empty((self._n, j), dtype=torch.float64, device=u.device)
        ws = self._workspace(m * j * 8, u.device)
        L.check(
            L.load().pls_ipb_forward(self._desc(), u.data_ptr(), L.ld(u), j, f.data_ptr(), max(j, 1), ws.data_ptr(),
                                     ws.numel() * 8, L.stream_ptr()),
            "pls_ipb_forward",
        )
        return f

    def particle_energy_potential(self, particles: torch.Tensor, cost: torch.Tensor | None) -> torch.Tensor:
        """e_j = cost_j + M/2 ||k(Z,Z)^-1 U_j||^2  (:95-114)."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        j = u.shape[1]
        c = None if cost is None else L.require_gpu_tensor(cost, "cost", promote=True).contiguous()
        e = torch.empty(j, dtype=torch.float64, device=u.device)
        ws = self._workspace(self.approximation_dimension * j * 8, u.device)
        L.check(
            L.load().pls_ipb_prior_energy(self._desc(), u.data_ptr(), L.ld(u), j, L.ptr(c), e.data_ptr(), ws.data_ptr(),
                                          ws.numel() * 8, L.stream_ptr()),
            "pls_ipb_prior_energy",
        )
        return e

    def calculate_energy_potential(self, particles: torch.Tensor, cost: torch.Tensor) -> float:
        return _ops.block_means(self.particle_energy_potential(particles, cost)).item()  # :115

    def _calculate_particle_update(self, particles: torch.Tensor, cost_derivative: torch.Tensor, step_size: float,
                                   noise: torch.Tensor | None = None) -> torch.Tensor:
        """dU = -eta k(Z,X) G - eta M k(Z,Z)^-1 U + sqrt(2 eta) e  (:117-150); ``noise`` is e itself."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        g = _rows_contiguous(L.require_gpu_tensor(cost_derivative, "cost_derivative", promote=True))
        j = u.shape[1]
        assert g.shape == (self._n, j), f"cost_derivative has shape {tuple(g.shape)}, expected ({self._n}, {j})"
        du = torch.empty_like(u, memory_format=torch.contiguous_format)
        m = self.approximation_dimension
        ws_bytes = 4 * ((m * j * 8 + 255) // 256 * 256)
        ws = self._workspace(ws_bytes, u.device)
        nd = self._draw_noise_spec(noise).desc()
        L.check(
            L.load().pls_ipb_particle_update(self._desc(), u.data_ptr(), L.ld(u), g.data_ptr(), L.ld(g), j, float(step_size), nd,
                                             du.data_ptr(), L.ld(du), ws.data_ptr(), ws_bytes, L.stream_ptr()),
            "pls_ipb_particle_update",
        )
        return du

    def supports_fused_step(self) -> bool:
        return True

    def fused_step(self, cost, particles: torch.Tensor, step_size: float, out: torch.Tensor | None = None,
                   new_state: bool = False, noise: NoiseSpec | None = None, force_generic: bool = False,
                   input_energy: torch.Tensor | None = None, blocks: BlockSpec | None = None,
                   workspace: torch.Tensor | None = None) -> torch.Tensor:
        """One whole Langevin step (pls_ipb_step).  ``input_energy`` (J,) receives the per-particle energy of
        ``particles`` as a by-product (cost of the same F + (M/2)||K^-1 U||^2).  ``blocks``: one step size per column
        block (pls_ipb_step_blocks; ``step_size`` is then ignored).  ``workspace``: a caller-owned buffer (graph captures)."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        j = u.shape[1]
        if out is None:
            out = torch.empty_like(u, memory_format=torch.contiguous_format)
        else:  # (written as float64 through a raw pointer: a buffer of another dtype or shape must never get this far)
            L.require_gpu_tensor(out, "out")
            assert out.shape == u.shape, f"out has shape {tuple(out.shape)}, the particles {tuple(u.shape)}"
        if j == 0:
            return out
        assert out.data_ptr() != u.data_ptr(), "fused_step: out must not alias particles"
        lib = L.load()
        gaussian = self._is_gaussian(cost, force_generic)
        if gaussian:
            self._prepare_for(cost)
        desc = self._desc(with_gaussian=gaussian)
        need_min = lib.pls_ipb_step_workspace_bytes(desc, j, 128)
        need_full = lib.pls_ipb_step_workspace_bytes(desc, j, self._n)
        ws_bytes = max(need_min, min(need_full, self.workspace_bytes))
        ws = self._pick_workspace(workspace, ws_bytes, u.device)
        nd = (noise if noise is not None else self._draw_noise_spec(None)).desc()
        mode = L.OUT_NEW_STATE if new_state else L.OUT_DELTA
        bd = None if blocks is None else blocks.desc()
        if not gaussian and 1 <= self.approximation_dimension <= self.SMALL_RANK_MAX:
            # (the one-launch small-rank step meets through zeroed counters: see OrthonormalBasis.fused_step)
            if bd is None:
                bd = L.BlockDesc()
                bd.block_cols, bd.eta = j, self._eta_word(step_size, u.device).data_ptr()
            if not bd.step_sync:
                bd.step_sync = self._step_sync(j, u.device).data_ptr()
        try:
            if bd is None:
                L.check(
                    lib.pls_ipb_step(desc, cost.desc(), cost.y_device().data_ptr(), u.data_ptr(), L.ld(u), j, float(step_size), nd,
                                     out.data_ptr(), L.ld(out), mode, 1 if force_generic else 0, L.ptr(input_energy), ws.data_ptr(),
                                     ws_bytes, L.stream_ptr()),
                    "pls_ipb_step",
                )
            else:
                L.check(
                    lib.pls_ipb_step_blocks(desc, cost.desc(), cost.y_device().data_ptr(), u.data_ptr(), L.ld(u), j, bd, nd,
                                            out.data_ptr(), L.ld(out), mode, 1 if force_generic else 0, L.ptr(input_energy),
                                            ws.data_ptr(), ws_bytes, L.stream_ptr()),
                    "pls_ipb_step_blocks",
                )
        except L.PlsHipError:
            self.zero_step_sync()
            raise
        return out

    def step_launcher(self, cost, state: torch.Tensor, step_size: float):
        """(see OrthonormalBasis.step_launcher) -- pls_ipb_step with the energy by-product + pls_block_means, pre-bound for a
        training loop that stays in the original coordinates (any cost but the Gaussian fast path's whitened loop)"""
        u = _rows_contiguous(L.require_gpu_tensor(state, "particles"))
        j = u.shape[1]
        lib = L.load()
        gaussian = self._is_gaussian(cost, False)
        if gaussian:
            self._prepare_for(cost)
        desc, cd, y = self._desc(with_gaussian=gaussian), cost.desc(), cost.y_device()
        ws_bytes = max(lib.pls_ipb_step_workspace_bytes(desc, j, 128), min(lib.pls_ipb_step_workspace_bytes(desc, j, self._n), self.workspace_bytes))
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=u.device)
        nd = L.NoiseDesc()
        nd.kind, nd.step, nd.j_offset = L.NOISE_PHILOX, 0, int(self.j_offset)
        step, means = lib.pls_ipb_step, lib.pls_block_means
        y_ptr, ws_ptr, stream, mode, eta = y.data_ptr(), ws.data_ptr(), L.stream_ptr(), L.OUT_NEW_STATE, float(step_size)

        def launch(u_ptr, ldu, out_ptr, ldo, seed, energy_ptr, mean_ptr):
            nd.seed = seed
            rc = step(desc, cd, y_ptr, u_ptr, ldu, j, eta, nd, out_ptr, ldo, mode, 0, energy_ptr, ws_ptr, ws_bytes, stream)
            if rc:
                L.check(rc, "pls_ipb_step")
            rc = means(energy_ptr, j, j, mean_ptr, stream)
            if rc:
                L.check(rc, "pls_block_means")

        launch.keep_alive = (desc, cd, y, ws, self)
        return launch

    def step_workspace_bytes(self, cost, j: int, with_energy: bool, force_generic: bool = False) -> int:
        """Bytes fused_step asks of its workspace for ``j`` columns (graph captures allocate their own buffer)."""
        lib = L.load()
        desc = self._desc(with_gaussian=self._is_gaussian(cost, force_generic) and self._B is not None)
        need_min = lib.pls_ipb_step_workspace_bytes(desc, j, 128)
        need_full = lib.pls_ipb_step_workspace_bytes(desc, j, self._n)
        return max(need_min, min(need_full, self.workspace_bytes))

    def supports_input_energy(self, cost) -> bool:
        return bool(cost.is_native())

    #: inducing-point counts up to which a cost without the Gaussian algebra takes the small-rank kernels
    SMALL_RANK_MAX = 128

    def supports_energy_sums(self, cost) -> bool:
        """the whitened Gaussian/identity route, and every other native cost on at most 128 inducing points (the one-launch
        small-rank step writes the sums itself; see OrthonormalBasis.supports_energy_sums)"""
        if not cost.is_native():
            return False
        return (self.whitened and self._is_gaussian(cost, False)) or self._one_launch_rank(cost)

    def _one_launch_rank(self, cost) -> bool:
        return (not self._is_gaussian(cost, False)) and 1 <= self.approximation_dimension <= self.SMALL_RANK_MAX

    def uses_sums16(self, cost) -> bool:
        """(see OrthonormalBasis.uses_sums16)"""
        return bool(cost.is_native()) and self._one_launch_rank(cost)

    def sums_step_launcher(self, cost, state: torch.Tensor, eta: torch.Tensor):
        """(see OrthonormalBasis.sums_step_launcher) -- pls_ipb_step_blocks with the energies of the input particles and their
        16-column sums, pre-bound: V = k(Z,Z)^-1 U and the coloured noise by their own launches, everything else of the step
        in ONE (csrc/small_rank_step.h) while the problem is launch-bound.  None for the Gaussian fast path."""
        if not (cost.is_native() and self._one_launch_rank(cost)):
            return None
        u = _rows_contiguous(L.require_gpu_tensor(state, "particles"))
        j = u.shape[1]
        lib = L.load()
        desc, cd, y = self._desc(with_gaussian=False), cost.desc(), cost.y_device()
        ws_bytes = max(lib.pls_ipb_step_workspace_bytes(desc, j, 128), min(lib.pls_ipb_step_workspace_bytes(desc, j, self._n), self.workspace_bytes))
        ws = torch.empty((ws_bytes + 7) // 8, dtype=torch.float64, device=u.device)
        sync = torch.zeros(max(int(lib.pls_step_sync_words(j)), 1), dtype=torch.int32, device=u.device)
        blocks, nd = L.BlockDesc(), L.NoiseDesc()
        blocks.block_cols, blocks.eta = j, L.require_gpu_tensor(eta, "eta").data_ptr()
        blocks.step_sync = sync.data_ptr()
        nd.kind, nd.step, nd.j_offset = L.NOISE_PHILOX, 0, int(self.j_offset)
        fn = lib.pls_ipb_step_blocks
        y_ptr, ws_ptr, stream, mode = y.data_ptr(), ws.data_ptr(), L.stream_ptr(), L.OUT_NEW_STATE

        def launch(u_ptr, ldu, out_ptr, ldo, seed, energy_ptr, sums_ptr):
            nd.seed = seed
            blocks.energy_sums16 = sums_ptr
            rc = fn(desc, cd, y_ptr, u_ptr, ldu, j, blocks, nd, out_ptr, ldo, mode, 0, energy_ptr, ws_ptr, ws_bytes, stream)
            if rc:
                L.check(rc, "pls_ipb_step_blocks")

        launch.keep_alive = (desc, cd, y, ws, sync, eta, self)
        return launch

    def fused_particle_energy(self, cost, particles: torch.Tensor, force_generic: bool = False) -> torch.Tensor:
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        j = u.shape[1]
        lib = L.load()
        gaussian = self._is_gaussian(cost, force_generic)
        if gaussian:
            self._prepare_for(cost)
        desc = self._desc(with_gaussian=gaussian)
        ws_bytes = lib.pls_ipb_energy_workspace_bytes(desc, j, self._n)
        ws = self._workspace(ws_bytes, u.device)
        e = torch.empty(j, dtype=torch.float64, device=u.device)
        L.check(
            lib.pls_ipb_energy(desc, cost.desc(), cost.y_device().data_ptr(), u.data_ptr(), L.ld(u), j, e.data_ptr(),
                               1 if force_generic else 0, ws.data_ptr(), ws_bytes, L.stream_ptr()),
            "pls_ipb_energy",
        )
        return e

    # ---- prediction ---------------------------------------------------------------------------------
    def sample_predictive_noise(self, particles: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """G([Z, x]) ~ N(0, r([Z,x],[Z,x]))  (:152-202); device normals keyed by the global particle column, the sampler's
        eigh remembered per test-point tensor (see OrthonormalBasis.sample_predictive_noise)."""
        from .orthonormal import _cached_factor

        lt = _cached_factor(self, x, self._predictive_covariance)
        predictive_noise = sample_multivariate_normal(
            mean=torch.zeros(lt.shape[0]), cov=None, size=(particles.shape[1],), factor=lt, j_offset=self.j_offset
        ).T
        if self.additional_predictive_noise_distribution is not None:
            extra = self.additional_predictive_noise_distribution.sample(predictive_noise.shape).reshape(predictive_noise.shape)
            predictive_noise = predictive_noise + _dev(extra)
        return predictive_noise.contiguous()

    def _predictive_covariance(self, x: torch.Tensor) -> torch.Tensor:
        gram_x = self.kernel.forward(x1=x, x2=x, additional_approximation_samples=x)
        gram_induce_x = self.kernel.forward(x1=self.x_induce, x2=x, additional_approximation_samples=x)
        return torch.cat(
            [torch.cat([self.gram_induce, gram_induce_x], dim=1), torch.cat([gram_induce_x.T, gram_x], dim=1)], dim=0
        )

    def predict_untransformed_samples(self, particles: torch.Tensor, x: torch.Tensor,
                                      noise: torch.Tensor | None = None) -> torch.Tensor:
        """G(x) + r(x,Z) r(Z,Z)^-1 (U - G(Z))  (:204-240); r(Z,Z) is factorised on the device once per call."""
        u = _rows_contiguous(L.require_gpu_tensor(particles, "particles", promote=True))
        gram_induce_x = self.kernel.forward(x1=self.x_induce, x2=x, additional_approximation_samples=x)  # r(Z,x) (M,N*)
        gram_induce = self.kernel.forward(x1=self.x_induce, x2=self.x_induce, additional_approximation_samples=x)
        if noise is None:
            noise = self.sample_predictive_noise(particles=particles, x=x)
        noise = L.require_gpu_tensor(noise, "noise", promote=True)
        m = self.approximation_dimension
        # r(Z,Z)^-1 r(Z,x): gpytorch.solve(lhs=r(x,Z), input=r(Z,Z), rhs=...) at :235-239 -- a psd-safe Cholesky of r(Z,Z)
        # (its condition number is the SQUARE of k(Z,Z)'s: the jitter schedule matters here) and two triangular solves
        q = _chol.cholesky_factor(gram_induce).solve(gram_induce_x)  # (M, N*)
        out = noise[m:, :].contiguous().clone()
        _ops.gemm_tn(q, u, 1.0, 1.0, out=out)
        _ops.gemm_tn(q, noise[:m, :].contiguous(), -1.0, 1.0, out=out)
        return out
