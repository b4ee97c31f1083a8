"""Basis base class (drop-in for src/projected_langevin_sampling/basis/base.py:7-193)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch

from .. import _lib as L
from ..kernel import _dev

#: default cap of the per-step G-chunk workspace (bytes); bigger = fewer, larger GEMM launches
DEFAULT_WORKSPACE_BYTES = 2 << 30


#: bit pattern a consumer writes into a slot of BlockSpec.energy_sums before it queues the launch that fills it: a quiet NaN
#: with a payload no computation produces (a diverged run's NaN / inf energies are ordinary values next to it)
UNWRITTEN_ENERGY_BITS = 0x7FF8DEADBEEF0001


class NoiseSpec:
    """How one Langevin step gets its noise.
    injected: a (M, J) device tensor used as-is (parity runs inject the oracle's noise);
    philox:   libplship's counter-based stream keyed by (seed, step, row, j_offset + column)."""

    def __init__(self, injected: torch.Tensor | None = None, seed: int | None = None, step: int = 0, j_offset: int = 0,
                 none: bool = False, step_base: torch.Tensor | None = None):
        self.injected, self.seed, self.step, self.j_offset, self.none = injected, seed, step, j_offset, none
        #: optional device int64 scalar added to ``step`` at kernel run time (lets a captured graph draw fresh noise)
        self.step_base = step_base

    def desc(self) -> L.NoiseDesc:
        d = L.NoiseDesc()
        if self.none:
            d.kind = L.NOISE_NONE
        elif self.injected is not None:
            xi = L.require_gpu_tensor(self.injected, "noise", promote=True)
            assert xi.stride(-1) == 1
            d.kind, d.xi, d.ldxi = L.NOISE_INJECTED, xi.data_ptr(), L.ld(xi)
        else:
            d.kind, d.seed, d.step, d.j_offset = L.NOISE_PHILOX, int(self.seed) & (2**64 - 1), int(self.step), int(self.j_offset)
            if self.step_base is not None:
                assert self.step_base.device.type == "cuda" and self.step_base.dtype == torch.int64 and self.step_base.numel() == 1
                d.step_base = self.step_base.data_ptr()
        return d


class BlockSpec:
    """One step size per column block of the particle matrix (pls_block_desc): the S candidates of a step-size search
    run as S blocks of ``block_cols`` columns.  ``eta`` is a device float64 vector, one entry per block; writing 0
    freezes a block."""

    def __init__(self, block_cols: int, eta: torch.Tensor, energy_sums: int | None = None,
                 energy_sync: torch.Tensor | None = None, energy_partials: torch.Tensor | None = None,
                 energy_partials_prev: torch.Tensor | None = None, energy_prev: torch.Tensor | None = None,
                 energy_sums_prev: int | None = None, energy_flush: bool = False, step_sync: torch.Tensor | None = None,
                 energy_sums16: int | None = None):
        """``energy_sums`` (optional): raw address (device, or pinned host memory) of cdiv(J, 256) doubles that receive the
        256-column chunk sums of the per-particle energies from the launch that finishes the step's energy by-product
        (Gaussian/identity fast paths; see pls_block_desc)."""
        assert block_cols > 0
        L.require_gpu_tensor(eta, "eta")
        assert eta.dim() == 1 and eta.is_contiguous()
        self.block_cols, self.eta, self.energy_sums = int(block_cols), eta, energy_sums
        #: optional: device int32 counters, one per 256 columns, zeroed once by the owner (pls_block_desc.energy_sync): the step
        #: launch then finishes the energies itself and leaves the counters zero
        if energy_sync is not None:
            assert energy_sync.device.type == "cuda" and energy_sync.dtype == torch.int32 and energy_sync.is_contiguous()
        self.energy_sync = energy_sync
        #: LAGGED energies (pls_block_desc.energy_partials ...): this launch leaves its partial rows in ``energy_partials``
        #: and finishes the previous launch's (``energy_partials_prev``) into ``energy_prev`` (J device doubles) and, optionally,
        #: ``energy_sums_prev`` (raw address of cdiv(J, 256) doubles, device or pinned host memory); ``energy_flush``: no
        #: step, only that finish (BasisWithFlush.flush_energies)
        for t in (energy_partials, energy_partials_prev, energy_prev):
            if t is not None:
                L.require_gpu_tensor(t, "energy buffer")
                assert t.is_contiguous()
        self.energy_partials, self.energy_partials_prev, self.energy_prev = energy_partials, energy_partials_prev, energy_prev
        self.energy_sums_prev, self.energy_flush = energy_sums_prev, bool(energy_flush)
        #: optional: device int32 counters (pls_step_sync_words(J) of them), zeroed once by the owner: the one-launch small-rank
        #: step (pls_block_desc.step_sync) meets through them and leaves them zero
        if step_sync is not None:
            assert step_sync.device.type == "cuda" and step_sync.dtype == torch.int32 and step_sync.is_contiguous()
        self.step_sync = step_sync
        #: optional: raw address (device, or pinned host memory) of cdiv(J, 16) doubles that receive the sums of the energies over
        #: each block of 16 columns (pls_block_desc.energy_sums16)
        self.energy_sums16 = energy_sums16

    def desc(self) -> L.BlockDesc:
        d = L.BlockDesc()
        d.block_cols, d.eta = self.block_cols, self.eta.data_ptr()
        d.energy_sums = self.energy_sums
        d.energy_sync = None if self.energy_sync is None else self.energy_sync.data_ptr()
        d.energy_partials = None if self.energy_partials is None else self.energy_partials.data_ptr()
        d.energy_partials_prev = None if self.energy_partials_prev is None else self.energy_partials_prev.data_ptr()
        d.energy_prev = None if self.energy_prev is None else self.energy_prev.data_ptr()
        d.energy_sums_prev = self.energy_sums_prev
        d.energy_flush = 1 if self.energy_flush else 0
        d.step_sync = None if self.step_sync is None else self.step_sync.data_ptr()
        d.energy_sums16 = self.energy_sums16
        return d


class PLSBasis(ABC):
    """Function-space basis: initialise particles, energy potential, particle update, predictive samples."""

    def __init__(self, additional_predictive_noise_distribution: Optional[torch.distributions.Distribution] = None):
        self.additional_predictive_noise_distribution = additional_predictive_noise_distribution
        #: global column index of this rank's first particle (J-sharding, see distributed.py)
        self.j_offset = 0
        self._ws: dict = {}
        self.workspace_bytes = DEFAULT_WORKSPACE_BYTES

    @property
    def approximation_dimension(self) -> int:
        raise NotImplementedError

    # ---- noise -------------------------------------------------------------------------------------------------
    def _draw_noise_spec(self, noise: torch.Tensor | None) -> NoiseSpec:
        """The reference draws the step noise from torch's GLOBAL CPU generator (samplers.py:30-35 with
        generator=None), so callers make runs reproducible with set_seed() before the loop (runners.py:364).
        Same contract here: one 63-bit draw from that generator keys the on-device Philox stream of this step."""
        if noise is not None:
            return NoiseSpec(injected=noise)
        seed = int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())
        return NoiseSpec(seed=seed, step=0, j_offset=self.j_offset)

    def _workspace(self, nbytes: int, device) -> torch.Tensor:
        """The basis' own scratch buffer for EAGER calls: grown on demand, so its address may change between calls.
        Anything that freezes a pointer (a captured hipGraph) must own its buffer and pass it as ``workspace=``."""
        key = str(device)
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)
            self._ws[key] = ws
        return ws

    def _pick_workspace(self, workspace: torch.Tensor | None, nbytes: int, device) -> torch.Tensor:
        if workspace is None:
            return self._workspace(nbytes, device)
        L.require_gpu_tensor(workspace, "workspace")
        if workspace.numel() * 8 < nbytes:
            raise L.PlsHipError(f"workspace of {workspace.numel() * 8} bytes handed in, {nbytes} needed")
        return workspace

    # ---- the one-launch small-rank step (csrc/small_rank_step.h): what both bases keep for it ----------------------
    def _step_sync(self, j: int, device) -> torch.Tensor:
        """The zeroed arrival counters of the one-launch small-rank step (pls_block_desc.step_sync) for eager calls: one set per
        stream (launches on one stream are ordered; two streams must not share counters), grown on demand.  A launch leaves
        them zero; after a FAILED launch they are dropped (zero_step_sync) -- stale counts would make every later step wrong."""
        words = int(L.load().pls_step_sync_words(j))
        key = (str(device), L.stream_ptr())
        pool = self.__dict__.setdefault("_sync_pool", {})
        t = pool.get(key)
        if t is None or t.numel() < words:
            t = torch.zeros(max(words, 64), dtype=torch.int32, device=device)
            pool[key] = t
        return t

    def zero_step_sync(self) -> None:
        """Forget every counter set (after a failed or aborted launch): the next step allocates zeroed ones."""
        self.__dict__.pop("_sync_pool", None)
        self.__dict__.pop("_eager", None)  # (the eager step binds a counter set)

    def _eta_word(self, step_size: float, device) -> torch.Tensor:
        """``step_size`` as a device word (pls_block_desc.eta), remembered per value: an eager caller steps with the same size
        thousands of times, and a host -> device copy per call would cost more than the step."""
        words = self.__dict__.setdefault("_eta_words", {})
        key = (float(step_size), str(device))
        t = words.get(key)
        if t is None:
            if len(words) >= 64:
                words.clear()
            t = torch.full((1,), float(step_size), dtype=torch.float64, device=device)
            words[key] = t
        return t

    # ---- particles ---------------------------------------------------------------------------------------------
    def _initialise_particles_noise(self, number_of_particles: int, seed: int | None = None, mean: float = 0.0,
                                    stdev: float = 1.0) -> torch.Tensor:
        """basis/base.py:39-63: torch.normal on a CPU generator, size (M, J)."""
        generator = None
        if seed is not None:
            generator = torch.Generator().manual_seed(seed)
        return torch.normal(
            mean=mean, std=stdev, size=(self.approximation_dimension, number_of_particles), generator=generator
        )

    @abstractmethod
    def _initialise_particles(self, number_of_particles: int, noise_only: bool = True, seed: int | None = None) -> torch.Tensor:
        raise NotImplementedError

    def initialise_particles(self, number_of_particles: int, noise_only: bool = True, seed: int | None = None) -> torch.Tensor:
        """basis/base.py:81-102; the particles always live on the MI355X as float64."""
        return _dev(self._initialise_particles(number_of_particles=number_of_particles, noise_only=noise_only, seed=seed))

    @abstractmethod
    def calculate_untransformed_train_prediction_samples(self, particles: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def calculate_energy_potential(self, particles: torch.Tensor, cost: torch.Tensor) -> float:
        raise NotImplementedError

    @abstractmethod
    def _calculate_particle_update(self, particles: torch.Tensor, cost_derivative: torch.Tensor, step_size: float,
                                   noise: torch.Tensor | None = None) -> torch.Tensor:
        raise NotImplementedError

    def calculate_particle_update(self, particles: torch.Tensor, cost_derivative: torch.Tensor, step_size: float,
                                  noise: torch.Tensor | None = None) -> torch.Tensor:
        """basis/base.py:143-163 (same assertion and message)."""
        assert (
            particles.shape[0] == self.approximation_dimension
        ), f"Particles have shape {particles.shape} but requires ({self.approximation_dimension}, J) dimension."
        extra = {} if noise is None else {"noise": noise}  # (subclasses with the reference's 3-argument signature keep working)
        return self._calculate_particle_update(
            particles=particles, cost_derivative=cost_derivative, step_size=step_size, **extra
        )

    @abstractmethod
    def sample_predictive_noise(self, particles: torch.Tensor, x: torch.Tensor):
        raise NotImplementedError

    @abstractmethod
    def predict_untransformed_samples(self, particles: torch.Tensor, x: torch.Tensor,
                                      noise: torch.Tensor | None = None) -> torch.Tensor:
        raise NotImplementedError

    # ---- fused native path (used by PLS when the cost is native) -------------------------------------------------
    def supports_fused_step(self) -> bool:
        return False


def padded_ld(cols: int) -> int:
    """Leading dimension rounded up to 16 doubles (128 B): every row of a library-owned matrix starts on a full line."""
    return (cols + 15) // 16 * 16


def alloc_matrix(rows: int, cols: int, device) -> torch.Tensor:
    """(rows, cols) float64 view with a padded leading dimension."""
    ldm = padded_ld(cols)
    return torch.empty((rows, ldm), dtype=torch.float64, device=device)[:, :cols]
