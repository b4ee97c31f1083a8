"""PLS facade (drop-in for src/projected_langevin_sampling/projected_langevin_sampling.py:7-204)."""
from __future__ import annotations

import torch

from . import _lib as L
from . import _ops
from .basis.base import NoiseSpec, PLSBasis
from .costs.base import PLSCost


class PLS:
    """Composes a basis and a cost (projected_langevin_sampling.py:19-27).

    When both are libplship-native, calculate_particle_update runs the whole step
    (forward projection -> cost derivative -> back-projection -> prior drift -> noise) inside libplship
    (pls_onb_step / pls_ipb_step); otherwise it composes the un-fused entry points around the user's
    Python cost or basis, exactly like the reference composes its methods."""

    def __init__(self, basis: PLSBasis, cost: PLSCost, name: str | None = None):
        self.basis = basis
        self.cost = cost
        self.name: str = name if name is not None else "pls"
        self._pong: torch.Tensor | None = None

    @property
    def observation_noise(self) -> None | float:
        return self.cost.observation_noise

    @observation_noise.setter
    def observation_noise(self, value: float):
        self.cost.observation_noise = value

    def _fused(self) -> bool:
        native_cost = getattr(self.cost, "is_native", lambda: False)()
        return native_cost and getattr(self.basis, "supports_fused_step", lambda: False)()

    def initialise_particles(self, number_of_particles: int, noise_only: bool = True, seed: int | None = None) -> torch.Tensor:
        return self.basis.initialise_particles(number_of_particles=number_of_particles, noise_only=noise_only, seed=seed)

    def sample_observation_noise(self, number_of_particles: int, seed: int | None = None) -> torch.Tensor:
        j_offset = getattr(self.basis, "j_offset", 0)
        if j_offset and getattr(self.cost, "is_native", lambda: False)():
            # (a J-sharded run: the draw of a particle is keyed by its GLOBAL column, so every shard holds its own draws)
            return self.cost.sample_observation_noise(number_of_particles=number_of_particles, seed=seed, j_offset=j_offset)
        return self.cost.sample_observation_noise(number_of_particles=number_of_particles, seed=seed)

    def sample_predictive_noise(self, particles: torch.Tensor, x: torch.Tensor):
        return self.basis.sample_predictive_noise(particles=particles, x=x)

    def calculate_cost(self, particles: torch.Tensor) -> torch.Tensor:
        """(M, J) -> (J,)  (:75-88)."""
        f = self.basis.calculate_untransformed_train_prediction_samples(particles=particles)
        return self.cost.calculate_cost(untransformed_train_prediction_samples=f)

    def calculate_cost_derivative(self, particles: torch.Tensor) -> torch.Tensor:
        """(M, J) -> (N, J)  (:90-105)."""
        f = self.basis.calculate_untransformed_train_prediction_samples(particles=particles)
        return self.cost.calculate_cost_derivative(untransformed_train_prediction_samples=f)

    def calculate_particle_update(self, particles: torch.Tensor, step_size: float,
                                  noise: torch.Tensor | None = None) -> torch.Tensor:
        """dU for one Langevin step (:107-123).  ``noise`` (extension) injects the step's noise matrix."""
        step_size = float(step_size)
        if self._fused():
            assert (
                particles.shape[0] == self.basis.approximation_dimension
            ), f"Particles have shape {particles.shape} but requires ({self.basis.approximation_dimension}, J) dimension."
            if noise is None:  # the drop-in loop's call: everything but the addresses bound once (basis.eager_step)
                eager = getattr(self.basis, "eager_step", None)
                update = eager(self.cost, particles, step_size) if eager is not None else None
                if update is not None:
                    return update
            spec = NoiseSpec(injected=noise) if noise is not None else None
            return self.basis.fused_step(self.cost, particles, step_size, noise=spec)
        cost_derivative = self.calculate_cost_derivative(particles=particles)
        # `noise` is this library's extension: a user-defined basis written against the reference's abstract signature
        # (particles, cost_derivative, step_size) never sees the keyword unless the caller injects noise
        extra = {} if noise is None else {"noise": noise}
        return self.basis.calculate_particle_update(
            particles=particles, cost_derivative=cost_derivative, step_size=step_size, **extra
        )

    def step_(self, particles: torch.Tensor, step_size: float, noise: torch.Tensor | None = None) -> torch.Tensor:
        """particles += calculate_particle_update(particles, step_size) in one fused launch sequence
        (the loop body of experiments/trainers.py:153-157).  Mutates and returns ``particles``."""
        if not self._fused():
            particles += self.calculate_particle_update(particles, step_size, noise=noise)
            return particles
        if particles.is_cuda and particles.dtype in L.PROMOTED_DTYPES:
            # float32 particles (the reference's bases compute in the caller's dtype, basis/base.py:52-63): the step runs in
            # float64 and the caller's tensor receives the rounded state, like `particles += update` would
            state = self.step_(particles.double(), step_size, noise=noise)
            particles.copy_(state)
            return particles
        if self._pong is None or self._pong.shape != particles.shape or self._pong.device != particles.device:
            self._pong = torch.empty_like(particles, memory_format=torch.contiguous_format)
        spec = NoiseSpec(injected=noise) if noise is not None else None
        self.basis.fused_step(self.cost, particles, float(step_size), out=self._pong, new_state=True, noise=spec)
        particles.copy_(self._pong)
        return particles

    def particle_energy_potential(self, particles: torch.Tensor) -> torch.Tensor:
        """Per-particle energies (J,) on the device: what the reference averages at orthonormal.py:126."""
        assert (
            particles.shape[0] == self.basis.approximation_dimension
        ), f"Particles have shape {particles.shape} but requires ({self.basis.approximation_dimension}, J) dimension."
        if self._fused():
            return self.basis.fused_particle_energy(self.cost, particles)
        cost = self.calculate_cost(particles=particles)
        return self.basis.particle_energy_potential(particles, cost)

    def calculate_energy_potential(self, particles: torch.Tensor) -> float:
        """Average energy potential (:125-138); returns a Python float (device sync, like the reference)."""
        if hasattr(self.basis, "particle_energy_potential"):
            e = self.particle_energy_potential(particles)
            return _ops.block_means(e).item() if e.is_cuda else e.mean().item()
        # a user-defined basis that only implements the reference's abstract interface
        assert (
            particles.shape[0] == self.basis.approximation_dimension
        ), f"Particles have shape {particles.shape} but requires ({self.basis.approximation_dimension}, J) dimension."
        cost = self.calculate_cost(particles=particles)
        return self.basis.calculate_energy_potential(particles=particles, cost=cost)

    def predict_samples(self, particles: torch.Tensor, x: torch.Tensor, predictive_noise: torch.Tensor | None = None,
                        observation_noise: torch.Tensor | None = None) -> torch.Tensor:
        untransformed_samples = self.predict_untransformed_samples(particles=particles, x=x, noise=predictive_noise)
        j_offset = getattr(self.basis, "j_offset", 0)
        if observation_noise is None and j_offset and getattr(self.cost, "is_native", lambda: False)():
            # (a J-sharded run: the library's costs key their per-particle draw by the GLOBAL particle column)
            return self.cost.predict_samples(untransformed_samples=untransformed_samples, observation_noise=None,
                                             j_offset=j_offset)
        return self.cost.predict_samples(untransformed_samples=untransformed_samples, observation_noise=observation_noise)

    def predict_untransformed_samples(self, particles: torch.Tensor, x: torch.Tensor,
                                      noise: torch.Tensor | None = None) -> torch.Tensor:
        return self.basis.predict_untransformed_samples(particles=particles, x=x, noise=noise)

    def predict(self, x: torch.Tensor, particles: torch.Tensor, predictive_noise: torch.Tensor | None = None,
                observation_noise: torch.Tensor | None = None) -> torch.distributions.Distribution:
        prediction_samples = self.predict_samples(
            particles=particles, x=x, predictive_noise=predictive_noise, observation_noise=observation_noise
        )
        return self.cost.predict(prediction_samples=prediction_samples)

    def __call__(self, x: torch.Tensor, particles: torch.Tensor, predictive_noise: torch.Tensor | None = None,
                 observation_noise: torch.Tensor | None = None) -> torch.distributions.Distribution:
        return self.predict(x=x, particles=particles, predictive_noise=predictive_noise, observation_noise=observation_noise)
