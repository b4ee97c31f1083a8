"""Variance tempering (drop-in for src/temper/base.py:8-59 and src/temper/pls.py:8-52)."""
from __future__ import annotations

import torch

from .costs import GaussianCost
from .kernel import _dev
from .projected_langevin_sampling import PLS


class TemperPLS:
    """scale = (2/N) sum_i (y_i - m(x_i))^2 / sigma_i^2 on a calibration set (temper/base.py:30-46); predictions keep
    their mean and get covariance * scale (:54-59).  ``number_of_particles`` / ``group`` make the J-reduction global on
    a sharded run (every rank then holds the same scale)."""

    def __init__(self, x_calibration: torch.Tensor, y_calibration: torch.Tensor, pls: PLS, particles: torch.Tensor,
                 debug: bool = False, number_of_particles: int | None = None, group=None):
        self.debug = debug
        if not self.debug:
            assert isinstance(pls.cost, GaussianCost)  # temper/pls.py:24-25
        self.pls = pls
        self.particles = particles
        self.number_of_particles = number_of_particles
        self.group = group
        self.scale = self._calculate_scale(x_calibration=x_calibration, y_calibration=y_calibration)

    def _untempered_predict(self, x: torch.Tensor) -> torch.distributions.MultivariateNormal:
        samples = self.pls.predict_samples(particles=self.particles, x=x)
        return self.pls.cost.predict(samples, number_of_particles=self.number_of_particles, group=self.group)

    def _calculate_scale(self, x_calibration: torch.Tensor, y_calibration: torch.Tensor) -> float:
        prediction = self._untempered_predict(x=x_calibration)
        y = _dev(y_calibration.reshape(-1))
        return 2 * torch.mean(torch.div(torch.square(y - prediction.mean), torch.diag(prediction.covariance_matrix))).item()

    def predict(self, x: torch.Tensor) -> torch.distributions.MultivariateNormal:
        prediction = self._untempered_predict(x=x)
        return torch.distributions.MultivariateNormal(
            loc=prediction.mean, covariance_matrix=prediction.covariance_matrix * self.scale
        )

    def __call__(self, x: torch.Tensor) -> torch.distributions.MultivariateNormal:
        return self.predict(x=x)
