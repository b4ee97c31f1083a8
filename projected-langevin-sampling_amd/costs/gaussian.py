"""Gaussian cost (drop-in for costs/gaussian.py:9-110)."""
import torch

from .. import _lib as L
from ..link_functions import PLSLinkFunction
from .base import PLSCost


class GaussianCost(PLSCost):
    """c_j = (1 / (2 * observation_noise)) * ||link(f_j) - y||^2  -- observation_noise acts as a VARIANCE
    (gaussian.py:71, :86) while sample_observation_noise uses it as a std (costs/base.py:106-111): kept."""

    cost_kind = L.COST_GAUSSIAN

    def __init__(self, observation_noise: float, y_train: torch.Tensor, link_function: PLSLinkFunction):
        super().__init__(link_function=link_function, observation_noise=observation_noise)
        self.y_train = y_train

    def _params(self):
        return (float(self.observation_noise), 0.0, 0.0, 0.0)

    def predict(self, prediction_samples: torch.Tensor, number_of_particles: int | None = None,
                group=None) -> torch.distributions.MultivariateNormal:
        """gaussian.py:40-52: mean and unbiased variance over the particle axis.  On a J-sharded run pass the global
        particle count: the two passes are then all-reduced over the ranks (distributed.predictive_moments)."""
        from ..distributed import predictive_moments

        mean, var = predictive_moments(prediction_samples, number_of_particles or prediction_samples.shape[1], group=group)
        return torch.distributions.MultivariateNormal(loc=mean, covariance_matrix=torch.diag(var))
