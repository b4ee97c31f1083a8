"""Poisson cost (drop-in for costs/poisson.py:10-104)."""
import torch

from .. import _lib as L
from ..link_functions import PLSLinkFunction
from .base import PLSCost


class PoissonCost(PLSCost):
    """c_j = sum_n -2 y_n log|f_nj| + link(f_nj)  (poisson.py:59-66)."""

    cost_kind = L.COST_POISSON

    def __init__(self, y_train: torch.Tensor, link_function: PLSLinkFunction):
        super().__init__(link_function=link_function, observation_noise=None)
        self.y_train = y_train

    def predict(self, prediction_samples: torch.Tensor) -> torch.distributions.Poisson:
        return torch.distributions.Poisson(rate=prediction_samples.mean(dim=1))  # poisson.py:43-45
