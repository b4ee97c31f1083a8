"""Bernoulli cost (drop-in for costs/bernoulli.py:10-99)."""
import torch

from .. import _lib as L
from ..link_functions import PLSLinkFunction
from .base import PLSCost


class BernoulliCost(PLSCost):
    """c_j = -sum_n y_n log p_nj + (1 - y_n) log(1 - p_nj), p = link(f)  (bernoulli.py:57-62)."""

    cost_kind = L.COST_BERNOULLI

    def __init__(self, y_train: torch.Tensor, link_function: PLSLinkFunction):
        super().__init__(link_function=link_function, observation_noise=None)
        self.y_train = y_train.type(torch.double)  # bernoulli.py:32

    def predict(self, prediction_samples: torch.Tensor) -> torch.distributions.Bernoulli:
        return torch.distributions.Bernoulli(probs=prediction_samples.mean(dim=1))  # bernoulli.py:43-46
