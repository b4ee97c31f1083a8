"""Two-mode Gaussian mixture cost (drop-in for costs/multimodal.py:7-91)."""
import torch

from .. import _lib as L
from ..link_functions import PLSLinkFunction
from .base import PLSCost


class MultiModalCost(PLSCost):
    """c_j = -sum_n logsumexp(log pi + N(y_n + shift | p, s), log(1 - pi) + N(y_n | p, s)); observation_noise is a
    STD here (multimodal.py:56).  The reference always differentiates it with autograd (:79-91); libplship
    evaluates the same derivative in closed form."""

    cost_kind = L.COST_MULTIMODAL

    def __init__(self, observation_noise: float, shift: float, bernoulli_noise: float, y_train: torch.Tensor,
                 link_function: PLSLinkFunction):
        super().__init__(link_function=link_function, observation_noise=observation_noise)
        self.shift = shift
        self.bernoulli_noise = bernoulli_noise
        self.y_train = y_train

    def _params(self):
        return (float(self.observation_noise), float(self.shift), float(self.bernoulli_noise), 0.0)

    def predict(self, prediction_samples: torch.Tensor) -> None:
        pass  # multimodal.py:29-35

    def calculate_cost_derivative(self, untransformed_train_prediction_samples: torch.Tensor, force_autograd: bool = True):
        return super().calculate_cost_derivative(untransformed_train_prediction_samples, force_autograd=True)
