"""Student-t cost (drop-in for costs/student_t.py:11-110)."""
import torch

from .. import _lib as L
from ..link_functions import PLSLinkFunction
from .base import PLSCost


class StudentTCost(PLSCost):
    """c_j = 0.5 (nu + 1) sum_n log(1 + (link(f_nj) - y_n)^2 / (nu s^2))  (student_t.py:57-72)."""

    cost_kind = L.COST_STUDENT_T

    def __init__(self, degrees_of_freedom: float, y_train: torch.Tensor, link_function: PLSLinkFunction, scale: float = 1.0):
        super().__init__(link_function=link_function, observation_noise=None)
        self.y_train = y_train
        self.degrees_of_freedom = degrees_of_freedom
        self.scale = scale

    def _params(self):
        return (float(self.degrees_of_freedom), float(self.scale), 0.0, 0.0)

    def predict(self, prediction_samples: torch.Tensor) -> torch.distributions.StudentT:
        """student_t.py:40-53 returns the reference's StudentTMarginals container (src/distributions.py,
        outside the hot path); the same marginals as a torch StudentT."""
        loc = self.link_function(prediction_samples).mean(dim=1)
        return torch.distributions.StudentT(
            df=torch.as_tensor(float(self.degrees_of_freedom), device=loc.device),
            loc=loc,
            scale=self.scale * torch.ones_like(loc),
        )
