"""Conformal prediction intervals (drop-in for src/conformalise/base.py:9-160 and src/conformalise/pls.py:8-62).

The per-test-point quantiles over the J particles are the J-reduction of this wrapper; they run as one LDS sort per
test point (pls_row_quantiles).  On a J-sharded run every rank predicts its own particles' samples, the rows (test points)
are dealt out over the ranks by an all-to-all, each rank sorts its N*/G rows of all J samples, and the quantiles are
all-gathered (distributed.sharded_row_quantiles: collective C3 of SURVEY.md 8e)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from . import _ops
from .kernel import _dev
from .projected_langevin_sampling import PLS


@dataclass
class ConformalPrediction:
    coverage: float
    mean: torch.Tensor
    lower: torch.Tensor
    upper: torch.Tensor


class ConformalisePLS:
    """conformalise/pls.py:8-62 on top of conformalise/base.py:19-160 (https://arxiv.org/abs/2107.07511)."""

    def __init__(self, x_calibration: torch.Tensor, y_calibration: torch.Tensor, pls: PLS, particles: torch.Tensor, group=None):
        self.pls = pls
        self.particles = particles
        self.group = group
        self.x_calibration = x_calibration
        self.y_calibration = y_calibration
        self.number_of_calibration_points = x_calibration.shape[0]

    def _quantiles(self, x: torch.Tensor, q) -> torch.Tensor:
        """(N*, len(q)) quantiles over all J particles: this rank's samples, then the sharded reduction"""
        from .distributed import sharded_row_quantiles

        samples = self.pls.predict_samples(x=x, particles=self.particles, predictive_noise=None, observation_noise=None)
        return sharded_row_quantiles(samples, q, self.group)

    def _predict_uncalibrated_coverage(self, coverage: float, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Quantiles 0.5 -/+ coverage/2 of the particle predictions (conformalise/pls.py:24-45)."""
        q = self._quantiles(x, [0.5 - coverage / 2, 0.5 + coverage / 2])
        return q[:, 0].contiguous(), q[:, 1].contiguous()

    def predict_median(self, x: torch.Tensor) -> torch.Tensor:
        return self._quantiles(x, [0.5])[:, 0].contiguous()  # conformalise/pls.py:47-62

    def _calculate_calibration(self, coverage: float) -> float:
        """conformalise/base.py:58-90: the (n+1) c / n quantile of the conformity scores."""
        lower, upper = self._predict_uncalibrated_coverage(x=self.x_calibration, coverage=coverage)
        y = _dev(self.y_calibration.reshape(-1))
        scores = torch.max(torch.stack([lower - y, y - upper], dim=1), dim=1).values
        level = float(np.clip((self.number_of_calibration_points + 1) * coverage / self.number_of_calibration_points, 0.0, 1.0))
        return _ops.row_quantiles(scores.reshape(1, -1), [level])[0, 0].item()

    def predict_coverage(self, x: torch.Tensor, coverage: float) -> Tuple[torch.Tensor, torch.Tensor]:
        """conformalise/base.py:92-114: calibrated bounds, clamped so that nothing crosses the median."""
        calibration = self._calculate_calibration(coverage)
        lower, upper = self._predict_uncalibrated_coverage(x=x, coverage=coverage)
        lower, upper = lower - calibration, upper + calibration
        median = self.predict_median(x)
        return torch.minimum(lower, median), torch.maximum(upper, median)

    def calculate_average_interval_width(self, x: torch.Tensor, coverage: float) -> float:
        lower, upper = self.predict_coverage(x=x, coverage=coverage)
        return torch.mean(upper - lower).item()  # base.py:116-128

    def predict_variance(self, x: torch.Tensor) -> torch.Tensor:
        lower, upper = self.predict_coverage(x=x, coverage=2 / 3)
        return (upper - lower) / 2  # base.py:130-141

    def predict(self, x: torch.Tensor, coverage: float) -> ConformalPrediction:
        lower, upper = self.predict_coverage(x=x, coverage=coverage)
        return ConformalPrediction(coverage=coverage, mean=self.predict_median(x=x), lower=lower, upper=upper)

    def __call__(self, x: torch.Tensor, coverage: float) -> ConformalPrediction:
        return self.predict(x=x, coverage=coverage)
