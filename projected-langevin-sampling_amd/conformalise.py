"""Conformal prediction intervals (drop-in for src/conformalise/base.py:9-160 and src/conformalise/pls.py:8-62).

The per-test-point quantiles over the J particles are the J-reduction of this wrapper; they run as one LDS sort per
test point (pls_row_quantiles).  On a J-sharded run the prediction samples of all ranks are all-gathered first
(collective C3 of SURVEY.md 2.2), because an order statistic needs every sample of its row."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _ops
from .kernel import _dev
from .projected_langevin_sampling import PLS


@dataclass
class ConformalPrediction:
    coverage: float
    mean: torch.Tensor
    lower: torch.Tensor
    upper: torch.Tensor


def _gather_columns(local: torch.Tensor, group=None) -> torch.Tensor:
    """(N*, J_local) per rank -> (N*, J) on every rank; shards may differ by one column."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    widths = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(widths, torch.tensor([local.shape[1]], dtype=torch.int64, device=local.device), group=group)
    widths = [int(w.item()) for w in widths]
    wmax = max(widths)
    padded = torch.zeros((wmax, local.shape[0]), dtype=local.dtype, device=local.device)
    padded[: local.shape[1]] = local.T
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:w].T for p, w in zip(parts, widths)], dim=1).contiguous()


class ConformalisePLS:
    """conformalise/pls.py:8-62 on top of conformalise/base.py:19-160 (https://arxiv.org/abs/2107.07511)."""

    def __init__(self, x_calibration: torch.Tensor, y_calibration: torch.Tensor, pls: PLS, particles: torch.Tensor, group=None):
        self.pls = pls
        self.particles = particles
        self.group = group
        self.x_calibration = x_calibration
        self.y_calibration = y_calibration
        self.number_of_calibration_points = x_calibration.shape[0]

    def _samples(self, x: torch.Tensor) -> torch.Tensor:
        samples = self.pls.predict_samples(x=x, particles=self.particles, predictive_noise=None, observation_noise=None)
        return _gather_columns(samples, self.group)

    def _predict_uncalibrated_coverage(self, coverage: float, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Quantiles 0.5 -/+ coverage/2 of the particle predictions (conformalise/pls.py:24-45)."""
        q = _ops.row_quantiles(self._samples(x), [0.5 - coverage / 2, 0.5 + coverage / 2])
        return q[:, 0].contiguous(), q[:, 1].contiguous()

    def predict_median(self, x: torch.Tensor) -> torch.Tensor:
        return _ops.row_quantiles(self._samples(x), [0.5])[:, 0].contiguous()  # conformalise/pls.py:47-62

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
