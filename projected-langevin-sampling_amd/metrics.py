"""The metrics the step-size search optimises (drop-in for experiments/metrics.py:23-125, the branches that apply to
PLS predictions).  Inputs are per-test-point vectors: host-side arithmetic on N* numbers."""
from __future__ import annotations

import math

import torch

from .conformalise import ConformalPrediction


def _point(prediction) -> torch.Tensor:
    if isinstance(prediction, torch.distributions.MultivariateNormal):
        return prediction.mean
    if isinstance(prediction, torch.distributions.Bernoulli):
        return prediction.probs
    if isinstance(prediction, torch.distributions.Poisson):
        return prediction.rate
    if isinstance(prediction, torch.distributions.StudentT):
        return prediction.loc
    if isinstance(prediction, ConformalPrediction):
        return prediction.mean
    raise ValueError(f"Prediction type {type(prediction)} not supported")


def calculate_mae(prediction, y: torch.Tensor) -> float:
    p = _point(prediction)
    return p.sub(y.to(p)).abs().mean().item()  # metrics.py:23-45


def calculate_mse(prediction, y: torch.Tensor) -> float:
    p = _point(prediction)
    return p.sub(y.to(p)).pow(2).mean().item()  # metrics.py:48-70


def calculate_nll(prediction, y: torch.Tensor) -> float:
    """metrics.py:73-125."""
    if isinstance(prediction, torch.distributions.MultivariateNormal):
        # gpytorch.metrics.mean_standardized_log_loss without train_y: mean 0.5 (log(2 pi s^2) + (y - m)^2 / s^2)
        m, v = prediction.mean, torch.diagonal(prediction.covariance_matrix)
        yy = y.to(m)
        return (0.5 * (torch.log(2 * math.pi * v) + torch.square(yy - m) / v)).mean().item()
    if isinstance(prediction, torch.distributions.Bernoulli):
        return torch.nn.functional.binary_cross_entropy(prediction.probs, y.to(prediction.probs), reduction="mean").item()
    if isinstance(prediction, torch.distributions.Poisson):
        return torch.nn.functional.poisson_nll_loss(prediction.rate, y.to(prediction.rate), reduction="mean").item()
    if isinstance(prediction, torch.distributions.StudentT):
        return prediction.log_prob(y.to(prediction.loc)).mean().item()  # metrics.py:98-99 (sign as in the reference)
    raise ValueError(f"Prediction type {type(prediction)} not supported")
