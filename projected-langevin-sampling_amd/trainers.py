"""Caller of the hot path (drop-in for experiments/trainers.py:139-162 and experiments/early_stopper.py:4-24)."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from .projected_langevin_sampling import PLS


class EarlyStopper:
    """Stops when the simulated time without an improving loss reaches ``patience`` (early_stopper.py:4-24)."""

    def __init__(self, patience: float = 1e-4):
        self.patience = patience
        self.simulation_time = 0
        self.min_loss = float("inf")

    def should_stop(self, loss: float, step_size: float) -> bool:
        if not np.isfinite(loss):
            return True
        elif loss >= self.min_loss:
            self.simulation_time += step_size
            return self.simulation_time >= self.patience
        else:
            self.min_loss = loss
            self.simulation_time = 0
            return False


def train_pls(
    pls: PLS,
    particles: torch.Tensor,
    number_of_epochs: int,
    step_size: float,
    early_stopper_patience: float,
    tqdm_desc: str | None = None,
    noises: List[torch.Tensor] | None = None,
    energy_reduce=None,
) -> Tuple[torch.Tensor, List[float]]:
    """trainers.py:139-162: update, in-place add, energy, early stop.

    ``noises`` (extension) injects the noise of each step; ``energy_reduce`` (extension) maps the local
    per-particle energy vector to the global mean (distributed.mean_over_particles for J-sharded runs)."""
    energy_potentials: List[float] = []
    early_stopper = EarlyStopper(patience=early_stopper_patience)
    for t in range(number_of_epochs):
        pls.step_(particles, step_size, noise=None if noises is None else noises[t])
        if energy_reduce is None:
            energy_potential = pls.calculate_energy_potential(particles=particles)
        else:
            energy_potential = energy_reduce(pls.particle_energy_potential(particles))
        if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
            break
        energy_potentials.append(energy_potential)
    return particles, energy_potentials
