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
    per-particle energy vector to the global mean (distributed.mean_over_particles for J-sharded runs).

    When the step kernel can emit the energy of its input particles as a by-product (Gaussian/identity on the
    orthonormal basis), the loop is software-pipelined: the launch of step t+1 also produces the energy the reference
    evaluates after step t, so every iteration is ONE kernel.  Results (particles, energy list, stop index, torch RNG
    state) are those of the plain loop: a step launched speculatively past the stop is discarded."""
    reduce = energy_reduce if energy_reduce is not None else (lambda e: e.mean().item())
    early_stopper = EarlyStopper(patience=early_stopper_patience)
    energy_potentials: List[float] = []
    pipelined = (
        number_of_epochs > 0 and pls._fused() and getattr(pls.basis, "supports_input_energy", lambda c: False)(pls.cost)
    )
    if not pipelined:
        for t in range(number_of_epochs):
            pls.step_(particles, step_size, noise=None if noises is None else noises[t])
            energy_potential = reduce(pls.particle_energy_potential(particles))
            if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
                break
            energy_potentials.append(energy_potential)
        return particles, energy_potentials

    from .basis.base import NoiseSpec

    cur = particles
    nxt = torch.empty_like(particles, memory_format=torch.contiguous_format)
    e_in = torch.empty(particles.shape[1], dtype=torch.float64, device=particles.device)
    stopped = False
    for t in range(number_of_epochs):
        rng_state = torch.get_rng_state()  # the speculative launch below may have to be un-drawn
        spec = NoiseSpec(injected=noises[t]) if noises is not None else None
        pls.basis.fused_step(pls.cost, cur, float(step_size), out=nxt, new_state=True, noise=spec, input_energy=e_in)
        if t >= 1:  # e_in = energy of `cur`, i.e. of the particles after update t-1 (trainers.py:158)
            energy_potential = reduce(e_in)
            if early_stopper.should_stop(loss=energy_potential, step_size=step_size):
                torch.set_rng_state(rng_state)
                stopped = True
                break
            energy_potentials.append(energy_potential)
        cur, nxt = nxt, cur
    if not stopped:  # energy after the last update
        energy_potential = reduce(pls.particle_energy_potential(cur))
        if not early_stopper.should_stop(loss=energy_potential, step_size=step_size):
            energy_potentials.append(energy_potential)
    if cur.data_ptr() != particles.data_ptr():
        particles.copy_(cur)
    return particles, energy_potentials
