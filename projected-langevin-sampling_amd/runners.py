"""Step-size search around the training loop (drop-in for experiments/runners.py:331-446).

Log-spaced step sizes from ``step_size_upper`` down to ``simulation_duration / maximum_number_of_steps``; each candidate
re-runs train_pls from the same initial particles after set_seed(seed); a run counts if it produced energies and finite
particles; the best run by ``metric_to_optimise`` is kept; the search ends once two consecutive accepted runs' final
energies agree to ``minimum_change_in_energy_potential`` (relative).  Plotting hooks of the reference are out of scope."""
from __future__ import annotations

from copy import deepcopy
from typing import Callable, Dict, List, Tuple

import numpy as np
import torch

from .metrics import calculate_mae, calculate_mse, calculate_nll
from .projected_langevin_sampling import PLS
from .trainers import train_pls
from .utils import set_seed


def train_pls_runner(
    pls: PLS,
    particle_name: str,
    x_train: torch.Tensor,
    y_train: torch.Tensor,
    simulation_duration: float,
    maximum_number_of_steps: int,
    early_stopper_patience: float,
    number_of_step_searches: int,
    step_size_upper: float,
    minimum_change_in_energy_potential: float,
    seed: int,
    particles: torch.Tensor,
    metric_to_optimise: str = "nll",
    train_fn: Callable = train_pls,
) -> Tuple[torch.Tensor, float, int]:
    """Returns (best particles, best step size, number of energies of the best run), like runners.py:446.
    (The reference takes an ExperimentData; only its train.x / train.y are used, :374-391.)"""
    if metric_to_optimise in ["nll", "mse", "mae", "loss"]:
        best_metric_value = float("inf")  # runners.py:347-348
    elif metric_to_optimise in ["acc", "auc", "f1"]:
        best_metric_value = 0
    else:
        raise NotImplementedError(f"Unknown metric to optimise {metric_to_optimise}.")
    best_lr = None
    energy_potentials_history: Dict[float, List[float]] = {}
    step_sizes = np.logspace(
        np.log10(step_size_upper), np.log10(simulation_duration / maximum_number_of_steps), number_of_step_searches
    )  # :356-360
    particles_out = particles.detach().clone()
    for i, step_size in enumerate(step_sizes):
        number_of_epochs = int(simulation_duration / step_size)  # :363
        set_seed(seed)  # :364
        particles_i, energy_potentials = train_fn(
            pls=pls,
            particles=particles.detach().clone(),
            number_of_epochs=number_of_epochs,
            step_size=step_size,
            early_stopper_patience=early_stopper_patience,
        )
        if energy_potentials and torch.isfinite(particles_i).all():  # :373
            energy_potentials_history[step_size] = energy_potentials
            if metric_to_optimise == "loss":
                metric_value = energy_potentials[-1]  # :407-408
            else:
                prediction = pls.predict(x=x_train, particles=particles_i)  # :375-378
                if metric_to_optimise == "nll":
                    metric_value = calculate_nll(prediction=prediction, y=y_train)
                elif metric_to_optimise == "mse":
                    metric_value = calculate_mse(prediction=prediction, y=y_train)
                elif metric_to_optimise == "mae":
                    metric_value = calculate_mae(prediction=prediction, y=y_train)
                else:  # acc / auc / f1 (:392-406) through sklearn, as in the reference
                    import sklearn.metrics

                    yt = y_train.cpu().detach().numpy()
                    probs = prediction.probs.cpu().detach().numpy()
                    if metric_to_optimise == "acc":
                        metric_value = sklearn.metrics.accuracy_score(y_true=yt, y_pred=probs.round())
                    elif metric_to_optimise == "auc":
                        metric_value = sklearn.metrics.roc_auc_score(y_true=yt, y_score=probs)
                    else:
                        metric_value = sklearn.metrics.f1_score(y_true=yt, y_pred=probs.round())
            better = (metric_to_optimise in ["nll", "mse", "mae", "loss"] and metric_value < best_metric_value) or (
                metric_to_optimise in ["acc", "auc", "f1"] and metric_value > best_metric_value
            )
            if better:  # :411-422
                best_metric_value = metric_value
                best_lr = step_size
                particles_out = deepcopy(particles_i.detach())
            if (
                i > 0
                and step_sizes[i - 1] in energy_potentials_history
                and abs(energy_potentials_history[step_sizes[i - 1]][-1] - energy_potentials[-1])
                / energy_potentials_history[step_sizes[i - 1]][-1]
                < minimum_change_in_energy_potential
            ):
                break  # :423-433
    return particles_out, best_lr, len(energy_potentials_history[best_lr])
