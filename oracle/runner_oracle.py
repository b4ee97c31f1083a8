"""Sequential step-size search: CPU restatement of experiments/runners.py:331-446.  TEST INFRASTRUCTURE ONLY (the
checker of projected_langevin_sampling_amd.runners, which runs the candidates as column blocks of one launch).

Follows the reference statement by statement -- candidates in order of decreasing step size (:356-360), set_seed before
each (:364), train_pls from a clone of the initial particles (:365-372), a run counts if it has energies and finite
particles (:373), best-by-metric bookkeeping (:411-422), stop once two consecutive accepted runs agree on their final
energy (:423-433), return (particles, step size, number of energies) of the best run (:446).  Only the "loss" metric and
a caller-supplied ``metric_fn`` are restated (the others go through pls.predict, which is pinned elsewhere).

``noise_fn(t)`` supplies the step-t noise of a candidate's run; it is re-created per candidate by ``make_noise_fn()``
AFTER set_seed, so a stream keyed by torch's global generator behaves exactly like the reference's torch.normal draws."""
from __future__ import annotations

from copy import deepcopy

import numpy as np
import torch

from . import pls_oracle as O


def set_seed(seed: int) -> None:
    """src/utils.py:8-22 (the generators this path draws from)."""
    np.random.seed(seed)
    torch.manual_seed(seed)


def train_pls_runner(pls, particles, simulation_duration, maximum_number_of_steps, early_stopper_patience,
                     number_of_step_searches, step_size_upper, minimum_change_in_energy_potential, seed,
                     metric_to_optimise="loss", metric_fn=None, make_noise_fn=None):
    assert metric_to_optimise in ("nll", "mse", "mae", "loss"), "restated for the minimised metrics"
    best_metric_value = float("inf")  # :347-348
    best_lr = None
    energy_potentials_history = {}
    step_sizes = np.logspace(
        np.log10(step_size_upper), np.log10(simulation_duration / maximum_number_of_steps), number_of_step_searches
    )  # :356-360
    particles_out = particles.detach().clone()
    for i, step_size in enumerate(step_sizes):
        number_of_epochs = int(simulation_duration / step_size)  # :363
        set_seed(seed)  # :364
        noise_fn = make_noise_fn() if make_noise_fn is not None else None
        particles_i, energy_potentials = O.train_pls(
            pls, particles.detach().clone(), number_of_epochs, step_size, early_stopper_patience, noise_fn=noise_fn
        )  # :365-372
        if energy_potentials and torch.isfinite(particles_i).all():  # :373
            energy_potentials_history[step_size] = energy_potentials
            metric_value = energy_potentials[-1] if metric_to_optimise == "loss" else metric_fn(particles_i)  # :374-410
            if metric_value < best_metric_value:  # :411-422
                best_metric_value = metric_value
                best_lr = step_size
                particles_out = deepcopy(particles_i.detach())
            if (
                i > 0
                and step_sizes[i - 1] in energy_potentials_history
                and abs(energy_potentials_history[step_sizes[i - 1]][-1] - energy_potentials[-1])
                / energy_potentials_history[step_sizes[i - 1]][-1]
                < minimum_change_in_energy_potential
            ):  # :423-433
                break
    return particles_out, best_lr, len(energy_potentials_history[best_lr]), energy_potentials_history
