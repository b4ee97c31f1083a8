"""Checkpoint format of the reference, read and written unchanged.

The reference stores a trained sampler as ``torch.save({"particles", "observation_noise", "best_lr",
"number_of_epochs"}, path)`` (experiments/uci/regression/main.py:300-308) and restores it with ``load_pls``
(experiments/loaders.py:10-28).  Files written by either implementation load in the other.  Two optional extra keys
make a J-sharded run exactly resumable: ``noise_step`` (the step counter of the Philox stream) and
``number_of_particles`` (the global J a shard belongs to); the reference ignores unknown keys."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .kernel import _dev
from .projected_langevin_sampling import PLS


def save_pls(pls: PLS, particles: torch.Tensor, model_path: str, best_lr: Optional[float] = None,
             number_of_epochs: Optional[int] = None, noise_step: Optional[int] = None,
             number_of_particles: Optional[int] = None) -> None:
    """Same dictionary as experiments/uci/regression/main.py:300-308; particles are stored on the CPU so the file
    loads on any machine (the reference's loader maps them back to the GPU when one is present)."""
    state = {
        "particles": particles.detach().cpu(),
        "observation_noise": pls.observation_noise,
        "best_lr": best_lr,
        "number_of_epochs": number_of_epochs,
    }
    if noise_step is not None:
        state["noise_step"] = int(noise_step)
    if number_of_particles is not None:
        state["number_of_particles"] = int(number_of_particles)
    torch.save(state, model_path)


def load_pls(pls: PLS, model_path: str) -> Tuple[PLS, torch.Tensor, Optional[float], Optional[int]]:
    """experiments/loaders.py:10-28: restores the particles (float64, on the MI355X) and the observation noise."""
    model_config = torch.load(model_path, map_location="cpu")
    particles = _dev(model_config["particles"])
    pls.observation_noise = model_config["observation_noise"]
    print(f"Loaded particles and observation_noise from {model_path=}.")
    best_lr = model_config.get("best_lr")
    number_of_epochs = model_config.get("number_of_epochs")
    return pls, particles, best_lr, number_of_epochs
