"""Checkpoint format of the reference, read and written unchanged.

The reference stores a trained sampler as ``torch.save({"particles", "observation_noise", "best_lr",
"number_of_epochs"}, path)`` (experiments/uci/regression/main.py:300-308) and restores it with ``load_pls``
(experiments/loaders.py:10-28).  Files written by either implementation load in the other.  Two optional extra keys
make a J-sharded run exactly resumable: ``noise_step`` (the step counter of the Philox stream) and
``number_of_particles`` (the global J a shard belongs to); the reference ignores unknown keys.

Particles of the orthonormal basis are COORDINATES in the eigenvector gauge of the basis they were trained with
(orthonormal.py:46-68), and an eigendecomposition is defined only up to signs and rotations inside clusters: the file
therefore also records ``spectrum_fingerprint`` (basis/spectrum.py: kept eigenvalues, projections of a probe vector, a
hash of the exact bits), and ``load_pls`` refuses particles whose gauge is not the one of the basis it restores them
into -- otherwise the resumed run would silently continue from another function."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .kernel import _dev
from .projected_langevin_sampling import PLS


def _basis_fingerprint(pls) -> Optional[dict]:
    fn = getattr(getattr(pls, "basis", None), "spectrum_fingerprint", None)
    return fn() if callable(fn) else None


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
    fingerprint = _basis_fingerprint(pls)
    if fingerprint is not None:
        state["spectrum_fingerprint"] = fingerprint
    torch.save(state, model_path)


def load_pls(pls: PLS, model_path: str, on_gauge_mismatch: str = "raise") -> Tuple[PLS, torch.Tensor, Optional[float], Optional[int]]:
    """experiments/loaders.py:10-28: restores the particles (float64, on the MI355X) and the observation noise.
    ``on_gauge_mismatch``: "raise" (default), "warn" or "ignore" when the file's ``spectrum_fingerprint`` does not describe
    ``pls.basis`` (files of the reference carry none and load as before)."""
    assert on_gauge_mismatch in ("raise", "warn", "ignore")
    model_config = torch.load(model_path, map_location="cpu")
    saved, current = model_config.get("spectrum_fingerprint"), _basis_fingerprint(pls)
    if saved is not None and current is not None and on_gauge_mismatch != "ignore":
        from .basis.spectrum import compare_fingerprints

        reason = compare_fingerprints(saved, current)
        if reason is not None:
            message = (f"{model_path}: {reason}.  Rebuild the basis with the spectrum the run was trained with "
                       "(OrthonormalBasis(spectrum=...)) or with the same eigh_device.")
            if on_gauge_mismatch == "raise":
                raise ValueError(message)
            import warnings

            warnings.warn(message)
    particles = _dev(model_config["particles"])
    pls.observation_noise = model_config["observation_noise"]
    print(f"Loaded particles and observation_noise from {model_path=}.")
    best_lr = model_config.get("best_lr")
    number_of_epochs = model_config.get("number_of_epochs")
    return pls, particles, best_lr, number_of_epochs
