"""CPU oracle for the projected-Langevin-sampling hot path.  TEST INFRASTRUCTURE ONLY.

This file is a torch-CPU restatement of the reference's algorithm for one Langevin step
(SURVEY.md section 8a).  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``projected-langevin-sampling_amd/``) never imports anything from ``oracle/``.

Why a restatement: the reference package cannot be imported here because ``gpytorch`` /
``linear_operator`` (pinned 1.15.2 / 0.6.1 in the reference's ``uv.lock``) are not
installed and there is no network.  Every function below cites the reference file:line
(paths relative to ``/root/reference``) it follows op for op.

Pinning (tests/test_oracle_goldens.py):
  * the literal golden tensors of the reference's own unit tests (tests/test_basis.py,
    tests/test_costs.py, tests/test_samplers.py, tests/test_pls_kernel.py), copied as
    data into tests/golden/reference_unit_goldens.json;
  * vectors produced HERE by executing the reference's own gpytorch-free source files
    (link_functions.py, costs/{base,poisson,bernoulli,multimodal}.py, src/samplers.py),
    script: tests/golden/make_reference_vectors.py.
  * NOT pinned by any reference test: ``_calculate_particle_update`` of either real basis
    and the RBF/ARD base kernel values (third party gpytorch.kernels.RBFKernel; restated
    from its published closed form) -> those two are "parity unpinned" beyond their
    building blocks (the kernel values are cross-checked against scikit-learn's
    ConstantKernel * RBF in tests/test_oracle_goldens.py: an independent implementation
    of the same closed form, not the reference's library).

Third-party arithmetic restated here:
  * ``gpytorch.solve(input=K, rhs=U[, lhs=L])`` == ``L @ K^{-1} @ U`` through a Cholesky
    factorisation (numerically reproduces the reference's IPB goldens at M=2).
  * ``ScaleKernel(RBFKernel(ard_num_dims=D))``: k(a,b) = s * exp(-0.5 * sum_d ((a_d-b_d)/l_d)^2).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# base kernels (third party in the reference)
# --------------------------------------------------------------------------------------


class LinearKernel:
    """mockers/kernel.py:13-23 (MockKernel): k(x1, x2) = x1 @ x2^T."""

    def __call__(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        return x1 @ x2.transpose(-1, -2)


class RBFARDKernel:
    """gpytorch ScaleKernel(RBFKernel(ard_num_dims=D)) as constructed at
    experiments/uci/regression/main.py:171-173 and README.md:144-146.
    k(a, b) = outputscale * exp(-0.5 * sum_d ((a_d - b_d) / lengthscale_d)^2).
    Restated from the published closed form (parity unpinned: no reference test evaluates it).
    """

    def __init__(self, lengthscale, outputscale: float = 1.0):
        self.lengthscale = torch.as_tensor(lengthscale, dtype=torch.float64).reshape(-1)
        self.outputscale = float(outputscale)

    def __call__(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        ls = self.lengthscale.to(x1.dtype)
        a = x1 / ls
        b = x2 / ls
        d2 = (a[:, None, :] - b[None, :, :]).square().sum(-1)
        return self.outputscale * torch.exp(-0.5 * d2)


def pls_kernel_r(
    base_kernel: Callable,
    approximation_samples: torch.Tensor,
    x1: torch.Tensor,
    x2: torch.Tensor,
    additional_approximation_samples: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """src/projected_langevin_sampling/kernel.py:31-76 (PLSKernel.forward):
    r(x1, x2) = (1/n_S) k(x1, S) k(x2, S)^T with S = unique rows of Z (+ extra)."""
    samples = [approximation_samples]
    if additional_approximation_samples is not None:
        samples.append(additional_approximation_samples)
    s = torch.cat(samples, dim=0).unique(dim=0)  # kernel.py:43-45
    g1 = base_kernel(x1, s)
    g2 = base_kernel(x2, s)
    return (1.0 / s.shape[0]) * (g1 @ g2.T)  # kernel.py:69-72


# --------------------------------------------------------------------------------------
# samplers
# --------------------------------------------------------------------------------------


def sample_multivariate_normal(
    mean: torch.Tensor,
    cov: torch.Tensor,
    size: Optional[Tuple[int, ...]] = None,
    seed: Optional[int] = None,
) -> torch.Tensor:
    """src/samplers.py:6-44: eigh(cov), clip eigenvalues at 0, Q sqrt(L) xi with
    xi = torch.normal on the CPU generator (global one if seed is None)."""
    generator = torch.Generator().manual_seed(seed) if seed is not None else None
    size = (1,) if not size else size
    eigenvalues, eigenvectors = torch.linalg.eigh(cov)
    eigenvalues = torch.clip(eigenvalues, 0, None)
    normal_sample = torch.normal(
        mean=0.0, std=1.0, size=(eigenvalues.shape[0], *size), generator=generator
    )
    return torch.real(
        mean[:, None] + eigenvectors @ torch.diag(torch.sqrt(eigenvalues)) @ normal_sample
    ).T


def initialise_particles_noise(
    approximation_dimension: int,
    number_of_particles: int,
    seed: Optional[int] = None,
    mean: float = 0.0,
    stdev: float = 1.0,
) -> torch.Tensor:
    """basis/base.py:39-63."""
    generator = torch.Generator().manual_seed(seed) if seed is not None else None
    return torch.normal(
        mean=mean,
        std=stdev,
        size=(approximation_dimension, number_of_particles),
        generator=generator,
    )


# --------------------------------------------------------------------------------------
# link functions  (link_functions.py:30-80)
# --------------------------------------------------------------------------------------


class IdentityLink:
    name = "identity"

    def __call__(self, y):
        return y  # link_functions.py:54-55


class SquareLink:
    name = "square"

    def __call__(self, y):
        return torch.square(y)  # link_functions.py:79-80


class SigmoidLink:
    name = "sigmoid"

    def __init__(self, jitter: float = 1e-10):
        self.jitter = jitter

    def __call__(self, y):
        # link_functions.py:67-70
        return torch.clip(torch.reciprocal(1 + torch.exp(-y)), self.jitter, 1 - self.jitter)


class ProbitLink:
    name = "probit"

    def __init__(self, jitter: float = 1e-10):
        self.jitter = jitter

    def __call__(self, y):
        # link_functions.py:39-45 (note: sqrt(2) is evaluated in float32 by torch.tensor(2.0))
        return torch.clip(
            (1 + torch.erf(y / torch.sqrt(torch.tensor(2.0)))) / 2,
            self.jitter,
            1 - self.jitter,
        )


# --------------------------------------------------------------------------------------
# costs  (costs/*.py)
# --------------------------------------------------------------------------------------


class _Cost:
    observation_noise: Optional[float] = None

    def calculate_cost(self, f: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _autograd(self, f: torch.Tensor) -> torch.Tensor:
        """costs/base.py:68-84: vmap(jacfwd(cost)) over particles.  Restated with reverse-mode
        autograd of sum_j c_j, which equals the Jacobian diagonal blocks because c_j only
        depends on column j."""
        f = f.detach().clone().requires_grad_(True)
        c = self.calculate_cost(f)
        (g,) = torch.autograd.grad(c.sum(), f)
        return g.detach()


class GaussianCost(_Cost):
    """costs/gaussian.py:54-110.  observation_noise is used as a VARIANCE here (:71, :86)."""

    def __init__(self, observation_noise: float, y_train: torch.Tensor, link_function):
        self.observation_noise = observation_noise
        self.y_train = y_train
        self.link_function = link_function

    def calculate_cost(self, f):
        p = self.link_function(f)
        errors = (p - self.y_train[:, None]).T  # (J, N)  gaussian.py:68
        return (1 / (2 * self.observation_noise)) * (errors * errors).sum(dim=1)  # :71-73

    def calculate_cost_derivative(self, f, force_autograd: bool = False):
        if isinstance(self.link_function, IdentityLink) and not force_autograd:
            p = self.link_function(f)
            return (1 / self.observation_noise) * (p - self.y_train[:, None])  # :86-88
        return self._autograd(f)


class PoissonCost(_Cost):
    """costs/poisson.py:47-104."""

    def __init__(self, y_train: torch.Tensor, link_function):
        self.y_train = y_train
        self.link_function = link_function

    def calculate_cost(self, f):
        p = self.link_function(f)
        return (-2 * torch.multiply(self.y_train[:, None], torch.log(torch.abs(f))) + p).sum(
            dim=0
        )  # poisson.py:59-66

    def calculate_cost_derivative(self, f, force_autograd: bool = False):
        if isinstance(self.link_function, SquareLink) and not force_autograd:
            return -2 * torch.divide(self.y_train[:, None], f) + 2 * f  # poisson.py:76-82
        return self._autograd(f)


class BernoulliCost(_Cost):
    """costs/bernoulli.py:48-99 (y_train cast to double at :32)."""

    def __init__(self, y_train: torch.Tensor, link_function):
        self.y_train = y_train.type(torch.double)
        self.link_function = link_function

    def calculate_cost(self, f):
        p = self.link_function(f)
        return -torch.log(p).T @ self.y_train - torch.log(1 - p).T @ (1 - self.y_train)  # :60-62

    def calculate_cost_derivative(self, f, force_autograd: bool = False):
        if isinstance(self.link_function, SigmoidLink) and not force_autograd:
            p = self.link_function(f)
            return -torch.mul(self.y_train[:, None], 1 - p) + torch.mul(
                1 - self.y_train[:, None], p
            )  # :75-77
        return self._autograd(f)


class StudentTCost(_Cost):
    """costs/student_t.py:55-110."""

    def __init__(self, degrees_of_freedom: float, y_train, link_function, scale: float = 1.0):
        self.degrees_of_freedom = degrees_of_freedom
        self.y_train = y_train
        self.link_function = link_function
        self.scale = scale

    def calculate_cost(self, f):
        p = self.link_function(f)
        errors = (p - self.y_train[:, None]).T
        return (
            0.5
            * (self.degrees_of_freedom + 1)
            * torch.log(
                1 + torch.square(errors) / (self.degrees_of_freedom * (self.scale**2))
            ).sum(dim=1)
        )  # :66-72

    def calculate_cost_derivative(self, f, force_autograd: bool = False):
        if isinstance(self.link_function, IdentityLink) and not force_autograd:
            errors = self.link_function(f) - self.y_train[:, None]
            return (self.degrees_of_freedom + 1) * torch.divide(
                errors, (self.degrees_of_freedom * (self.scale**2) + torch.square(errors))
            )  # :85-88
        return self._autograd(f)


class MultiModalCost(_Cost):
    """costs/multimodal.py:37-91.  observation_noise is used as a STD here (:56, :62)."""

    def __init__(self, observation_noise, shift, bernoulli_noise, y_train, link_function):
        self.observation_noise = observation_noise
        self.shift = shift
        self.bernoulli_noise = bernoulli_noise
        self.y_train = y_train
        self.link_function = link_function

    def calculate_cost(self, f):
        p = self.link_function(f)
        e1 = self.y_train[:, None] - p + self.shift
        e2 = self.y_train[:, None] - p
        norm = torch.log(torch.sqrt(2 * torch.tensor([torch.pi]) * (self.observation_noise**2)))
        ll1 = -0.5 * (torch.square(e1) / (self.observation_noise**2)) - norm
        ll2 = -0.5 * (torch.square(e2) / (self.observation_noise**2)) - norm
        return -torch.logsumexp(
            torch.stack(
                [
                    torch.log(torch.tensor(self.bernoulli_noise)) + ll1,
                    torch.log(torch.tensor(1 - self.bernoulli_noise)) + ll2,
                ]
            ),
            dim=0,
        ).sum(axis=0)  # :67-77

    def calculate_cost_derivative(self, f, force_autograd: bool = True):
        return self._autograd(f)  # multimodal.py:79-91: ALWAYS autograd


# --------------------------------------------------------------------------------------
# bases
# --------------------------------------------------------------------------------------


def _chol_solve(k: torch.Tensor, rhs: torch.Tensor) -> torch.Tensor:
    """gpytorch.solve(input=k, rhs=rhs) restated as a Cholesky solve (see module docstring)."""
    chol = torch.linalg.cholesky(k)
    return torch.cholesky_solve(rhs, chol)


class OrthonormalBasis:
    """basis/orthonormal.py:22-159."""

    def __init__(self, base_kernel, x_induce, x_train, eigenvalue_threshold: float = 0.0, r_kernel=None, spectrum=None):
        self.base_kernel = base_kernel
        self.x_induce = x_induce
        # r(x1, x2, extra approximation samples): kernel.py:31-76 unless a test double is handed in
        self.r_kernel = r_kernel or (lambda x1, x2, extra: pls_kernel_r(base_kernel, x_induce, x1, x2, extra))
        self.base_gram_induce = base_kernel(x_induce, x_induce)  # :36-38
        self.base_gram_induce_train = base_kernel(x_induce, x_train)  # :39-41
        if spectrum is None:
            self.eigenvalues, self.eigenvectors = torch.linalg.eigh(
                (1 / self.x_induce.shape[0]) * self.base_gram_induce
            )  # :46-48
        else:  # (not in the reference) a frozen eigendecomposition of k(Z,Z)/M: tests/golden/oracle_step_vectors.npz pins the
            # eigenvector gauge (signs, rotations inside clusters), which LAPACK is free to choose differently per version
            self.eigenvalues, self.eigenvectors = (torch.as_tensor(t, dtype=torch.float64) for t in spectrum)
        idx = torch.where(self.eigenvalues > eigenvalue_threshold)[0]  # :52
        self.eigenvalues = self.eigenvalues[idx].real
        self.eigenvectors = self.eigenvectors[:, idx].real
        self.scaled_eigenvectors = torch.multiply(
            torch.reciprocal(torch.sqrt(self.approximation_dimension * self.eigenvalues))[None, :],
            self.eigenvectors,
        )  # :63-68  (note M_k, not M)

    @property
    def approximation_dimension(self) -> int:
        return self.eigenvalues.shape[0]  # :70-76

    def initialise_particles(self, number_of_particles, noise_only=True, seed=None):
        if not noise_only:
            raise ValueError("For ONB base, noise_only must be True.")  # :91-92
        return initialise_particles_noise(self.approximation_dimension, number_of_particles, seed)

    def calculate_untransformed_train_prediction_samples(self, particles):
        # :106-108, left-to-right association
        return self.base_gram_induce_train.T @ self.scaled_eigenvectors @ particles

    def calculate_energy_potential(self, particles, cost) -> float:
        e = cost + 1 / 2 * torch.multiply(
            particles, torch.diag(torch.reciprocal(self.eigenvalues)) @ particles
        ).sum(dim=0)  # :120-125
        return e.mean().item()

    def sample_update_noise(self, particles) -> torch.Tensor:
        # :141-145 (global CPU generator, default dtype, eigh(I) every step)
        return sample_multivariate_normal(
            mean=torch.zeros(particles.shape[0]),
            cov=torch.eye(particles.shape[0]),
            size=(particles.shape[1],),
        ).T

    def calculate_particle_update(self, particles, cost_derivative, step_size, noise=None):
        assert particles.shape[0] == self.approximation_dimension  # base.py:156-158
        if noise is None:
            noise = self.sample_update_noise(particles)
        return (
            -step_size * self.scaled_eigenvectors.T @ self.base_gram_induce_train @ cost_derivative
            - step_size * torch.diag(torch.reciprocal(self.eigenvalues)) @ particles
            + math.sqrt(2.0 * step_size) * noise
        )  # :151-158


    def sample_predictive_noise(self, particles, x):
        """orthonormal.py:161-214 (without the optional additional noise distribution)."""
        gram_x = self.r_kernel(x, x, x)  # :174-178
        base_gram_x_induce = self.base_kernel(x, self.x_induce)  # :179-182
        off = base_gram_x_induce @ self.scaled_eigenvectors @ torch.diag(self.eigenvalues)  # :183-185
        cov = torch.concatenate(
            [
                torch.concatenate([torch.diag(self.eigenvalues), off.T], dim=1),
                torch.concatenate([off, gram_x], dim=1),
            ],
            dim=0,
        )  # :186-204
        return sample_multivariate_normal(
            mean=torch.zeros(cov.shape[0]), cov=cov, size=(particles.shape[1],)
        ).T  # :205-209

    def predict_untransformed_samples(self, particles, x, noise=None):
        """orthonormal.py:216-244."""
        base_gram_x_induce = self.base_kernel(x, self.x_induce)
        if noise is None:
            noise = self.sample_predictive_noise(particles, x)
        mk = self.approximation_dimension
        return noise[mk:, :] + (base_gram_x_induce @ self.scaled_eigenvectors @ (particles - noise[:mk, :]))


class InducingPointBasis:
    """basis/inducing_point.py:23-240."""

    def __init__(self, base_kernel, x_induce, y_induce, x_train, r_kernel=None):
        self.base_kernel = base_kernel
        self.x_induce = x_induce
        self.y_induce = y_induce
        self.r_kernel = r_kernel or (lambda x1, x2, extra: pls_kernel_r(base_kernel, x_induce, x1, x2, extra))
        self.gram_induce = self.r_kernel(x_induce, x_induce, None)  # :38-40
        self.base_gram_induce = base_kernel(x_induce, x_induce)  # :41-43
        self.base_gram_induce_train = base_kernel(x_induce, x_train)  # :44-46

    @property
    def approximation_dimension(self) -> int:
        return self.x_induce.shape[0]  # :52-58

    def initialise_particles(self, number_of_particles, noise_only=True, seed=None):
        noise = initialise_particles_noise(self.approximation_dimension, number_of_particles, seed)
        return noise if noise_only else (self.y_induce[:, None] + noise)  # :77-79

    def calculate_untransformed_train_prediction_samples(self, particles):
        # :89-93  k(X,Z) k(Z,Z)^{-1} U
        return self.base_gram_induce_train.T @ _chol_solve(self.base_gram_induce, particles)

    def calculate_energy_potential(self, particles, cost) -> float:
        v = _chol_solve(self.base_gram_induce, particles)  # :104-106
        e = cost + self.approximation_dimension / 2 * torch.square(v).sum(dim=0)  # :109-114
        return e.mean().item()

    def sample_update_noise(self, particles) -> torch.Tensor:
        # :133-137: e ~ N(0, k(Z,Z)) through eigh every step
        return sample_multivariate_normal(
            mean=torch.zeros(particles.shape[0]),
            cov=self.base_gram_induce,
            size=(particles.shape[1],),
        ).T

    def calculate_particle_update(self, particles, cost_derivative, step_size, noise=None):
        assert particles.shape[0] == self.approximation_dimension
        v = _chol_solve(self.base_gram_induce, particles)  # :130-132
        if noise is None:
            noise = self.sample_update_noise(particles)
        return (
            -step_size * self.base_gram_induce_train @ cost_derivative
            - step_size * self.approximation_dimension * v
            + math.sqrt(2.0 * step_size) * noise
        )  # :143-149


    def sample_predictive_noise(self, particles, x):
        """inducing_point.py:152-202 (without the optional additional noise distribution)."""
        gram_x = self.r_kernel(x, x, x)
        gram_induce_x = self.r_kernel(self.x_induce, x, x)
        cov = torch.concatenate(
            [
                torch.concatenate([self.gram_induce, gram_induce_x], dim=1),
                torch.concatenate([gram_induce_x.T, gram_x], dim=1),
            ],
            dim=0,
        )
        return sample_multivariate_normal(
            mean=torch.zeros(cov.shape[0]), cov=cov, size=(particles.shape[1],)
        ).T

    def predict_untransformed_samples(self, particles, x, noise=None):
        """inducing_point.py:204-240: G(x) + r(x,Z) r(Z,Z)^-1 (U - G(Z))."""
        gram_x_induce = self.r_kernel(x, self.x_induce, x)
        gram_induce = self.r_kernel(self.x_induce, self.x_induce, x)
        if noise is None:
            noise = self.sample_predictive_noise(particles, x)
        m = self.approximation_dimension
        return noise[m:, :] + gram_x_induce @ _chol_solve(gram_induce, particles - noise[:m, :])


def cost_sample_observation_noise(observation_noise, number_of_particles, seed=None):
    """costs/base.py:86-115."""
    if observation_noise is None:
        return torch.zeros(number_of_particles)
    generator = torch.Generator().manual_seed(seed) if seed is not None else None
    return torch.normal(mean=0.0, std=observation_noise, size=(number_of_particles,), generator=generator).flatten()


def cost_predict_samples(link_function, untransformed_samples, observation_noise):
    """costs/base.py:117-133: link(f + eps_j)."""
    return link_function(untransformed_samples + observation_noise[None, :])


def gaussian_predict_moments(prediction_samples):
    """costs/gaussian.py:40-52: mean and (unbiased) variance over the particle axis."""
    return prediction_samples.mean(dim=1), prediction_samples.var(axis=1)


def temper_scale(y_calibration, mean, variance) -> float:
    """temper/base.py:30-46: 2 * mean((y - m)^2 / sigma^2)."""
    return 2 * torch.mean(torch.div(torch.square(y_calibration - mean), variance)).item()


def conformal_uncalibrated(samples, coverage):
    """conformalise/pls.py:24-45: per-x quantiles 0.5 -/+ coverage/2 over the particle axis."""
    return (torch.quantile(samples, q=0.5 - coverage / 2, dim=1), torch.quantile(samples, q=0.5 + coverage / 2, dim=1))


def conformal_predict_coverage(samples_fn, x_calibration, y_calibration, x, coverage):
    """conformalise/base.py:58-114 with samples_fn(x) -> (N*, J) prediction samples (conformalise/pls.py:36-41)."""
    n = x_calibration.shape[0]
    lo_c, up_c = conformal_uncalibrated(samples_fn(x_calibration), coverage)
    scores = torch.max(torch.stack([lo_c - y_calibration, y_calibration - up_c], dim=1), dim=1).values
    calibration = torch.quantile(scores, float(np.clip((n + 1) * coverage / n, 0.0, 1.0))).item()
    lo, up = conformal_uncalibrated(samples_fn(x), coverage)
    median = torch.quantile(samples_fn(x), q=0.5, dim=1)
    return (
        torch.min(torch.stack([lo - calibration, median], dim=1), dim=1).values,
        torch.max(torch.stack([up + calibration, median], dim=1), dim=1).values,
    )


# --------------------------------------------------------------------------------------
# PLS facade + training loop
# --------------------------------------------------------------------------------------


class PLS:
    """projected_langevin_sampling.py:7-138."""

    def __init__(self, basis, cost):
        self.basis = basis
        self.cost = cost

    def calculate_cost(self, particles):
        f = self.basis.calculate_untransformed_train_prediction_samples(particles)  # :81-85
        return self.cost.calculate_cost(f)

    def calculate_cost_derivative(self, particles):
        f = self.basis.calculate_untransformed_train_prediction_samples(particles)  # :98-102
        return self.cost.calculate_cost_derivative(f)

    def calculate_particle_update(self, particles, step_size, noise=None):
        g = self.calculate_cost_derivative(particles)  # :118
        return self.basis.calculate_particle_update(particles, g, step_size, noise=noise)

    def calculate_energy_potential(self, particles) -> float:
        assert particles.shape[0] == self.basis.approximation_dimension  # :131-133
        return self.basis.calculate_energy_potential(particles, self.calculate_cost(particles))


class EarlyStopper:
    """experiments/early_stopper.py:4-24."""

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
    noises: Optional[List[torch.Tensor]] = None,
    noise_fn=None,
) -> Tuple[torch.Tensor, List[float]]:
    """experiments/trainers.py:139-162 (tqdm dropped).  ``noises[t]`` / ``noise_fn(t)`` injects the step-t noise."""
    energy_potentials: List[float] = []
    early_stopper = EarlyStopper(patience=early_stopper_patience)
    for t in range(number_of_epochs):
        step_noise = noise_fn(t) if noise_fn is not None else (None if noises is None else noises[t])
        update = pls.calculate_particle_update(
            particles, step_size, noise=step_noise
        )
        particles += update  # :157 in place
        energy = pls.calculate_energy_potential(particles)
        if early_stopper.should_stop(loss=energy, step_size=step_size):
            break
        energy_potentials.append(energy)
    return particles, energy_potentials
