"""Generate golden vectors by EXECUTING the reference's own gpytorch-free source files.

Run in the build container only (needs /root/reference):  python tests/golden/make_reference_vectors.py
Writes tests/golden/reference_vectors.npz  (data only: inputs and the reference's outputs).

The reference package ``src.projected_langevin_sampling`` cannot be imported normally because its
``__init__`` pulls in ``kernel.py`` -> ``gpytorch`` (not installed, no network).  The modules used
here import nothing from gpytorch themselves, so they are loaded as ordinary modules under empty
parent packages (the real ``__init__`` files are simply not executed; no stand-in library is made):
  src/samplers.py, src/projected_langevin_sampling/link_functions.py,
  src/projected_langevin_sampling/costs/{base,poisson,bernoulli,multimodal}.py
GaussianCost / StudentTCost / both bases / kernel.py import gpytorch at module level and stay
out of reach; they are pinned by the literal goldens in reference_unit_goldens.json instead.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"


def _empty_package(name: str, path: str) -> None:
    mod = types.ModuleType(name)
    mod.__path__ = [path]
    sys.modules[name] = mod


def load_reference_modules():
    sys.path.insert(0, REF)
    _empty_package("src", os.path.join(REF, "src"))
    _empty_package("src.projected_langevin_sampling", os.path.join(REF, "src/projected_langevin_sampling"))
    _empty_package(
        "src.projected_langevin_sampling.costs", os.path.join(REF, "src/projected_langevin_sampling/costs")
    )
    mods = {}
    for name in [
        "src.samplers",
        "src.projected_langevin_sampling.link_functions",
        "src.projected_langevin_sampling.costs.base",
        "src.projected_langevin_sampling.costs.poisson",
        "src.projected_langevin_sampling.costs.bernoulli",
        "src.projected_langevin_sampling.costs.multimodal",
    ]:
        mods[name.rsplit(".", 1)[-1]] = importlib.import_module(name)
    return mods


def main():
    torch.set_default_dtype(torch.float64)
    m = load_reference_modules()
    lf = m["link_functions"]
    out = {}
    g = torch.Generator().manual_seed(1234)
    n, j = 37, 11
    f = torch.randn(n, j, generator=g, dtype=torch.float64) * 1.7
    f_wide = torch.linspace(-30.0, 30.0, n * j, dtype=torch.float64).reshape(n, j)  # exercises the clips
    y_count = torch.poisson(torch.full((n,), 3.0), generator=g).to(torch.float64)
    y_bin = (torch.rand(n, generator=g) > 0.5).to(torch.float64)
    y_real = torch.randn(n, generator=g, dtype=torch.float64)
    out["f"] = f.numpy()
    out["f_wide"] = f_wide.numpy()
    out["y_count"] = y_count.numpy()
    out["y_bin"] = y_bin.numpy()
    out["y_real"] = y_real.numpy()

    for lname, link in [
        ("identity", lf.IdentityLinkFunction()),
        ("square", lf.SquareLinkFunction()),
        ("sigmoid", lf.SigmoidLinkFunction()),
        ("probit", lf.ProbitLinkFunction()),
    ]:
        out[f"link_{lname}"] = link(f).numpy()
        out[f"link_{lname}_wide"] = link(f_wide).numpy()

    links = {
        "identity": lf.IdentityLinkFunction,
        "square": lf.SquareLinkFunction,
        "sigmoid": lf.SigmoidLinkFunction,
        "probit": lf.ProbitLinkFunction,
    }
    # Poisson: closed form with square link, autograd with identity/square
    for lname in ["square", "identity"]:
        c = m["poisson"].PoissonCost(y_train=y_count, link_function=links[lname]())
        out[f"poisson_{lname}_cost"] = c.calculate_cost(f).numpy()
        out[f"poisson_{lname}_dcost"] = c.calculate_cost_derivative(f).numpy()
        out[f"poisson_{lname}_dcost_autograd"] = c.calculate_cost_derivative(f, force_autograd=True).numpy()
    # Bernoulli: sigmoid closed form + autograd, probit autograd; wide inputs hit the clip
    for lname in ["sigmoid", "probit"]:
        c = m["bernoulli"].BernoulliCost(y_train=y_bin, link_function=links[lname]())
        for tag, ff in [("", f), ("_wide", f_wide)]:
            out[f"bernoulli_{lname}_cost{tag}"] = c.calculate_cost(ff).numpy()
            out[f"bernoulli_{lname}_dcost{tag}"] = c.calculate_cost_derivative(ff).numpy()
            out[f"bernoulli_{lname}_dcost_autograd{tag}"] = c.calculate_cost_derivative(
                ff, force_autograd=True
            ).numpy()
    # MultiModal (always autograd); float32 constants inside (torch.tensor([pi])) are part of the reference
    c = m["multimodal"].MultiModalCost(
        observation_noise=0.7, shift=2.5, bernoulli_noise=0.3, y_train=y_real, link_function=links["identity"]()
    )
    out["multimodal_params"] = np.array([0.7, 2.5, 0.3])
    out["multimodal_identity_cost"] = c.calculate_cost(f).numpy()
    out["multimodal_identity_dcost"] = c.calculate_cost_derivative(f).numpy()

    # samplers: identity and a correlated covariance, seeded
    smp = m["samplers"]
    a = torch.randn(6, 6, generator=g, dtype=torch.float64)
    cov = a @ a.T / 6 + 0.1 * torch.eye(6)
    out["mvn_cov"] = cov.numpy()
    out["mvn_sample_seed7"] = smp.sample_multivariate_normal(
        mean=torch.zeros(6), cov=cov, size=(9,), seed=7
    ).numpy()
    out["mvn_eye_sample_seed7"] = smp.sample_multivariate_normal(
        mean=torch.zeros(6), cov=torch.eye(6), size=(9,), seed=7
    ).numpy()
    # observation-noise sampler from costs/base.py:86-115 (through PoissonCost -> zeros, MultiModal -> normal)
    out["obs_noise_multimodal_seed3"] = c.sample_observation_noise(number_of_particles=5, seed=3).numpy()

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
