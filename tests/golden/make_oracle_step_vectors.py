"""Freeze the Langevin step: oracle-generated vectors for tests/golden/oracle_step_vectors.npz.

SURVEY.md section 7 step 2 / row H: the reference's tests pin F, the costs, the kernels and the energies, but NOT
`_calculate_particle_update` of either real basis (basis/orthonormal.py:128-159, basis/inducing_point.py:117-150) nor
`train_pls` (experiments/trainers.py:139-162).  The building blocks of oracle/pls_oracle.py are tied to the reference's
own goldens (tests/test_oracle_goldens.py); this script freezes what the oracle makes of them for a whole step, so that
a simultaneous drift of oracle and kernels shows up:

  for the orthonormal and the inducing-point basis x every (cost, link) pair the reference dispatches natively, at
  (N, M, J) = (512, 32, 64) and at BASELINE configs[0]'s shape (100, 10, 64):  U0, noise, eta  ->  F, G, dU, E
  (F and G stored for every 8th training row), and the 200-step configs[0] trajectory of train_pls (Gaussian cost, noise
  injected) with the early-stop run next to it: final particles, every energy, the stop index.

The eigendecomposition of k(Z,Z)/M used by the orthonormal basis is stored too (`spectrum`): LAPACK is free to pick
another eigenvector gauge in another version, the vectors are not.  Everything is float64.

    python tests/golden/make_oracle_step_vectors.py          # rewrites the .npz next to this file
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pls_oracle as O  # noqa: E402

ROW_STRIDE = 8  # F and G are stored for rows 0, 8, 16, ...
PAIRS = ["gaussian/identity", "poisson/square", "bernoulli/sigmoid", "bernoulli/probit", "student_t/identity",
         "multimodal/identity"]


def costs(y, y_count, y_bin):
    """the oracle's cost objects of the six native (cost, link) pairs; tests build the library's from the same data"""
    return {
        "gaussian/identity": O.GaussianCost(0.3, y, O.IdentityLink()),
        "poisson/square": O.PoissonCost(y_count, O.SquareLink()),
        "bernoulli/sigmoid": O.BernoulliCost(y_bin, O.SigmoidLink()),
        "bernoulli/probit": O.BernoulliCost(y_bin, O.ProbitLink()),
        "student_t/identity": O.StudentTCost(3.0, y, O.IdentityLink(), 0.7),
        "multimodal/identity": O.MultiModalCost(0.7, 1.5, 0.3, y, O.IdentityLink()),
    }


def problem(tag):
    g = torch.Generator().manual_seed({"a": 20260401, "c1": 20260402}[tag])
    if tag == "a":
        n, m, j, d = 512, 32, 64, 3
        x = torch.rand(n, d, generator=g) * 2 - 1
        z = x[torch.randperm(n, generator=g)[:m]].clone()
        w = torch.randn(d, generator=g)
        fstar = torch.sin(2.0 * (x @ w))
        ls, scale = (0.5 + torch.rand(d, generator=g)) * 0.7, 1.3  # cond(k(Z,Z)) ~ 2e3
    else:  # BASELINE configs[0]: README.md:94-146 (1-D sin regression, M = 10 evenly spaced inducing points)
        n, m, j, d = 100, 10, 64, 1
        x = torch.linspace(-1, 1, n).reshape(-1, 1)
        z = x[:: n // m][:m].clone()
        fstar = torch.sin(2 * torch.pi * x.reshape(-1))
        ls, scale = torch.tensor([0.15]), 3.0
    y = fstar + 0.1 * torch.randn(n, generator=g)
    y_count = torch.poisson((2.0 * fstar) ** 2 + 0.5, generator=g)
    y_bin = (torch.rand(n, generator=g) < torch.sigmoid(2 * fstar)).double()
    return dict(n=n, m=m, j=j, d=d, x=x, z=z, y=y, y_count=y_count, y_bin=y_bin, ls=ls, scale=scale, gen=g)


def main():
    torch.set_default_dtype(torch.float64)
    out = {"row_stride": np.array(ROW_STRIDE)}
    for tag in ("a", "c1"):
        pr = problem(tag)
        g, m, j = pr["gen"], pr["m"], pr["j"]
        kern = O.RBFARDKernel(pr["ls"], pr["scale"])
        lam, vec = torch.linalg.eigh((1 / m) * kern(pr["z"], pr["z"]))
        thr = 1e-6 if tag == "a" else 0.0
        onb = O.OrthonormalBasis(kern, pr["z"], pr["x"], thr, spectrum=(lam, vec))
        ipb = O.InducingPointBasis(kern, pr["z"], pr["y"][:m], pr["x"])
        mk = onb.approximation_dimension
        eta = 1e-3
        # particles away from the pole of the Poisson / f^2 derivative (-2 y / f): start from noise around a smooth mean
        u_onb = 0.3 * torch.randn(mk, j, generator=g) * torch.sqrt(onb.eigenvalues)[:, None]
        u_onb = u_onb + torch.linalg.lstsq(onb.base_gram_induce_train.T @ onb.scaled_eigenvectors,
                                           (1.5 + 0.2 * pr["x"].sum(dim=1))[:, None]).solution
        u_ipb = 1.6 + 0.25 * torch.sin(pr["z"] @ torch.randn(pr["d"], j, generator=g)) + 0.02 * torch.randn(m, j, generator=g)
        xi = torch.randn(mk, j, generator=g)
        e_col = torch.linalg.cholesky(ipb.base_gram_induce + 1e-10 * torch.eye(m)) @ torch.randn(m, j, generator=g)
        for k in ("x", "z", "y", "y_count", "y_bin", "ls"):
            out[f"{tag}/{k}"] = pr[k].numpy()
        out[f"{tag}/scale"], out[f"{tag}/threshold"], out[f"{tag}/eta"] = np.array(pr["scale"]), np.array(thr), np.array(eta)
        out[f"{tag}/spectrum_values"], out[f"{tag}/spectrum_vectors"] = lam.numpy(), vec.numpy()
        out[f"{tag}/onb/u0"], out[f"{tag}/onb/noise"] = u_onb.numpy(), xi.numpy()
        out[f"{tag}/ipb/u0"], out[f"{tag}/ipb/noise"] = u_ipb.numpy(), e_col.numpy()
        cs = costs(pr["y"], pr["y_count"], pr["y_bin"])
        for bname, basis, u, nz in (("onb", onb, u_onb, xi), ("ipb", ipb, u_ipb, e_col)):
            f = basis.calculate_untransformed_train_prediction_samples(u)
            out[f"{tag}/{bname}/F"] = f[::ROW_STRIDE].numpy()
            for name in PAIRS:
                pls = O.PLS(basis, cs[name])
                gmat = cs[name].calculate_cost_derivative(f)
                assert torch.isfinite(gmat).all(), (tag, bname, name)
                out[f"{tag}/{bname}/{name}/G"] = gmat[::ROW_STRIDE].numpy()
                out[f"{tag}/{bname}/{name}/dU"] = pls.calculate_particle_update(u.clone(), eta, noise=nz).numpy()
                out[f"{tag}/{bname}/{name}/E"] = np.array(pls.calculate_energy_potential(u.clone()))
        if tag == "c1":  # train_pls, configs[0]: 200 steps, Gaussian cost (README.md:255-256), then the early-stop run
            steps = 200
            gc = O.GaussianCost(0.5, pr["y"], O.IdentityLink())
            u0 = O.initialise_particles_noise(mk, j, 0).double()
            noises = [torch.randn(mk, j, generator=g) for _ in range(steps)]
            ut, en = O.train_pls(O.PLS(onb, gc), u0.clone(), steps, eta, 1e9, noises=noises)
            us, es = O.train_pls(O.PLS(onb, gc), u0.clone(), steps, eta, 2.5 * eta, noises=noises)
            out["c1/train/u0"], out["c1/train/noises"] = u0.numpy(), torch.stack(noises).numpy()
            out["c1/train/particles"], out["c1/train/energies"] = ut.numpy(), np.array(en)
            out["c1/train/stop_patience"] = np.array(2.5 * eta)
            out["c1/train/stop_particles"], out["c1/train/stop_energies"] = us.numpy(), np.array(es)
            assert 0 < len(es) < steps, len(es)
    path = os.path.join(HERE, "oracle_step_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
