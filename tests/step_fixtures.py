"""Loader for tests/golden/oracle_step_vectors.npz (written by tests/golden/make_oracle_step_vectors.py): the frozen
Langevin step.  Builds the oracle's objects -- and, for the GPU tests, the library's -- from the stored inputs."""
import os

import numpy as np
import torch

from oracle import pls_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
PAIRS = ["gaussian/identity", "poisson/square", "bernoulli/sigmoid", "bernoulli/probit", "student_t/identity",
         "multimodal/identity"]
TAGS = ["a", "c1"]
BASES = ["onb", "ipb"]


def load():
    return dict(np.load(os.path.join(HERE, "golden", "oracle_step_vectors.npz")))


def t(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def rel(got, want):
    got, want = t(got.detach().cpu() if isinstance(got, torch.Tensor) else got), t(want)
    assert got.shape == want.shape, (got.shape, want.shape)
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-300)).item()


def oracle_costs(v, tag):
    y, yc, yb = t(v[f"{tag}/y"]), t(v[f"{tag}/y_count"]), t(v[f"{tag}/y_bin"])
    return {
        "gaussian/identity": O.GaussianCost(0.3, y, O.IdentityLink()),
        "poisson/square": O.PoissonCost(yc, O.SquareLink()),
        "bernoulli/sigmoid": O.BernoulliCost(yb, O.SigmoidLink()),
        "bernoulli/probit": O.BernoulliCost(yb, O.ProbitLink()),
        "student_t/identity": O.StudentTCost(3.0, y, O.IdentityLink(), 0.7),
        "multimodal/identity": O.MultiModalCost(0.7, 1.5, 0.3, y, O.IdentityLink()),
    }


def oracle_bases(v, tag):
    kern = O.RBFARDKernel(t(v[f"{tag}/ls"]), float(v[f"{tag}/scale"]))
    x, z, y = t(v[f"{tag}/x"]), t(v[f"{tag}/z"]), t(v[f"{tag}/y"])
    onb = O.OrthonormalBasis(kern, z, x, float(v[f"{tag}/threshold"]),
                             spectrum=(t(v[f"{tag}/spectrum_values"]), t(v[f"{tag}/spectrum_vectors"])))
    ipb = O.InducingPointBasis(kern, z, y[: z.shape[0]], x)
    return {"onb": onb, "ipb": ipb}


def gpu_costs(P, v, tag):
    y, yc, yb = t(v[f"{tag}/y"]), t(v[f"{tag}/y_count"]), t(v[f"{tag}/y_bin"])
    C, Lk = P.costs, P.links
    return {
        "gaussian/identity": C.GaussianCost(0.3, y, Lk.IdentityLinkFunction()),
        "poisson/square": C.PoissonCost(yc, Lk.SquareLinkFunction()),
        "bernoulli/sigmoid": C.BernoulliCost(yb, Lk.SigmoidLinkFunction()),
        "bernoulli/probit": C.BernoulliCost(yb, Lk.ProbitLinkFunction()),
        "student_t/identity": C.StudentTCost(3.0, y, Lk.IdentityLinkFunction(), 0.7),
        "multimodal/identity": C.MultiModalCost(0.7, 1.5, 0.3, y, Lk.IdentityLinkFunction()),
    }


def gpu_bases(P, v, tag):
    x, z, y = t(v[f"{tag}/x"]), t(v[f"{tag}/z"]), t(v[f"{tag}/y"])
    kern = P.pkg.PLSKernel(P.pkg.ARDKernel(t(v[f"{tag}/ls"]), float(v[f"{tag}/scale"])), z)
    onb = P.basis.OrthonormalBasis(kern, z, x, float(v[f"{tag}/threshold"]),
                                   spectrum=(t(v[f"{tag}/spectrum_values"]), t(v[f"{tag}/spectrum_vectors"])), verbose=False)
    ipb = P.basis.InducingPointBasis(kern, z, y[: z.shape[0]], x)
    return {"onb": onb, "ipb": ipb}
