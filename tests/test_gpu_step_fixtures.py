"""GPU: the HIP path against the frozen oracle step vectors (tests/golden/oracle_step_vectors.npz), TOL 1e-9.

Same file the CPU test tests/test_oracle_step_fixtures.py holds the oracle to: U0, noise, eta -> F, G, dU, E for both
bases x the six native (cost, link) pairs at (512, 32, 64) and at BASELINE configs[0]'s shape, and the 200-step
configs[0] trajectory of train_pls with its early-stop index (basis/orthonormal.py:128-159, basis/inducing_point.py:117-150,
experiments/trainers.py:139-162)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import step_fixtures as SF
from test_gpu_ksplit import P, _f64_default  # noqa: F401
from test_gpu_parity import TOL, cu


@pytest.fixture(scope="module")
def V():
    return SF.load()


@pytest.mark.parametrize("tag", SF.TAGS)
@pytest.mark.parametrize("bname", SF.BASES)
def test_hip_step_reproduces_the_frozen_vectors(P, V, tag, bname):
    stride = int(V["row_stride"])
    basis, costs = SF.gpu_bases(P, V, tag)[bname], SF.gpu_costs(P, V, tag)
    u0, noise, eta = cu(SF.t(V[f"{tag}/{bname}/u0"])), cu(SF.t(V[f"{tag}/{bname}/noise"])), float(V[f"{tag}/eta"])
    assert u0.shape[0] == basis.approximation_dimension
    f = basis.calculate_untransformed_train_prediction_samples(u0)
    assert SF.rel(f[::stride], V[f"{tag}/{bname}/F"]) < TOL
    for name in SF.PAIRS:
        pls = P.pkg.PLS(basis, costs[name])
        assert SF.rel(costs[name].calculate_cost_derivative(f)[::stride], V[f"{tag}/{bname}/{name}/G"]) < TOL, name
        # the fused step (one launch chain) and the un-fused composition the reference's own methods spell out
        assert SF.rel(pls.calculate_particle_update(u0, eta, noise=noise), V[f"{tag}/{bname}/{name}/dU"]) < TOL, name
        g = costs[name].calculate_cost_derivative(f)
        assert SF.rel(basis.calculate_particle_update(u0, g, eta, noise=noise), V[f"{tag}/{bname}/{name}/dU"]) < TOL, name
        e = pls.calculate_energy_potential(u0)
        assert abs(e - float(V[f"{tag}/{bname}/{name}/E"])) <= TOL * abs(e), name


def test_hip_train_pls_reproduces_the_frozen_trajectory(P, V):
    onb = SF.gpu_bases(P, V, "c1")["onb"]
    gc = P.costs.GaussianCost(0.5, SF.t(V["c1/y"]), P.links.IdentityLinkFunction())
    u0, eta = SF.t(V["c1/train/u0"]), float(V["c1/eta"])
    noises = [cu(SF.t(n)) for n in V["c1/train/noises"]]
    ut, en = P.pkg.train_pls(P.pkg.PLS(onb, gc), cu(u0), len(noises), eta, 1e9, noises=noises)
    assert len(en) == 200 and SF.rel(ut, V["c1/train/particles"]) < 1e-8
    assert np.allclose(en, V["c1/train/energies"], rtol=TOL, atol=0)
    us, es = P.pkg.train_pls(P.pkg.PLS(onb, gc), cu(u0), len(noises), eta, float(V["c1/train/stop_patience"]), noises=noises)
    assert len(es) == len(V["c1/train/stop_energies"]), "stop index"
    assert SF.rel(us, V["c1/train/stop_particles"]) < 1e-8 and np.allclose(es, V["c1/train/stop_energies"], rtol=TOL, atol=0)
