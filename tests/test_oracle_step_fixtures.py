"""CPU: the oracle against its own frozen step vectors (tests/golden/oracle_step_vectors.npz).

The reference's tests do not pin `_calculate_particle_update` of the real bases (basis/orthonormal.py:128-159,
basis/inducing_point.py:117-150) nor `train_pls` (experiments/trainers.py:139-162); the oracle's building blocks are
tied to the reference's goldens in tests/test_oracle_goldens.py, and THESE vectors freeze what it makes of a whole step,
so the oracle cannot drift together with the kernels it checks (tests/test_gpu_step_fixtures.py holds the HIP path to the
same file).  Tolerance 1e-13 of the largest entry for one step (the arithmetic is re-run, BLAS may re-associate), 1e-11
for the 200-step trajectory."""
import numpy as np
import pytest
import torch

import step_fixtures as SF
from oracle import pls_oracle as O


@pytest.fixture(scope="module")
def V():
    return SF.load()


@pytest.fixture(autouse=True)
def _f64():
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(prev)


@pytest.mark.parametrize("tag", SF.TAGS)
@pytest.mark.parametrize("bname", SF.BASES)
def test_oracle_step_reproduces_the_frozen_vectors(V, tag, bname):
    stride = int(V["row_stride"])
    basis, costs = SF.oracle_bases(V, tag)[bname], SF.oracle_costs(V, tag)
    u0, noise, eta = SF.t(V[f"{tag}/{bname}/u0"]), SF.t(V[f"{tag}/{bname}/noise"]), float(V[f"{tag}/eta"])
    assert u0.shape[0] == basis.approximation_dimension
    f = basis.calculate_untransformed_train_prediction_samples(u0)
    assert SF.rel(f[::stride], V[f"{tag}/{bname}/F"]) < 1e-13
    for name in SF.PAIRS:
        pls = O.PLS(basis, costs[name])
        assert SF.rel(costs[name].calculate_cost_derivative(f)[::stride], V[f"{tag}/{bname}/{name}/G"]) < 1e-13, name
        assert SF.rel(pls.calculate_particle_update(u0.clone(), eta, noise=noise), V[f"{tag}/{bname}/{name}/dU"]) < 1e-13, name
        e = pls.calculate_energy_potential(u0.clone())
        assert abs(e - float(V[f"{tag}/{bname}/{name}/E"])) <= 1e-13 * abs(e), name


def test_oracle_train_pls_reproduces_the_frozen_trajectory(V):
    """BASELINE configs[0] (N = 100, M = 10, J = 64, Gaussian cost, eta = 1e-3): 200 steps, and the early-stopped run"""
    onb = SF.oracle_bases(V, "c1")["onb"]
    gc = O.GaussianCost(0.5, SF.t(V["c1/y"]), O.IdentityLink())
    u0, eta = SF.t(V["c1/train/u0"]), float(V["c1/eta"])
    noises = [SF.t(n) for n in V["c1/train/noises"]]
    ut, en = O.train_pls(O.PLS(onb, gc), u0.clone(), len(noises), eta, 1e9, noises=noises)
    assert len(en) == len(V["c1/train/energies"]) == 200
    assert SF.rel(ut, V["c1/train/particles"]) < 1e-11 and np.allclose(en, V["c1/train/energies"], rtol=1e-11, atol=0)
    us, es = O.train_pls(O.PLS(onb, gc), u0.clone(), len(noises), eta, float(V["c1/train/stop_patience"]), noises=noises)
    assert len(es) == len(V["c1/train/stop_energies"]), "stop index"
    assert SF.rel(us, V["c1/train/stop_particles"]) < 1e-11 and np.allclose(es, V["c1/train/stop_energies"], rtol=1e-11, atol=0)


def test_the_fixture_holds_what_its_generator_says(V):
    assert int(V["row_stride"]) == 8
    for tag, (n, m, j) in (("a", (512, 32, 64)), ("c1", (100, 10, 64))):
        assert V[f"{tag}/x"].shape[0] == n and V[f"{tag}/z"].shape[0] == m and V[f"{tag}/ipb/u0"].shape == (m, j)
        for b in SF.BASES:
            assert V[f"{tag}/{b}/F"].shape == ((n + 7) // 8, j)
            for name in SF.PAIRS:
                assert np.isfinite(V[f"{tag}/{b}/{name}/dU"]).all() and V[f"{tag}/{b}/{name}/G"].shape == ((n + 7) // 8, j)
