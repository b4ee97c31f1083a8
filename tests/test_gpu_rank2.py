"""GPU: the fused drift kernel for 129 .. 256 basis functions (csrc/small_rank2.h: the rank split over wave pairs, F
exchanged through LDS, the N x J intermediates never written) against plain torch fp64 on the host, against the CPU
oracle, and against the two-GEMM path it replaces (pls_set_option(PLS_OPT_SMALL_RANK2_MAX, 0)).
Reference: projected_langevin_sampling.py:107-123, basis/orthonormal.py:106-108, :128-159, costs/{*}.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pls_oracle as O
from test_gpu_ksplit import P, _f64_default  # noqa: F401
from test_gpu_parity import FUZZ_SEED, TOL, build_onb, cu, make_costs, make_problem, relerr


class rank2:
    """with rank2(P, limit): ...  -- every rank 129 .. limit through the wave-pair kernel for the block (0 = two-GEMM path)"""

    def __init__(self, P, limit):
        self.L, self.lib, self.limit = P.pkg._lib, P.pkg._lib.load(), limit

    def __enter__(self):
        L, lib = self.L, self.lib
        self.prev = (lib.pls_get_option(L.OPT_SMALL_RANK2_MIN), lib.pls_get_option(L.OPT_SMALL_RANK2_MAX))
        assert self.prev == (161, 0), self.prev  # (off by default since the row-block back-projection)
        L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MIN, 129))
        L.check(lib.pls_set_option(L.OPT_SMALL_RANK2_MAX, self.limit))

    def __exit__(self, *exc):
        self.L.check(self.lib.pls_set_option(self.L.OPT_SMALL_RANK2_MIN, self.prev[0]))
        self.L.check(self.lib.pls_set_option(self.L.OPT_SMALL_RANK2_MAX, self.prev[1]))
        return False


# every (KB0, KB1) instantiation: 16 * (KB0 + KB1) = 144 .. 256, exact multiples of 16 and ranks inside a block, odd ranks
# (a pair that straddles K), one rank per half-split parity; N on and off the 16-row tile grid, ragged column blocks
@pytest.mark.parametrize("mk,n,j", [(129, 4000, 200), (144, 3001, 64), (150, 2000, 130), (160, 40000, 2200), (161, 1000, 65),
                                    (176, 999, 64), (185, 2500, 100), (192, 5000, 128), (200, 700, 31), (208, 16, 64),
                                    (224, 3000, 64), (233, 1234, 77), (240, 2000, 64), (255, 3000, 96), (256, 4100, 192)])
def test_rank2_step_and_energy_against_the_host_product(P, mk, n, j):
    gen = torch.Generator().manual_seed(900 + mk + FUZZ_SEED)
    eta, s2 = 1e-3, 0.4
    a = torch.randn(mk, n, generator=gen) / mk ** 0.5
    lam = torch.rand(mk, generator=gen) + 0.5
    u = torch.randn(mk, j, generator=gen)
    xi = torch.randn(mk, j, generator=gen)
    y = torch.randn(n, generator=gen)
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam), poison_padding=True)
    gc = P.costs.GaussianCost(s2, y, P.links.IdentityLinkFunction())
    f = a.T @ u
    want = -eta * (a @ ((f - y[:, None]) / s2)) - eta * u / lam[:, None] + math.sqrt(2 * eta) * xi
    e_want = ((f - y[:, None]) ** 2).sum(0) / (2 * s2) + 0.5 * (u * u / lam[:, None]).sum(0)
    for limit in (256, 0):
        with rank2(P, limit):
            e_in = torch.empty(j, device="cuda")
            got = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True, input_energy=e_in)
            plain = basis.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
        assert relerr(got, want) < 1e-11 and relerr(plain, want) < 1e-11, (mk, limit)
        assert relerr(e_in, e_want) < 1e-11, (mk, limit)


@pytest.mark.parametrize("n,m,j,d", [(2100, 150, 260, 4), (1800, 200, 96, 3)])
def test_rank2_all_costs_against_the_oracle(P, n, m, j, d):
    """the six native (cost, link) pairs and the two autograd-only ones through the wave-pair kernel, against the oracle"""
    pr = make_problem(n, m, j, d, seed=n + m + FUZZ_SEED)
    ob, gb = build_onb(P, pr, threshold=0.0)
    mk = ob.approximation_dimension
    assert 128 < mk <= 256, mk
    scale = torch.sqrt(ob.eigenvalues)[:, None]
    u = (pr["u"][:mk] * scale).contiguous()
    # keep Poisson / f^2 away from its pole: particles around a smooth positive mean function
    u = u * 0.05 + torch.linalg.lstsq(ob.base_gram_induce_train.T @ ob.scaled_eigenvectors,
                                      (1.5 + 0.2 * pr["x"].sum(dim=1))[:, None]).solution
    xi = torch.randn(mk, j, generator=pr["gen"])
    checked = 0
    for name, oc, gc in make_costs(P, pr["y"], pr["fstar"], pr["gen"]):
        want = O.PLS(ob, oc).calculate_particle_update(u.clone(), 1e-3, noise=xi)
        e_want = O.PLS(ob, oc).calculate_energy_potential(u.clone())
        if not torch.isfinite(want).all():
            continue
        with rank2(P, 256):
            e_in = torch.empty(j, device="cuda")
            got = gb.fused_step(gc, cu(u), 1e-3, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True, input_energy=e_in)
        with rank2(P, 0):
            old = gb.fused_step(gc, cu(u), 1e-3, noise=P.basis.NoiseSpec(injected=cu(xi)), force_generic=True)
        assert relerr(got, want) < TOL, name
        assert relerr(got, old) < TOL, name
        assert abs(e_in.mean().item() - e_want) < TOL * abs(e_want), name
        checked += 1
    assert checked >= 7, checked
