"""GPU: solves with the inverse Cholesky factor and the inducing-point Gaussian step in whitened coordinates.

Reference: V = gpytorch.solve(k(Z,Z), ., U) at basis/inducing_point.py:89-93, :130-132 and the update of :117-150 under
costs/gaussian.py:86-88.  Round 3 replaces the block substitution (serial over the block rows: 32 workgroups for a
1024-column shard) by triangular products with Lc^-1, and the step by  S = Lc^-1 U -> dS = -eta (Q S - c~) + sqrt(2 eta) xi
-> dU = Lc dS  ("whitened coordinates").  These tests hold both to the CPU oracle, to LAPACK and to
the round-2 route they replace.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pls_oracle as O
from test_gpu_ksplit import P, _f64_default, ksplit  # noqa: F401
from test_gpu_parity import FUZZ_SEED, TOL, build_ipb, cu, make_problem, relerr

TOL0 = TOL


class solve_mode:
    def __init__(self, P, mode):
        self.L, self.lib, self.mode = P.pkg._lib, P.pkg._lib.load(), mode

    def __enter__(self):
        self.prev = self.lib.pls_get_option(self.L.OPT_SOLVE_MODE)
        self.L.check(self.lib.pls_set_option(self.L.OPT_SOLVE_MODE, self.mode))

    def __exit__(self, *exc):
        self.L.check(self.lib.pls_set_option(self.L.OPT_SOLVE_MODE, self.prev))
        return False


def _spd(m, d, ls, jitter, g):
    z = torch.rand(m, d, generator=g) * 2 - 1
    k = O.RBFARDKernel(torch.full((d,), ls), 1.0)(z, z) + jitter * torch.eye(m)
    return k


@pytest.mark.parametrize("m,j,ls,jitter", [(100, 64, 0.4, 0.0), (257, 130, 0.6, 1e-9), (1024, 256, 0.5, 1e-8), (300, 40, 1.2, 1e-7)])
def test_inverse_factor_solves_match_substitution_and_lapack(P, m, j, ls, jitter):
    """With the SAME factor (LAPACK's, uploaded): triangular products with Lc^-1 == block substitution == LAPACK's
    cholesky_solve, to 1e-11 and better, up to cond(K) ~ 1e9; forward half and narrow right-hand sides included."""
    from projected_langevin_sampling_amd import _chol

    g = torch.Generator().manual_seed(3 + FUZZ_SEED)
    k = _spd(m, 3, ls, jitter, g)
    lc = torch.linalg.cholesky(k)
    f = _chol.factor_from_host(lc).build_inverse()
    assert relerr(f.Linv, torch.linalg.solve_triangular(lc, torch.eye(m), upper=False)) < 1e-10 * max(1.0, torch.linalg.cond(lc).item() / 1e4)
    assert torch.equal(f.Linv.T.contiguous(), f.LinvT.contiguous()), "LinvT is the transpose of Linv"
    assert f.Linv.triu(1).abs().max().item() == 0.0, "the inverse factor is exactly lower triangular"
    for jj in (j, 1, 33):
        u = torch.randn(m, jj, generator=g)
        want = torch.cholesky_solve(u, lc)
        with solve_mode(P, 1):
            prod = f.solve(cu(u))
            fwd = f.forward_solve(cu(u))
        with solve_mode(P, 0):
            sub = f.solve(cu(u))
            fwd0 = f.forward_solve(cu(u))
        assert relerr(prod, sub) < 1e-11, (m, jj)
        assert relerr(fwd, fwd0) < 1e-11 and relerr(fwd, torch.linalg.solve_triangular(lc, u, upper=False)) < 1e-11
        assert relerr(prod, want) < 1e-11 * max(1.0, torch.linalg.cond(k).item() / 1e6), (m, jj)


def _gauss_pair(P, pr, sigma2=0.3):
    return (O.GaussianCost(sigma2, pr["y"], O.IdentityLink()),
            P.costs.GaussianCost(sigma2, pr["y"], P.links.IdentityLinkFunction()))


@pytest.mark.parametrize("n,m,j,d,ls_scale", [(512, 24, 64, 3, 0.35), (700, 33, 130, 2, 0.35), (900, 150, 70, 3, 0.35),
                                              (600, 60, 50, 2, 0.95), (1500, 300, 200, 3, 0.5)])
@pytest.mark.parametrize("factor", ["device", "shared"])
def test_whitened_route_of_the_inducing_point_step_against_the_oracle(P, n, m, j, d, ls_scale, factor):
    """pls_ipb_step's Gaussian path (forward solve, fused Q S kernel, Lc dS) against the oracle's update with the SAME
    coloured noise e injected; delta and in-place forms, the energy by-product, and the round-2 route next to it."""
    pr = make_problem(n, m, j, d, seed=7 * n + m + FUZZ_SEED)
    pr["ls"] = pr["ls"] * ls_scale
    ob, gb = build_ipb(P, pr, factor=factor)
    # TOL with no conditioning allowance holds up to cond(k(Z,Z)) ~ 1e8, as for the block substitution (two valid
    # Cholesky factors of one matrix already move the update by cond * 1e-17: the oracle's LAPACK factor is one of them)
    cond = torch.linalg.cond(ob.base_gram_induce).item()
    if FUZZ_SEED != 0 and cond > 1e9:
        pytest.skip(f"soak draw with cond(k(Z,Z)) = {cond:.1e}: beyond what this case is built to hold")
    assert cond <= 1e10, f"test construction: cond(k(Z,Z)) = {cond:.1e}"
    # (the committed draws hold TOL as it is; a soak run with PLS_FUZZ_SEED may draw a worse-conditioned k(Z,Z), and two valid
    # Cholesky factors of one matrix -- the oracle's LAPACK factor is one, the device factor another -- move the update by
    # ~2.5 cond 1e-17)
    TOL = TOL0 if FUZZ_SEED == 0 else TOL0 * max(1.0, cond / 4e7)
    oc, gc = _gauss_pair(P, pr)
    u = pr["u"]
    e_noise = torch.randn(m, j, generator=pr["gen"])
    eta = 1e-3
    want = O.PLS(ob, oc).calculate_particle_update(u.clone(), eta, noise=e_noise)
    e_want = O.PLS(ob, oc).calculate_energy_potential(u.clone())
    assert gb.whitened
    e_in = torch.empty(j, device="cuda")
    got = gb.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(e_noise)), input_energy=e_in)
    assert gb._Q is not None and gb._q_inv_noise == 1.0 / 0.3
    new = gb.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(e_noise)), new_state=True)
    assert relerr(got, want) < TOL and relerr(new, u + want) < TOL
    # (the energy holds (M/2) |k(Z,Z)^-1 U|^2: twice the sensitivity of the update to the factor)
    assert abs(e_in.mean().item() - e_want) < (TOL if FUZZ_SEED == 0 else 4.0 * TOL) * abs(e_want)
    assert relerr(gb.fused_particle_energy(gc, cu(u)), e_in) < 1e-10
    gb.whitened = False  # the round-2 route: solve, B V, update
    try:
        old = gb.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(injected=cu(e_noise)))
        e_old = gb.fused_particle_energy(gc, cu(u))
    finally:
        gb.whitened = True
    assert relerr(old, want) < TOL and relerr(got, old) < TOL and relerr(e_in, e_old) < TOL


def test_whitened_loop_is_the_same_chain_as_the_step_by_step_loop(P):
    """T steps that keep S = Lc^-1 U between them (one contraction per step) == T calls of the U -> U step with the same
    Philox counters: same particles (after U = Lc S), same per-step energies; and a different observation noise rebuilds
    the whitened operator."""
    pr = make_problem(800, 96, 160, 3, seed=23 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.4
    ob, gb = build_ipb(P, pr)
    _, gc = _gauss_pair(P, pr)
    u = cu(pr["u"])
    eta, steps = 2e-3, 25
    e_u, e_s = [], []
    cur = u.clone()
    for t in range(steps):
        e = torch.empty(cur.shape[1], device="cuda")
        cur = gb.fused_step(gc, cur, eta, new_state=True, noise=P.basis.NoiseSpec(seed=4, step=t), input_energy=e)
        e_u.append(e.mean().item())
    s = gb.whiten(u)
    assert relerr(gb.unwhiten(s), u) < 1e-12
    for t in range(steps):
        e = torch.empty(s.shape[1], device="cuda")
        s = gb.whitened_step(gc, s, eta, new_state=True, noise=P.basis.NoiseSpec(seed=4, step=t), input_energy=e)
        e_s.append(e.mean().item())
    assert relerr(gb.unwhiten(s), cur) < TOL
    assert max(abs(a - b) / abs(a) for a, b in zip(e_u, e_s)) < TOL
    assert relerr(gb.whitened_particle_energy(gc, s), gb.fused_particle_energy(gc, gb.unwhiten(s))) < 1e-9
    # another observation noise: the operator is rebuilt, the step follows the oracle again
    oc2, gc2 = _gauss_pair(P, pr, sigma2=0.05)
    e_noise = torch.randn(96, 160, generator=pr["gen"])
    want = O.PLS(ob, oc2).calculate_particle_update(pr["u"].clone(), eta, noise=e_noise)
    got = gb.fused_step(gc2, u, eta, noise=P.basis.NoiseSpec(injected=cu(e_noise)))
    assert gb._q_inv_noise == 1.0 / 0.05 and relerr(got, want) < TOL


def test_whitened_step_size_blocks_and_narrow_shards(P):
    """per-block step sizes (the batched step-size search) and a J-shard with a column offset through the whitened route"""
    pr = make_problem(600, 80, 192, 2, seed=31 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.4
    ob, gb = build_ipb(P, pr)
    _, gc = _gauss_pair(P, pr)
    u = cu(pr["u"])
    etas = [1e-3, 0.0, 4e-3]
    blocks = P.basis.BlockSpec(64, cu(torch.tensor(etas)))
    got = gb.fused_step(gc, u, 0.0, blocks=blocks, noise=P.basis.NoiseSpec(seed=8, step=2))
    for b, eta in enumerate(etas):
        alone = gb.fused_step(gc, u[:, 64 * b:64 * (b + 1)].contiguous(), eta, noise=P.basis.NoiseSpec(seed=8, step=2))
        assert relerr(got[:, 64 * b:64 * (b + 1)], alone) < 1e-12 if eta else got[:, 64 * b:64 * (b + 1)].abs().max().item() == 0.0
    full = gb.fused_step(gc, u, 1e-3, noise=P.basis.NoiseSpec(seed=8, step=2))
    shard = gb.fused_step(gc, u[:, 100:].contiguous(), 1e-3, noise=P.basis.NoiseSpec(seed=8, step=2, j_offset=100))
    assert relerr(shard, full[:, 100:]) < 1e-12


def test_whitened_philox_noise_is_coloured_by_kzz(P):
    pr = make_problem(200, 6, 20000, 1, seed=5)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    gc = P.costs.GaussianCost(0.5, pr["y"], P.links.IdentityLinkFunction())
    u = torch.zeros(6, 20000, dtype=torch.float64, device="cuda")
    eta = 0.5
    base = gb.fused_step(gc, u, eta, noise=P.basis.NoiseSpec(none=True))
    got = gb.fused_step(gc, u, eta, noise=P.basis.NoiseSpec(seed=3, step=0))
    e = (got - base) / math.sqrt(2 * eta)  # = Lc xi
    assert relerr((e @ e.T / e.shape[1]).cpu(), ob.base_gram_induce) < 5e-2
    gb.whitened = False
    try:
        old = gb.fused_step(gc, u, eta, noise=P.basis.NoiseSpec(seed=3, step=0))
    finally:
        gb.whitened = True
    assert relerr(got, old) < 1e-10, "both routes draw the same xi"


def test_train_pls_keeps_the_inducing_point_particles_whitened_between_steps(P):
    """experiments/trainers.py:139-162 on the inducing-point basis with the Gaussian cost: the loop whitens once, steps S
    (one contraction per iteration) and maps back once -- particles, every energy, the stop index and torch's RNG state
    are those of the plain loop that calls the U -> U step and the energy every iteration."""
    import numpy as np

    from projected_langevin_sampling_amd import trainers

    pr = make_problem(700, 64, 96, 2, seed=41 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.4
    ob, gb = build_ipb(P, pr)
    _, gc = _gauss_pair(P, pr)
    u0 = cu(pr["u"])
    eta, steps = 2e-3, 40
    for patience in (1e9, 6 * eta):
        torch.manual_seed(3)
        pls = P.pkg.PLS(gb, gc)
        assert trainers._LoopSpace(pls, None).whitened
        ua, ea = P.pkg.train_pls(pls, u0.clone(), steps, eta, patience)
        state_a = torch.get_rng_state()
        torch.manual_seed(3)
        ub, eb, es = u0.clone(), [], trainers.EarlyStopper(patience=patience)
        for _ in range(steps):
            pls.step_(ub, eta)
            e = pls.calculate_energy_potential(ub)
            if es.should_stop(e, eta):
                break
            eb.append(e)
        assert len(ea) == len(eb) and np.allclose(ea, eb, rtol=TOL, atol=0), (len(ea), len(eb))
        assert relerr(ua, ub) < TOL
        assert torch.equal(state_a, torch.get_rng_state())
    # injected (coloured) noise keeps the loop in the original coordinates
    noises = [cu(torch.randn(64, 96, generator=pr["gen"])) for _ in range(5)]
    assert not trainers._LoopSpace(P.pkg.PLS(gb, gc), noises).whitened
    oc, _ = _gauss_pair(P, pr)
    uw, ew = O.train_pls(O.PLS(ob, oc), pr["u"].clone(), 5, eta, 1e9, noises=[n.cpu() for n in noises])
    ug, eg = P.pkg.train_pls(P.pkg.PLS(gb, gc), u0.clone(), 5, eta, 1e9, noises=noises)
    assert relerr(ug, uw) < 1e-8 and np.allclose(eg, ew, rtol=TOL)
