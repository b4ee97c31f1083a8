"""The one-launch small-rank step (csrc/small_rank_step.h; pls_block_desc.step_sync / energy_sums16,
PLS_OPT_SMALL_RANK_STEP) on the GPU: against the CPU oracle on the same seeded inputs (TOL as in test_gpu_parity.py), against
the slab kernels + update launch it replaces in the launch-bound regime, and its own contracts -- several row slabs per
column block meeting through the arrival counters, counters left zero, the same bits from launch to launch, the energy
sums it delivers, the memset fall-back for callers without counters, column blocks with their own step sizes.
Reference path: projected_langevin_sampling.py:107-138, basis/orthonormal.py:98-159, experiments/trainers.py:149-158."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pls_oracle as O

from test_gpu_parity import FUZZ_SEED, TOL, build_ipb, build_onb, cu, make_costs, make_problem, relerr, step_tolerance  # noqa: E402
from test_gpu_parity import P, _f64_default  # noqa: F401,E402  (fixtures)


@pytest.fixture
def route(P):
    """set(mode): PLS_OPT_SMALL_RANK_STEP for the duration of a test (0 slab kernels, 1 default, 2 one launch wherever it applies)"""
    L = P.pkg._lib
    lib = L.load()
    prev = lib.pls_get_option(L.OPT_SMALL_RANK_STEP)

    def set_mode(mode):
        L.check(lib.pls_set_option(L.OPT_SMALL_RANK_STEP, mode), "pls_set_option")

    yield set_mode
    L.check(lib.pls_set_option(L.OPT_SMALL_RANK_STEP, prev), "pls_set_option")


def _timeline_names(P, fn):
    with P.pkg._lib.Timeline(64) as tl:
        fn()
    return sorted(tl.summary())


# (n, m, j, d): one slab / ragged J and odd rank / several slabs of a narrow rank / several slabs of a wide rank with a row tail /
# a rank of exactly 128 / fewer rows than one round of tiles
SHAPES = [(100, 10, 64, 1), (333, 17, 37, 2), (3000, 30, 40, 2), (1530, 120, 200, 4), (900, 140, 48, 3), (40, 12, 5, 1)]


@pytest.mark.parametrize("n,m,j,d", SHAPES)
def test_one_launch_step_against_the_oracle_and_the_slab_kernels(P, route, n, m, j, d):
    pr = make_problem(n, m, j, d, seed=3 * n + m + FUZZ_SEED)
    ob, gb = build_onb(P, pr, threshold=1e-7)
    mk = ob.approximation_dimension
    if mk > 128:
        pytest.skip(f"{mk} functions kept: not a small-rank basis")
    u = pr["u"][:mk].contiguous()
    xi = torch.randn(mk, j, generator=pr["gen"])
    eta = 1e-3
    checked = 0
    for name, oc, gc in make_costs(P, pr["y"], pr["fstar"], pr["gen"]):
        if name == "gaussian/identity":
            continue  # (the B = A A^T fast path: not this kernel's business; gaussian/square below is the generic Gaussian)
        want = O.PLS(ob, oc).calculate_particle_update(u.clone(), eta, noise=xi)
        tol = step_tolerance(ob, oc, u, eta, xi, want)
        if tol >= 1e-8:
            continue
        checked += 1
        spec = P.basis.NoiseSpec(injected=cu(xi))
        e_new = torch.full((j,), float("nan"), device="cuda")
        route(2)
        names = _timeline_names(P, lambda: gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_new))
        assert names == ["small_rank_step"], f"{name}: launches {names}"
        got = gb.fused_step(gc, cu(u), eta, noise=spec)
        got_e = gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_new)
        assert torch.equal(got, got_e), "the energy by-product must not move the step"
        assert relerr(got, want) < tol, name
        e_want = oc.calculate_cost(ob.calculate_untransformed_train_prediction_samples(u)) + 0.5 * ((u * u) / ob.eigenvalues[:, None]).sum(dim=0)
        assert relerr(e_new, e_want) < 1e-9, name
        route(0)
        e_old = torch.empty(j, device="cuda")
        old = gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_old)
        assert relerr(got, old) < 1e-11 and relerr(e_new, e_old) < 1e-11, name
        route(2)
        # new state instead of the update; Philox noise: the stream of the other routes, bit for bit (same counters)
        ph = P.basis.NoiseSpec(seed=77, step=5, j_offset=3)
        a = gb.fused_step(gc, cu(u), eta, noise=ph, new_state=True)
        route(0)
        b = gb.fused_step(gc, cu(u), eta, noise=ph, new_state=True)
        assert relerr(a, b) < 1e-11, name
        nz_new = a - cu(u) - gb.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(none=True))
        route(2)
        nz_old = a - cu(u) - gb.fused_step(gc, cu(u), eta, noise=P.basis.NoiseSpec(none=True))
        assert relerr(nz_new, nz_old) < 1e-9
    assert checked >= 5, f"only {checked} (cost, link) pairs were checked"


def test_slabs_meet_through_counters_that_are_left_zero_and_the_bits_do_not_move(P, route):
    """Several row slabs per column block: the last arriver adds them in ascending order whoever it is -- the same bits from
    launch to launch --, the caller's counters are zero again after every launch, and a basis keeps one counter set per stream."""
    pr = make_problem(6000, 40, 96, 3, seed=41 + FUZZ_SEED)
    _, gb = build_onb(P, pr)
    mk = gb.approximation_dimension
    _, _, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]  # bernoulli/sigmoid
    u = cu(pr["u"][:mk].contiguous())
    route(2)
    lib = P.pkg._lib.load()
    words = int(lib.pls_step_sync_words(96))
    assert words == 6 + 1
    sync = torch.zeros(words, dtype=torch.int32, device="cuda")
    eta = torch.full((1,), 1e-3, device="cuda")
    spec = P.basis.NoiseSpec(seed=9, step=1)
    outs = []
    for rep in range(6):
        e = torch.empty(96, device="cuda")
        s16 = torch.full((6,), float("nan"), device="cuda")
        s256 = torch.full((1,), float("nan"), device="cuda")
        blocks = P.basis.BlockSpec(96, eta, step_sync=sync, energy_sums16=s16.data_ptr(), energy_sums=s256.data_ptr())
        out = gb.fused_step(gc, u, 0.0, noise=spec, input_energy=e, blocks=blocks)
        assert int(sync.abs().sum()) == 0, "the launch must leave its counters zero"
        outs.append((out, e, s16, s256))
    for out, e, s16, s256 in outs[1:]:
        assert torch.equal(out, outs[0][0]) and torch.equal(e, outs[0][1]) and torch.equal(s16, outs[0][2]) and torch.equal(s256, outs[0][3])
    out, e, s16, s256 = outs[0]
    # the sums it delivers: 16 columns each in ascending order; the 256-column chunk in the library's fixed order
    want16 = torch.stack([e[16 * b:16 * b + 16].cpu().cumsum(0)[-1] for b in range(6)])
    assert torch.equal(s16.cpu(), want16)
    chunk = torch.empty(1, device="cuda")
    P.pkg._lib.check(lib.pls_chunk_sums(e.data_ptr(), 96, chunk.data_ptr(), P.pkg._lib.stream_ptr()), "pls_chunk_sums")
    assert torch.equal(s256, chunk)
    alone = torch.empty(6, device="cuda")
    P.pkg._lib.check(lib.pls_sums16(e.data_ptr(), 96, alone.data_ptr(), P.pkg._lib.stream_ptr()), "pls_sums16")
    assert torch.equal(alone, s16)
    # the basis' own counters: one set per stream, zero after use, dropped after a failed launch
    gb.zero_step_sync()
    a = gb.fused_step(gc, u, 1e-3, noise=spec)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b = gb.fused_step(gc, u, 1e-3, noise=spec)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, out)
    pool = gb._sync_pool
    assert len(pool) == 2 and all(int(t.abs().sum()) == 0 for t in pool.values())
    # stale counters (a launch that died half way) make later steps wrong: the FIRST workgroup to arrive at column block 0
    # takes itself for the last and finishes with slabs nobody has written yet.  A failed launch makes the basis drop its
    # counters (fused_step's error path); zero_step_sync is the same repair by hand
    key = next(k for k in pool if k[1] == P.pkg._lib.stream_ptr())
    assert int(pool[key].abs().sum()) == 0
    gb.zero_step_sync()
    assert not hasattr(gb, "_sync_pool")
    good = gb.fused_step(gc, u, 1e-3, noise=spec)
    assert torch.equal(good, out)
    with pytest.raises(P.pkg._lib.PlsHipError):  # (a launch the library refuses: out aliases the particles ...)
        P.pkg._lib.check(1, "pls_onb_step_blocks")
    calls = []
    real = gb.zero_step_sync
    gb.zero_step_sync = lambda: (calls.append(1), real())[1]
    try:
        with pytest.raises(P.pkg._lib.PlsHipError):  # ... and a failing step call drops the counters before it re-raises
            gb.fused_step(gc, u, 1e-3, noise=P.basis.NoiseSpec(injected=torch.zeros(mk, 8, device="cuda")))  # (8 < 96 columns)
    finally:
        del gb.zero_step_sync
    assert calls == [1]


def test_callers_without_counters_get_a_memset_in_front_of_the_launch(P, route):
    """pls_onb_step straight through the C ABI, no pls_block_desc: the counters come out of the workspace and a memset node
    zeroes them -- whatever the workspace held."""
    pr = make_problem(5000, 24, 70, 2, seed=8 + FUZZ_SEED)
    _, gb = build_onb(P, pr)
    mk = gb.approximation_dimension
    _, _, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[1]  # poisson/square
    u = cu(1.0 + 0.1 * pr["u"][:mk].contiguous())
    L = P.pkg._lib
    lib = L.load()
    route(2)
    desc, cd, y = gb._desc(), gc.desc(), gc.y_device()
    nbytes = int(lib.pls_onb_step_workspace_bytes(desc, 70, 5000))
    ws = torch.full(((nbytes + 7) // 8,), float("nan"), device="cuda")  # (NaN bit patterns where the counters will be)
    nd = P.basis.NoiseSpec(seed=4, step=0).desc()
    out = torch.empty_like(u)
    e = torch.empty(70, device="cuda")
    for rep in range(2):
        names = _timeline_names(P, lambda: L.check(lib.pls_onb_step(desc, cd, y.data_ptr(), u.data_ptr(), L.ld(u), 70, 1e-3, nd, out.data_ptr(),
                                                                      L.ld(out), L.OUT_NEW_STATE, 0, e.data_ptr(), ws.data_ptr(), nbytes,
                                                                      L.stream_ptr()), "pls_onb_step"))
        assert names == ["small_rank_step"]
    want_e = torch.empty(70, device="cuda")
    want = gb.fused_step(gc, u, 1e-3, noise=P.basis.NoiseSpec(seed=4, step=0), new_state=True, input_energy=want_e)
    assert torch.equal(out, want) and torch.equal(e, want_e)
    # a workspace too small for the slabs: the call falls back to the general route's own message
    small = torch.empty(16, device="cuda")
    rc = lib.pls_onb_step(desc, cd, y.data_ptr(), u.data_ptr(), L.ld(u), 70, 1e-3, nd, out.data_ptr(), L.ld(out), L.OUT_NEW_STATE, 0,
                          None, small.data_ptr(), 128, L.stream_ptr())
    assert rc != 0 and b"workspace" in lib.pls_last_error()


def test_column_blocks_with_their_own_step_sizes(P, route):
    """The S candidates of a step-size search as S column blocks of one launch (experiments/runners.py:331-446): every block
    equals a stand-alone run of block_cols particles with its step size and its own noise stream; a frozen block (eta = 0)
    does not move."""
    pr = make_problem(800, 16, 90, 2, seed=13 + FUZZ_SEED)
    _, gb = build_onb(P, pr)
    mk = gb.approximation_dimension
    _, _, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
    u = cu(pr["u"][:mk].contiguous())
    route(2)
    etas = torch.tensor([1e-3, 0.0, 4e-3], device="cuda")
    spec = P.basis.NoiseSpec(seed=21, step=3)
    e = torch.empty(90, device="cuda")
    blocks = P.basis.BlockSpec(30, etas)
    got = gb.fused_step(gc, u, 0.0, noise=spec, new_state=True, input_energy=e, blocks=blocks)
    for b, eta in enumerate([1e-3, 0.0, 4e-3]):
        ub = u[:, 30 * b:30 * b + 30].contiguous()
        eb = torch.empty(30, device="cuda")
        alone = gb.fused_step(gc, ub, eta, noise=spec, new_state=True, input_energy=eb)
        assert torch.equal(got[:, 30 * b:30 * b + 30], alone), b
        assert torch.equal(e[30 * b:30 * b + 30], eb), b
    assert torch.equal(got[:, 30:60], u[:, 30:60])


def test_training_loop_is_one_launch_per_iteration_and_equals_the_plain_loop(P, route):
    """train_pls for a cost without the Gaussian algebra on a small basis: every iteration of the pipelined loop is ONE launch
    (step, energies of its input, their 16-column sums into the pinned slot the host polls); same particles, energies, stop
    index and torch generator state as the plain loop (step, then a separate energy pass), early stop included."""
    pr = make_problem(600, 20, 48, 2, seed=5 + FUZZ_SEED)
    ob, gb = build_onb(P, pr)
    mk = ob.approximation_dimension
    name, oc, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
    pls = P.pkg.PLS(gb, gc)
    u0 = cu(pr["u"][:mk].contiguous())
    route(1)
    with P.pkg._lib.Timeline(256) as tl:
        torch.manual_seed(3)
        P.pkg.train_pls(pls, u0.clone(), 30, 2e-6, 1e9)
    summary = tl.summary()
    assert set(summary) <= {"small_rank_step", "small_rank_value", "other"}, summary
    assert summary["small_rank_step"]["launches"] >= 30  # (speculative launches past the last iteration are allowed)
    # a run that goes on to the end, and one that must stop: a step size beyond the stability bound of the stiffest mode
    # (eta / lambda_min = 2.5 > 2: its energy grows 2.25-fold per step and turns the mean energy round within a few steps),
    # particles drawn from the prior, patience = 2.5 such steps
    eta0 = 0.5 * float(gb.eigenvalues.min())
    ueq = u0 * gb.eigenvalues.sqrt()[:, None]
    for start, step, patience in ((u0, 2e-6, 1e9), (ueq, 5.0 * eta0, 12.5 * eta0)):
        runs = {}
        for mode in ("pipelined", "plain"):
            if mode == "plain":
                gb.supports_input_energy = lambda c: False
            try:
                torch.manual_seed(44)
                out, energies = P.pkg.train_pls(pls, start.clone(), 40, step, patience)
                runs[mode] = (out, energies, torch.get_rng_state())
            finally:
                if mode == "plain":
                    del gb.supports_input_energy
        assert len(runs["pipelined"][1]) == len(runs["plain"][1])
        assert torch.equal(runs["pipelined"][0], runs["plain"][0])
        assert np.allclose(runs["pipelined"][1], runs["plain"][1], rtol=1e-11)
        assert torch.equal(runs["pipelined"][2], runs["plain"][2])
    assert 2 <= len(runs["plain"][1]) < 40, "the unstable run must stop somewhere in the middle"


def test_the_drop_in_loop_takes_a_pre_bound_step_and_nothing_changes(P, route):
    """`particles += pls.calculate_particle_update(particles, step_size)` (README.md:257-262, experiments/profiler/main.py:77-82)
    goes through basis.eager_step -- descriptors, workspace and counters bound once per (cost, J, step size, stream) --: the same
    updates as fused_step with the same draws from torch's generator, bit for bit, for the Gaussian fast path and for the
    one-launch step; a new step size, observation noise or target vector binds afresh."""
    pr = make_problem(400, 14, 52, 2, seed=23 + FUZZ_SEED)
    _, gb = build_onb(P, pr)
    mk = gb.approximation_dimension
    u0 = cu(pr["u"][:mk].contiguous())
    costs = make_costs(P, pr["y"], pr["fstar"], pr["gen"])
    for name, _, gc in (costs[0], costs[2]):  # gaussian/identity, bernoulli/sigmoid
        pls = P.pkg.PLS(gb, gc)
        torch.manual_seed(7)
        a = u0.clone()
        for step in (1e-4, 1e-4, 3e-4, 3e-4):
            a += pls.calculate_particle_update(a, step)
        state_a = torch.get_rng_state()
        torch.manual_seed(7)
        b = u0.clone()
        for step in (1e-4, 1e-4, 3e-4, 3e-4):
            b += gb.fused_step(gc, b, step)
        assert torch.equal(a, b) and torch.equal(state_a, torch.get_rng_state()), name
        assert gb._eager[0][2] == 3e-4
    # rebinding: observation noise (a field of the cost descriptor) and a new target vector
    gauss = costs[0][2]
    pls = P.pkg.PLS(gb, gauss)
    torch.manual_seed(8)
    first = pls.calculate_particle_update(u0, 1e-4)
    pls.observation_noise = 2.0 * gauss.observation_noise
    torch.manual_seed(8)
    second = pls.calculate_particle_update(u0, 1e-4)
    torch.manual_seed(8)
    assert torch.equal(second, gb.fused_step(gauss, u0, 1e-4)) and not torch.equal(first, second)
    gauss.y_train = gauss.y_train + 1.0
    torch.manual_seed(8)
    third = pls.calculate_particle_update(u0, 1e-4)
    torch.manual_seed(8)
    assert torch.equal(third, gb.fused_step(gauss, u0, 1e-4)) and not torch.equal(third, second)
    # a strided view of a wider tensor and float32 particles still work (the latter through fused_step's promotion)
    wide = cu(torch.randn(mk, 80, generator=pr["gen"]))
    torch.manual_seed(9)
    v = pls.calculate_particle_update(wide[:, 3:55], 1e-4)
    torch.manual_seed(9)
    assert torch.equal(v, gb.fused_step(gauss, wide[:, 3:55].contiguous(), 1e-4))
    torch.manual_seed(9)
    assert pls.calculate_particle_update(u0.float(), 1e-4).dtype == torch.float64


@pytest.mark.parametrize("n,mk,j", [(1, 1, 1), (17, 1, 16), (64, 128, 3), (65, 127, 33), (700, 5, 600), (2600, 64, 530)])
def test_edges_of_the_one_launch_step(P, route, n, mk, j):
    """One data row, one function, one particle; a rank of exactly 128 and an odd one below it; more columns than one
    256-column chunk with several slabs (the chunk sums' second hand-over across column blocks); a targets vector whose
    address is not 16-byte aligned (the route steps aside).  Against the slab kernels + update launch at 1e-12 and against a
    host evaluation of the Gaussian-generic step."""
    g = torch.Generator().manual_seed(1000 * n + mk + j + FUZZ_SEED)
    a = torch.randn(mk, n, generator=g) / math.sqrt(max(n, 1))
    lam = torch.rand(mk, generator=g) + 0.5
    basis = P.basis.OrthonormalBasis.from_projection(cu(a), cu(lam))
    y = torch.randn(n, generator=g)
    u = torch.randn(mk, j, generator=g)
    xi = torch.randn(mk, j, generator=g)
    eta = 1e-3
    spec = P.basis.NoiseSpec(injected=cu(xi))
    lib = P.pkg._lib.load()
    for cost in (P.costs.GaussianCost(0.3, y, P.links.IdentityLinkFunction()), P.costs.BernoulliCost((y > 0).double(), P.links.SigmoidLinkFunction())):
        gauss = isinstance(cost, P.costs.GaussianCost)
        res = {}
        for mode in (2, 0):
            route(mode)
            e = torch.full((j,), float("nan"), device="cuda")
            nchunk = (j + 255) // 256
            s256 = torch.full((nchunk,), float("nan"), device="cuda")
            s16 = torch.full(((j + 15) // 16,), float("nan"), device="cuda")
            sync = torch.zeros(int(lib.pls_step_sync_words(j)), dtype=torch.int32, device="cuda")
            blocks = P.basis.BlockSpec(j, torch.full((1,), eta, device="cuda"), energy_sums=s256.data_ptr(), energy_sums16=s16.data_ptr(),
                                       step_sync=sync)
            names = _timeline_names(P, lambda: basis.fused_step(cost, cu(u), 0.0, noise=spec, force_generic=gauss, input_energy=e, blocks=blocks))
            assert ("small_rank_step" in names) == (mode == 2), names
            out = basis.fused_step(cost, cu(u), 0.0, noise=spec, force_generic=gauss, input_energy=e, blocks=blocks)
            assert int(sync.abs().sum()) == 0
            res[mode] = (out, e.clone(), s256.clone(), s16.clone())
        assert relerr(res[2][0], res[0][0]) < 1e-12 and relerr(res[2][1], res[0][1]) < 1e-12
        # the sums: of THIS route's energies, in the library's orders
        chunk = torch.empty_like(res[2][2])
        P.pkg._lib.check(lib.pls_chunk_sums(res[2][1].data_ptr(), j, chunk.data_ptr(), P.pkg._lib.stream_ptr()), "pls_chunk_sums")
        b16 = torch.empty_like(res[2][3])
        P.pkg._lib.check(lib.pls_sums16(res[2][1].data_ptr(), j, b16.data_ptr(), P.pkg._lib.stream_ptr()), "pls_sums16")
        assert torch.equal(res[2][2], chunk) and torch.equal(res[2][3], b16)
        assert torch.equal(res[0][2], torch.stack([res[0][1][256 * c:256 * c + 256].sum() for c in range(nchunk)])) or relerr(res[0][2], chunk) < 1e-12
        if gauss:
            f = a.T @ u
            want = -eta * (a @ ((f - y[:, None]) / 0.3)) - eta * u / lam[:, None] + math.sqrt(2 * eta) * xi
            assert relerr(res[2][0], want) < 1e-12
            e_want = ((f - y[:, None]) ** 2).sum(dim=0) / (2 * 0.3) + 0.5 * (u * u / lam[:, None]).sum(dim=0)
            assert relerr(res[2][1], e_want) < 1e-12
    # targets at an address that is not a multiple of 16: the one-launch route steps aside, the result does not change
    cost = P.costs.BernoulliCost((y > 0).double(), P.links.SigmoidLinkFunction())
    route(2)
    aligned = basis.fused_step(cost, cu(u), eta, noise=spec)
    base = torch.empty(n + 1, device="cuda")
    base[1:] = cost.y_device()
    cost._y_dev = base[1:]
    assert cost.y_device().data_ptr() % 16 == 8
    names = _timeline_names(P, lambda: basis.fused_step(cost, cu(u), eta, noise=spec))
    assert "small_rank_step" not in names
    assert relerr(basis.fused_step(cost, cu(u), eta, noise=spec), aligned) < 1e-12


# ------------------------------------------------------------------------------------------------------------
# the inducing-point basis (basis/inducing_point.py:60-135): V = k(Z,Z)^-1 U and the coloured noise by their own launches, then
# projection k(X,Z) V, cost, back-projection, prior drift M V, update and energies in the same one launch
IPB_SHAPES = [(100, 10, 64, 1), (333, 17, 37, 2), (3000, 30, 40, 2), (1100, 128, 90, 3), (640, 100, 130, 2), (500, 65, 16, 2)]


def _ipb_prep(P, mode):
    L = P.pkg._lib
    L.check(L.load().pls_set_option(L.OPT_IPB_PREP, mode), "pls_set_option")


@pytest.mark.parametrize("n,m,j,d", IPB_SHAPES)
def test_inducing_point_basis_takes_the_one_launch_step(P, route, n, m, j, d):
    pr = make_problem(n, m, j, d, seed=11 * n + m + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    u = pr["u"]
    e_noise = torch.randn(m, j, generator=pr["gen"])
    eta = 1e-3
    checked = 0
    for name, oc, gc in make_costs(P, pr["y"], pr["fstar"], pr["gen"]):
        if name == "gaussian/identity":
            continue
        want = O.PLS(ob, oc).calculate_particle_update(u.clone(), eta, noise=e_noise)
        tol = step_tolerance(ob, oc, u, eta, e_noise, want)
        if tol >= 1e-8:
            continue
        checked += 1
        spec = P.basis.NoiseSpec(injected=cu(e_noise))
        e_new = torch.full((j,), float("nan"), device="cuda")
        route(2)
        names = _timeline_names(P, lambda: gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_new))
        assert names == ["ipb_prep", "small_rank_step"], names  # (solve [+ coloured noise], then everything else of the step)
        got = gb.fused_step(gc, cu(u), eta, noise=spec)
        _ipb_prep(P, 0)
        try:  # the solve as two triangular products of their own: the same V to rounding
            names = _timeline_names(P, lambda: gb.fused_step(gc, cu(u), eta, noise=spec))
            assert names == ["gemm_store", "small_rank_step"], names
            # (a tenth of the oracle tolerance: for a cost with a pole the two solves' rounding is amplified like anything else)
            assert relerr(gb.fused_step(gc, cu(u), eta, noise=spec), got) < max(1e-11, 0.1 * tol), name
        finally:
            _ipb_prep(P, 1)
        got_e = gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_new)
        assert torch.equal(got, got_e), "the energy by-product must not move the step"
        assert relerr(got, want) < tol, name
        v = torch.cholesky_solve(u, torch.linalg.cholesky(ob.base_gram_induce))
        e_want = oc.calculate_cost(ob.calculate_untransformed_train_prediction_samples(u)) + 0.5 * m * (v * v).sum(dim=0)
        assert relerr(e_new, e_want) < 1e-9, name
        new_state = gb.fused_step(gc, cu(u), eta, noise=spec, new_state=True)
        assert relerr(new_state, u + want) < max(tol, 1e-12), name
        route(0)
        e_old = torch.empty(j, device="cuda")
        old = gb.fused_step(gc, cu(u), eta, noise=spec, input_energy=e_old)
        assert relerr(got, old) < max(1e-11, 0.1 * tol) and relerr(e_new, e_old) < 1e-11, name
        # Philox noise: the same draws, coloured by the same factor, whichever kernels finish the step
        pspec = P.basis.NoiseSpec(seed=77, step=3)
        old_p = gb.fused_step(gc, cu(u), eta, noise=pspec)
        route(2)
        new_p = gb.fused_step(gc, cu(u), eta, noise=pspec)
        assert relerr(new_p, old_p) < max(1e-11, 0.1 * tol), name
        assert _timeline_names(P, lambda: gb.fused_step(gc, cu(u), eta, noise=pspec)) == ["ipb_prep", "small_rank_step"]
        _ipb_prep(P, 0)
        try:
            assert relerr(gb.fused_step(gc, cu(u), eta, noise=pspec), new_p) < max(1e-11, 0.1 * tol), name
        finally:
            _ipb_prep(P, 1)
    assert checked >= 5, checked
    assert int(gb._step_sync(j, torch.device("cuda")).abs().sum()) == 0, "the arrival counters are left zero"


def test_inducing_point_training_loop_polls_the_sums_of_the_one_launch_step(P, route):
    """train_pls on the inducing-point basis, Bernoulli/sigmoid on a small basis: the pipelined loop (solve + noise + ONE step
    launch per iteration, 16-column energy sums into the pinned slot) and the plain loop give the same particles, energies, stop
    index and generator state."""
    pr = make_problem(500, 16, 48, 1, seed=21 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    name, oc, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
    pls = P.pkg.PLS(gb, gc)
    assert gb.supports_energy_sums(gc) and gb.uses_sums16(gc)
    assert not gb.supports_lagged_energies(gc), "lagged energies belong to the Gaussian/identity routes"
    u0 = cu(pr["u"])
    # a step size inside the stability bound of the stiffest mode of the prior drift M k(Z,Z)^-1 (eta M / lambda_min < 2)
    eta = 0.25 * float(torch.linalg.eigvalsh(ob.base_gram_induce).min()) / 16
    route(1)
    assert gb.whitened_generic_applies(gc, 48)  # (builds the whitened operand: one product, once per basis)
    with P.pkg._lib.Timeline(512) as tl:
        torch.manual_seed(3)
        P.pkg.train_pls(pls, u0.clone(), 20, eta, 1e9)
    summary = tl.summary()
    assert summary["small_rank_step"]["launches"] >= 20, summary
    # (whitened coordinates: S = Lc^-1 U once, ONE launch per iteration, U = Lc S once)
    assert not {"small_rank", "gemm_cost", "langevin_update", "block_means", "ipb_prep", "normal_fill"} & set(summary), summary
    assert summary.get("gemm_store", {"launches": 0})["launches"] <= 2, summary
    runs = {}
    for mode in ("pipelined", "plain"):
        if mode == "plain":
            gb.supports_input_energy = lambda c: False
        try:
            torch.manual_seed(44)
            out, energies = P.pkg.train_pls(pls, u0.clone(), 40, eta, 1e9)
            runs[mode] = (out, energies, torch.get_rng_state())
        finally:
            if mode == "plain":
                del gb.supports_input_energy
    assert len(runs["pipelined"][1]) == len(runs["plain"][1]) == 40 and np.isfinite(runs["plain"][1]).all()
    # ... and the chain in the ORIGINAL coordinates -- solve + coloured noise + step per call, the keys train_pls draws --: the
    # same draws (Lc^-1 Lc xi = xi), the same particles and energies to the conditioning of two routes through k(Z,Z)
    torch.manual_seed(44)
    keys = torch.randint(0, 2**62, (256,), dtype=torch.int64).tolist()
    u, es = u0.clone(), []
    for t in range(40):
        u = gb.fused_step(gc, u, eta, new_state=True, noise=P.basis.NoiseSpec(seed=keys[t], step=0))
        es.append(pls.particle_energy_potential(u).mean().item())
    assert relerr(runs["pipelined"][0], u) < 1e-9
    assert np.allclose(runs["pipelined"][1], es, rtol=1e-9)
    assert relerr(runs["pipelined"][0], runs["plain"][0]) < 1e-12
    assert np.allclose(runs["pipelined"][1], runs["plain"][1], rtol=1e-11)
    assert torch.equal(runs["pipelined"][2], runs["plain"][2])


@pytest.mark.parametrize("m", [1, 2, 3, 5, 15, 16, 17, 18, 31, 32, 33, 48, 49, 63, 64, 65, 80, 81, 97, 112, 113, 127])
def test_solve_and_coloured_noise_in_one_launch_at_every_rank(P, route, m):
    """csrc/ipb_prep.h against the launches it replaces (two triangular products, fill, third product) for ranks on both sides
    of every 16-row tile edge and ragged column counts: same V, same draws, to rounding."""
    n = 200 + m
    for j in (1, 16, 17, 50):
        pr = make_problem(n, m, j, 2, seed=13 * m + j + FUZZ_SEED)
        pr["ls"] = pr["ls"] * 0.3
        ob, gb = build_ipb(P, pr)
        name, oc, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
        u = cu(pr["u"])
        route(2)
        spec = P.basis.NoiseSpec(seed=5, step=7)
        e1 = torch.empty(j, device="cuda")
        got = gb.fused_step(gc, u, 1e-3, noise=spec, input_energy=e1)
        assert _timeline_names(P, lambda: gb.fused_step(gc, u, 1e-3, noise=spec)) == ["ipb_prep", "small_rank_step"]
        _ipb_prep(P, 0)
        try:
            e0 = torch.empty(j, device="cuda")
            want = gb.fused_step(gc, u, 1e-3, noise=spec, input_energy=e0)
        finally:
            _ipb_prep(P, 1)
        assert torch.isfinite(got).all()
        assert relerr(got, want) < 1e-11 and relerr(e1, e0) < 1e-11, (m, j)


@pytest.mark.parametrize("j,k", [(40, 4), (300, 8)])
def test_inducing_point_captured_training_equals_the_plain_loop(P, route, j, k):
    """train_pls_captured (K steps + energies per hipGraph replay: the capture owns the counters of the one-launch step and asks
    for the 256-column chunk sums, the kernel's second hand-over) on the inducing-point basis with a cost without the Gaussian
    algebra, against the plain loop over the same counter-based noise stream."""
    pr = make_problem(400, 12, j, 2, seed=19 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    name, oc, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
    pls = P.pkg.PLS(gb, gc)
    eta = 0.25 * float(torch.linalg.eigvalsh(ob.base_gram_induce).min()) / 12
    seed, epochs = 4242, 21
    route(1)
    u = cu(pr["u"])
    nxt = torch.empty_like(u)
    want_e = []
    for t in range(epochs):
        gb.fused_step(gc, u, eta, out=nxt, new_state=True, noise=P.basis.NoiseSpec(seed=seed, step=t))
        u, nxt = nxt, u
        want_e.append(pls.particle_energy_potential(u).mean().item())
    got_u, got_e = P.pkg.train_pls_captured(pls, cu(pr["u"]), epochs, eta, 1e9, steps_per_replay=k, seed=seed)
    assert len(got_e) == epochs and np.isfinite(got_e).all()
    assert torch.equal(got_u, u)
    assert np.allclose(got_e, want_e, rtol=1e-11)


@pytest.mark.parametrize("n,m,j,d", [(100, 10, 64, 1), (333, 17, 37, 2), (1000, 32, 100, 2), (1100, 128, 90, 3), (520, 65, 16, 2)])
def test_whitened_step_of_every_cost_against_the_oracle(P, route, n, m, j, d):
    """pls_ipb_whitened_generic_step (inducing_point.py:117-150 in the coordinates S = Lc^-1 U: ONE launch, the prior as M rows of
    the forward operand, white noise) against the oracle's update in the ORIGINAL coordinates with the coloured noise e = Lc xi
    injected: U + dU = Lc (S + dS), the energies of the input particles, delta and new-state forms, column blocks with their own
    step sizes; and against the library's own step in the original coordinates over the same Philox draws."""
    pr = make_problem(n, m, j, d, seed=17 * n + m + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    lc = torch.linalg.cholesky(ob.base_gram_induce)
    cond = torch.linalg.cond(ob.base_gram_induce).item()
    u = pr["u"]
    xi = torch.randn(m, j, generator=pr["gen"])
    e_noise = lc @ xi
    eta = 1e-3
    s_dev = gb.whiten(cu(u))
    assert relerr(s_dev, torch.linalg.solve_triangular(lc, u, upper=False)) < 1e-11 * max(1.0, cond / 1e5)
    checked = 0
    route(1)
    for name, oc, gc in make_costs(P, pr["y"], pr["fstar"], pr["gen"]):
        if name == "gaussian/identity":
            continue  # (its whitened route is the M x M x J contraction of test_gpu_whitened.py)
        assert gb.whitened_generic_applies(gc, j), name
        want = O.PLS(ob, oc).calculate_particle_update(u.clone(), eta, noise=e_noise)
        tol = step_tolerance(ob, oc, u, eta, e_noise, want)
        if tol >= 1e-8:
            continue
        tol = tol * max(1.0, cond / 1e5)
        checked += 1
        e_in = torch.full((j,), float("nan"), device="cuda")
        spec = P.basis.NoiseSpec(injected=cu(xi))  # (white: the xi of e = Lc xi)
        names = _timeline_names(P, lambda: gb.whitened_step(gc, s_dev, eta, noise=spec, input_energy=e_in))
        assert names == ["small_rank_step"], names
        ds = gb.whitened_step(gc, s_dev, eta, noise=spec)
        s_new = gb.whitened_step(gc, s_dev, eta, noise=spec, new_state=True, input_energy=e_in)
        assert relerr(s_new, s_dev + ds) < 1e-14
        assert relerr(gb.unwhiten(ds), want) < tol, name
        assert relerr(gb.unwhiten(s_new), u + want) < max(tol, 1e-12), name
        v = torch.cholesky_solve(u, lc)
        e_want = oc.calculate_cost(ob.calculate_untransformed_train_prediction_samples(u)) + 0.5 * m * (v * v).sum(dim=0)
        assert relerr(e_in, e_want) < 1e-9 * max(1.0, cond / 1e5), name
        assert relerr(gb.whitened_particle_energy(gc, s_dev), e_in) < 1e-12, name
        # the library's step in the original coordinates, Philox noise: the same draws
        pspec = P.basis.NoiseSpec(seed=31, step=2)
        du = gb.fused_step(gc, cu(u), eta, noise=pspec)
        assert relerr(gb.unwhiten(gb.whitened_step(gc, s_dev, eta, noise=pspec)), du) < tol, name
        # two column blocks with their own step sizes
        if j >= 4:
            half = j // 2
            etas = torch.tensor([eta, 3.0 * eta], device="cuda")
            blocks = P.basis.BlockSpec(half, etas)
            jj = 2 * half
            sb = s_dev[:, :jj].contiguous()
            got_b = gb.whitened_step(gc, sb, 0.0, noise=P.basis.NoiseSpec(none=True), blocks=blocks)
            a = gb.whitened_step(gc, sb[:, :half].contiguous(), eta, noise=P.basis.NoiseSpec(none=True))
            b = gb.whitened_step(gc, sb[:, half:].contiguous(), 3.0 * eta, noise=P.basis.NoiseSpec(none=True))
            assert relerr(got_b, torch.cat([a, b], dim=1)) < 1e-12, name
    assert checked >= 5, checked
    assert int(gb._step_sync(j, torch.device("cuda")).abs().sum()) == 0


def test_inducing_point_routes_are_invariant_under_particle_sharding(P, route):
    """a rank of a J-sharded run holds a block of particle columns and draws the noise of its GLOBAL columns.  The
    inducing-point basis' small routes -- a single call (solve + coloured noise in one launch, then the step) and the whitened
    step of a loop -- give, shard by shard, the columns of the unsharded call."""
    pr = make_problem(400, 20, 96, 2, seed=29 + FUZZ_SEED)
    pr["ls"] = pr["ls"] * 0.35
    ob, gb = build_ipb(P, pr)
    name, oc, gc = make_costs(P, pr["y"], pr["fstar"], pr["gen"])[2]
    u = cu(pr["u"])
    s = gb.whiten(u)
    route(1)
    assert gb.whitened_generic_applies(gc, 96)
    full = gb.fused_step(gc, u, 1e-3, noise=P.basis.NoiseSpec(seed=5, step=7, j_offset=0))
    full_w = gb.whitened_step(gc, s, 1e-3, noise=P.basis.NoiseSpec(seed=5, step=7, j_offset=0))
    assert relerr(gb.unwhiten(full_w), full) < 1e-10
    for world in (2, 3):
        parts, parts_w = [], []
        for r in range(world):
            j0, j1 = P.dist.shard_bounds(96, r, world)
            spec = P.basis.NoiseSpec(seed=5, step=7, j_offset=j0)
            parts.append(gb.fused_step(gc, u[:, j0:j1].contiguous(), 1e-3, noise=spec))
            parts_w.append(gb.whitened_step(gc, s[:, j0:j1].contiguous(), 1e-3, noise=spec))
        assert relerr(torch.cat(parts, dim=1), full) < 1e-13
        assert relerr(torch.cat(parts_w, dim=1), full_w) < 1e-13
