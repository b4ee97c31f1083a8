"""CPU: host-side logic of the drop-in layer that needs no GPU (sharding math, stop rule, descriptors,
argument validation in the C ABI, loud failure without a device)."""
import math

import numpy as np
import pytest
import torch

import projected_langevin_sampling_amd as pkg
from projected_langevin_sampling_amd import costs, distributed, link_functions
from projected_langevin_sampling_amd.basis.base import NoiseSpec, padded_ld

L = pkg._lib


def test_shard_bounds_partition_every_column_once():
    for j in (0, 1, 7, 64, 8192, 8193):
        for world in (1, 2, 3, 4, 8):
            bounds = [distributed.shard_bounds(j, r, world) for r in range(world)]
            assert bounds[0][0] == 0 and bounds[-1][1] == j
            assert all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
            widths = [b - a for a, b in bounds]
            assert max(widths) - min(widths) <= 1
    with pytest.raises(ValueError):
        distributed.shard_bounds(8, 3, 2)


def test_early_stopper_matches_reference_rule():
    es = pkg.EarlyStopper(patience=0.25)
    seq = [(1.0, False), (1.0, False), (2.0, False), (0.5, False), (0.5, False), (0.5, False), (0.5, True)]
    for loss, want in seq:
        assert es.should_stop(loss, 0.1) is want
    assert pkg.EarlyStopper().should_stop(float("inf"), 1e-3) is True
    assert pkg.EarlyStopper().should_stop(float("nan"), 1e-3) is True


def test_cost_descriptors():
    y = torch.zeros(4)
    d = costs.GaussianCost(0.3, y, link_functions.IdentityLinkFunction()).desc()
    assert (d.cost, d.link, d.deriv_mode, d.p[0]) == (L.COST_GAUSSIAN, L.LINK_IDENTITY, L.DERIV_REFERENCE, 0.3)
    d = costs.StudentTCost(3, y, link_functions.IdentityLinkFunction(), scale=0.5).desc(force_autograd=True)
    assert (d.cost, d.deriv_mode, d.p[0], d.p[1]) == (L.COST_STUDENT_T, L.DERIV_AUTOGRAD, 3.0, 0.5)
    d = costs.MultiModalCost(0.7, 1.5, 0.3, y, link_functions.SigmoidLinkFunction(jitter=1e-6)).desc()
    assert (d.cost, d.link, d.p[0], d.p[1], d.p[2], d.jitter) == (L.COST_MULTIMODAL, L.LINK_SIGMOID, 0.7, 1.5, 0.3, 1e-6)
    assert costs.BernoulliCost(torch.tensor([0, 1]), link_functions.ProbitLinkFunction()).y_train.dtype == torch.float64

    class UserLink(link_functions.PLSLinkFunction):
        def transform(self, y):
            return y

    c = costs.PoissonCost(y, UserLink())
    assert not c.is_native()
    with pytest.raises(L.PlsHipError):
        c.desc()


def test_noise_spec_descriptor():
    d = NoiseSpec(seed=2**64 + 5, step=9, j_offset=4096).desc()
    assert (d.kind, d.seed, d.step, d.j_offset) == (L.NOISE_PHILOX, 5, 9, 4096)
    assert NoiseSpec(none=True).desc().kind == L.NOISE_NONE
    with pytest.raises(L.PlsHipError):
        NoiseSpec(injected=torch.zeros(2, 2, dtype=torch.float64)).desc()  # CPU tensor: no fallback


def test_padded_leading_dimension():
    assert [padded_ld(c) for c in (1, 16, 17, 1021, 1024)] == [16, 16, 32, 1024, 1024]


def test_cabi_argument_validation_without_a_gpu():
    """Validation happens before any HIP call, so these run on the CPU-only box."""
    lib = L.load()
    assert lib.pls_gemm_tn(None, 1, None, 1, None, 1, 1, 1, 1, 1.0, 0.0, None) == 1
    assert b"NULL" in lib.pls_last_error()
    assert lib.pls_gemm_tn(8, 1, 8, 1, 8, 1, 4, 4, 4, 1.0, 0.0, None) == 1  # ld < size
    assert b"leading dimension" in lib.pls_last_error()
    bad = L.CostDesc()
    bad.cost, bad.link = 0, 7
    assert lib.pls_cost_derivative(bad, 8, 1, 8, 1, 1, 8, 1, None) == 1
    assert b"unknown link" in lib.pls_last_error()
    g = L.CostDesc()
    g.cost, g.link, g.p[0] = L.COST_GAUSSIAN, L.LINK_IDENTITY, 0.0
    assert lib.pls_cost_derivative(g, 8, 1, 8, 1, 1, 8, 1, None) == 1
    assert b"observation_noise" in lib.pls_last_error()
    o = L.OnbDesc()
    assert lib.pls_onb_forward(o, 8, 1, 1, 8, 1, None) == 1
    assert lib.pls_kernel_gram(0, 8, 1, 8, 1, 100, 8, 1.0, 8, 1, None) == 1  # d > 64
    assert lib.pls_cost_value_workspace_bytes(1000, 8) == 4 * 8 * 8
    assert lib.pls_onb_step_workspace_bytes(None, 8, 0) == 0


def test_product_fails_loudly_without_a_device():
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    k = pkg.ARDKernel([1.0], 1.0)
    with pytest.raises(L.PlsHipError, match="no CPU fallback"):
        k(torch.zeros(2, 1), torch.zeros(3, 1))
    c = costs.GaussianCost(0.3, torch.zeros(4), link_functions.IdentityLinkFunction())
    with pytest.raises(L.PlsHipError, match="no CPU fallback"):
        c.calculate_cost(torch.zeros(4, 2, dtype=torch.float64))


def test_predictive_moments_single_process():
    s = torch.randn(5, 64, dtype=torch.float64)
    mean, var = distributed.predictive_moments(s, 64)
    assert np.allclose(mean, s.mean(dim=1)) and np.allclose(var, s.var(dim=1))
    assert np.isclose(distributed.mean_over_particles(s[0], 64), s[0].mean().item())


def test_checkpoint_format_matches_the_reference(tmp_path, monkeypatch):
    """experiments/uci/regression/main.py:300-308 / experiments/loaders.py:10-28: same keys, readable by plain torch.load."""
    from projected_langevin_sampling_amd import checkpoint

    class FakePLS:
        observation_noise = 0.25

    u = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    path = str(tmp_path / "pls.pth")
    checkpoint.save_pls(FakePLS(), u, path, best_lr=1e-3, number_of_epochs=17, noise_step=5, number_of_particles=64)
    raw = torch.load(path, map_location="cpu")
    assert set(raw) >= {"particles", "observation_noise", "best_lr", "number_of_epochs"}
    assert torch.equal(raw["particles"], u) and raw["observation_noise"] == 0.25 and raw["best_lr"] == 1e-3
    # a file written by the reference (no extra keys, float32 particles) loads too
    torch.save({"particles": u.float(), "observation_noise": 0.5}, path)
    monkeypatch.setattr(checkpoint, "_dev", lambda t: t.double())  # CPU-only box: skip the device move
    p = FakePLS()
    _, particles, best_lr, epochs = checkpoint.load_pls(p, path)
    assert particles.dtype == torch.float64 and torch.equal(particles, u) and p.observation_noise == 0.5
    assert best_lr is None and epochs is None


def test_step_size_search_skeleton_matches_the_reference_rules():
    """experiments/runners.py:331-446 with a scripted training function: which candidates run, which is kept, when it stops."""
    from projected_langevin_sampling_amd.runners import train_pls_runner

    calls = []
    # energies per candidate step size (index): 0 diverges (nan particles), 1 and 2 converge, 3 equals 2 -> search stops after 3
    script = {0: ([5.0], float("nan")), 1: ([9.0, 4.0], 1.0), 2: ([9.0, 3.0], 2.0), 3: ([9.0, 3.0000001], 3.0), 4: ([1.0], 4.0)}

    def fake_train(pls, particles, number_of_epochs, step_size, early_stopper_patience):
        i = len(calls)
        calls.append((step_size, number_of_epochs, torch.initial_seed()))
        energies, fill = script[i]
        return torch.full_like(particles, fill), list(energies)

    p0 = torch.zeros(2, 3)
    out, best_lr, n_energy = train_pls_runner(
        pls=None, particle_name="t", x_train=None, y_train=None, simulation_duration=1.0, maximum_number_of_steps=1000,
        early_stopper_patience=1.0, number_of_step_searches=5, step_size_upper=0.1, minimum_change_in_energy_potential=1e-6,
        seed=11, particles=p0, metric_to_optimise="loss", train_fn=fake_train)
    steps = np.logspace(np.log10(0.1), np.log10(1e-3), 5)
    assert len(calls) == 4 and np.allclose([c[0] for c in calls], steps[:4])  # stopped after the 4th (relative change < 1e-6)
    assert [c[1] for c in calls] == [int(1.0 / s) for s in steps[:4]] and all(c[2] == 11 for c in calls)
    assert best_lr == steps[2] and n_energy == 2 and torch.all(out == 2.0)  # candidate 3's energy is not lower than 2's
    assert torch.all(p0 == 0)  # the caller's particles are never modified
    with pytest.raises(NotImplementedError):
        train_pls_runner(None, "t", None, None, 1.0, 10, 1.0, 2, 0.1, 1e-3, 0, p0, metric_to_optimise="bogus", train_fn=fake_train)


def test_metrics_match_their_definitions():
    from projected_langevin_sampling_amd import metrics

    m, v, y = torch.tensor([1.0, 2.0]), torch.tensor([0.5, 2.0]), torch.tensor([1.5, 0.0])
    d = torch.distributions.MultivariateNormal(m, covariance_matrix=torch.diag(v))
    assert np.isclose(metrics.calculate_mae(d, y), 1.25) and np.isclose(metrics.calculate_mse(d, y), (0.25 + 4.0) / 2)
    want = (0.5 * (torch.log(2 * torch.pi * v) + (y - m) ** 2 / v)).mean().item()
    assert np.isclose(metrics.calculate_nll(d, y), want)
    b = torch.distributions.Bernoulli(probs=torch.tensor([0.9, 0.2]))
    assert np.isclose(metrics.calculate_nll(b, torch.tensor([1.0, 0.0])), -(np.log(0.9) + np.log(0.8)) / 2)


def test_selector_accepts_a_scale_kernel_shaped_object_and_rejects_the_pls_kernel():
    """The reference hands the selector a gpytorch ScaleKernel(RBFKernel) (experiments/uci/regression/main.py:203):
    lengthscale AND outputscale must both be read; a PLSKernel (r, not k) is refused instead of being silently
    reduced to its base kernel.  Host-only: no Gram matrix is built here."""
    from projected_langevin_sampling_amd.inducing_point_selectors import ConditionalVarianceInducingPointSelector
    from projected_langevin_sampling_amd.kernel import ARDKernel, PLSKernel, as_base_kernel

    class RBFStub:
        lengthscale = torch.tensor([[0.5, 2.0]])

    class ScaleStub:
        base_kernel = RBFStub()
        outputscale = torch.tensor(3.0)

    k = as_base_kernel(ScaleStub())
    assert isinstance(k, ARDKernel) and k.outputscale == 3.0 and k.lengthscale.tolist() == [0.5, 2.0]
    with pytest.raises(TypeError):
        as_base_kernel(RBFStub())  # an un-scaled inner kernel is not what the reference passes
    sel = ConditionalVarianceInducingPointSelector()
    with pytest.raises(TypeError, match="base kernel"):
        sel(torch.zeros(8, 2), 3, PLSKernel(ARDKernel([1.0, 1.0]), torch.zeros(4, 2)))


def test_block_spec_and_batched_runner_bookkeeping():
    """Host side of the batched step-size search: candidate grid, epochs, and the ledger's rules in candidate order."""
    from projected_langevin_sampling_amd.runners import _CandidateRun, _SearchLedger, candidate_step_sizes

    steps = candidate_step_sizes(1e-2, 1.0, 1000, 4)
    assert np.allclose(steps, np.logspace(-2, -3, 4)) and steps[0] > steps[-1]
    runs = [_CandidateRun(float(s), int(1.0 / s)) for s in steps]
    assert [r.number_of_epochs for r in runs] == [100, 215, 464, 1000] or runs[0].number_of_epochs == 99
    ledger = _SearchLedger(None, "loss", None, None, 1e-3, fallback_particles=torch.zeros(2, 2))
    # candidate 1 finishes before candidate 0: nothing is judged out of order
    runs[1].particles, runs[1].energies, runs[1].finished = torch.ones(2, 2), [5.0, 4.0], True
    ledger.judge_ready(runs)
    assert ledger.cursor == 0 and ledger.best_step_size is None
    runs[0].particles, runs[0].energies, runs[0].finished = torch.full((2, 2), float("inf")), [9.0], True  # diverged: not accepted
    ledger.judge_ready(runs)
    assert ledger.cursor == 2 and ledger.best_step_size == runs[1].step_size and not ledger.closed
    runs[2].particles, runs[2].energies, runs[2].finished = 2 * torch.ones(2, 2), [4.5, 4.0001], True
    ledger.judge_ready(runs)  # |4.0 - 4.0001| / 4.0 < 1e-3: consecutive accepted runs agree -> the search closes
    assert ledger.closed and ledger.cursor == 3
    best, lr, n = ledger.result()
    assert lr == runs[1].step_size and n == 2 and torch.equal(best, torch.ones(2, 2))
    with pytest.raises(NotImplementedError):
        _SearchLedger(None, "bogus", None, None, 1e-3, fallback_particles=torch.zeros(1, 1))


def test_bench_self_launches_its_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher starts N ranks under torch.distributed.run as a CHILD process
    (rendezvous on 127.0.0.1) before anything touches the GPU, and returns the child's exit code."""
    import subprocess
    import sys

    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_whitened_route_and_inverse_factor_products_are_as_accurate_as_substitution():
    """Numerics behind round 3's inducing-point Gaussian step, in plain torch on the host.
    With ONE Cholesky factor of k(Z,Z) (cond 2e10 here) the whitened route  Lc (Q S - c~),  Q = Lc^-1 (B / sigma2 + M I) Lc^-T,
    S = Lc^-1 U  -- with Lc^-1 applied as an explicit triangular inverse or by substitution -- equals the reference's
    (B V - c) / sigma2 + M V,  V = k(Z,Z)^-1 U  (inducing_point.py:130-150, gaussian.py:86-88) to 1e-12, whereas two
    equally valid factors of the same matrix (LAPACK's and a right-looking blocked one) already move either route by
    ~cond * 1e-17: the factorisation, not the way its inverse is applied, sets the accuracy."""
    import torch

    from oracle import pls_oracle as O

    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        g = torch.Generator().manual_seed(6660)
        n, m, j, sigma2 = 600, 60, 50, 0.3
        x = torch.rand(n, 2, generator=g) * 2 - 1
        z = x[torch.randperm(n, generator=g)[:m]].clone()
        y = torch.sin(2.0 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
        u = torch.randn(m, j, generator=g)
        kern = O.RBFARDKernel(torch.tensor([0.55, 0.75]), 1.3)
        k, kzx = kern(z, z), kern(z, x)
        cond = torch.linalg.cond(k).item()
        assert 1e7 < cond < 1e12, cond
        b, c = kzx @ kzx.T, kzx @ y

        def blocked(a, nb=8):  # right-looking blocked Cholesky: the rounding pattern of pls_chol_factor, not LAPACK's
            a = a.clone()
            for s in range(0, m, nb):
                e = min(m, s + nb)
                a[s:e, s:e] = torch.linalg.cholesky(a[s:e, s:e])
                if e < m:
                    a[e:, s:e] = torch.linalg.solve_triangular(a[s:e, s:e], a[e:, s:e].T, upper=False).T
                    a[e:, e:] -= a[e:, s:e] @ a[e:, s:e].T
            return torch.tril(a)

        def reference_route(lc):
            v = torch.cholesky_solve(u, lc)
            return (b @ v - c[:, None]) / sigma2 + m * v

        def whitened_route(lc, explicit_inverse):
            if explicit_inverse:
                linv = torch.linalg.solve_triangular(lc, torch.eye(m), upper=False)
                fwd = lambda t: linv @ t
            else:
                fwd = lambda t: torch.linalg.solve_triangular(lc, t, upper=False)
            q = fwd(fwd(b / sigma2 + m * torch.eye(m)).T)
            return lc @ (q @ fwd(u) - fwd(c[:, None]) / sigma2)

        rel = lambda a, r: ((a - r).abs().max() / r.abs().max()).item()
        l0, l1 = torch.linalg.cholesky(k), blocked(k)
        for lc in (l0, l1):
            ref = reference_route(lc)
            assert rel(whitened_route(lc, True), ref) < 1e-11 and rel(whitened_route(lc, False), ref) < 1e-11
        moved = rel(reference_route(l1), reference_route(l0))
        assert moved > 1e-11, "two factors of an ill-conditioned matrix are expected to disagree beyond the route's own rounding"
        assert moved < cond * 1e-15
    finally:
        torch.set_default_dtype(prev)


def test_eigh_device_resolution():
    """samplers.resolve_eigh_device: an explicit request wins, then the module default; "auto" = where the matrix lives
    (the reference's torch.linalg.eigh(matrix) semantics).  conftest pins the default to "cpu" for the parity tests."""
    from projected_langevin_sampling_amd import samplers

    m = torch.eye(3, dtype=torch.float64)
    assert samplers.DEFAULT_EIGH_DEVICE == "cpu"  # (the autouse fixture of tests/conftest.py)
    assert samplers.resolve_eigh_device(None, m) == "cpu"
    assert samplers.resolve_eigh_device("auto", m) == "cpu"  # a host matrix
    assert samplers.resolve_eigh_device("cuda", m) == "cuda"
    prev = samplers.DEFAULT_EIGH_DEVICE
    try:
        samplers.DEFAULT_EIGH_DEVICE = "auto"
        assert samplers.resolve_eigh_device(None, m) == "cpu"
        assert samplers.resolve_eigh_device("cpu", m) == "cpu"
    finally:
        samplers.DEFAULT_EIGH_DEVICE = prev
    with pytest.raises(AssertionError):
        samplers.resolve_eigh_device("tpu", m)

    class OnDevice:  # (no GPU here: what resolve_eigh_device looks at)
        is_cuda = True

        def __init__(self, n):
            self.shape = (n, n)

    # "auto": small matrices go to host LAPACK wherever they live (the reference's benchmark sizes: all latency on the device)
    assert samplers.EIGH_HOST_BELOW == 128
    assert samplers.resolve_eigh_device("auto", OnDevice(100)) == "cpu" and samplers.resolve_eigh_device("auto", OnDevice(128)) == "cpu"
    assert samplers.resolve_eigh_device("auto", OnDevice(129)) == "cuda" and samplers.resolve_eigh_device("cuda", OnDevice(10)) == "cuda"
    # the host call narrows torch's thread pool for small matrices and puts it back
    before = torch.get_num_threads()
    g = torch.randn(20, 20, dtype=torch.float64)
    lam, vec = samplers.host_eigh(g @ g.T)
    assert torch.get_num_threads() == before and torch.allclose((vec * lam) @ vec.T, g @ g.T, atol=1e-10)


def test_sign_canonicalisation_and_spectrum_fingerprint(tmp_path, monkeypatch):
    """basis/spectrum.py: two eigensolvers that agree up to the sign of every eigenvector give ONE canonical matrix; the
    fingerprint a checkpoint carries tells a flipped sign, a rotation inside a cluster, another count and another kernel
    apart from rounding noise; load_pls refuses particles of another gauge (experiments/loaders.py:10-28 would restore them
    into whatever basis the caller rebuilt)."""
    from projected_langevin_sampling_amd import checkpoint
    from projected_langevin_sampling_amd.basis import spectrum as S

    g = torch.Generator().manual_seed(4)
    a = torch.randn(12, 12, generator=g, dtype=torch.float64)
    lam, vec = torch.linalg.eigh(a @ a.T / 12)
    flips = torch.tensor([1.0, -1.0] * 6, dtype=torch.float64)
    c0, c1 = S.canonicalise_signs(vec), S.canonicalise_signs(vec * flips[None, :])
    assert torch.equal(c0, c1)
    assert bool((c0.gather(0, c0.abs().argmax(dim=0)[None, :]) > 0).all())
    assert torch.allclose(c0 @ torch.diag(lam) @ c0.T, a @ a.T / 12, atol=1e-12)  # still the same decomposition
    assert S.canonicalise_signs(torch.zeros(3, 0)).shape == (3, 0)

    fp = S.spectrum_fingerprint(lam, vec)
    assert fp["m"] == 12 and fp["mk"] == 12 and fp["probe"].shape == (12,)
    assert S.compare_fingerprints(fp, S.spectrum_fingerprint(lam, vec.clone())) is None  # same bits
    noisy = vec + 1e-13 * torch.randn(12, 12, generator=g, dtype=torch.float64)
    assert S.compare_fingerprints(fp, S.spectrum_fingerprint(lam * (1 + 1e-14), noisy)) is None  # rounding noise
    assert "opposite sign" in S.compare_fingerprints(fp, S.spectrum_fingerprint(lam, vec * flips[None, :]))
    th = 0.3
    rot = vec.clone()
    rot[:, 3], rot[:, 4] = math.cos(th) * vec[:, 3] - math.sin(th) * vec[:, 4], math.sin(th) * vec[:, 3] + math.cos(th) * vec[:, 4]
    assert "rotation" in S.compare_fingerprints(fp, S.spectrum_fingerprint(lam, rot))
    assert "eigen-directions" in S.compare_fingerprints(fp, S.spectrum_fingerprint(lam[1:], vec[:, 1:]))
    assert "eigenvalues differ" in S.compare_fingerprints(fp, S.spectrum_fingerprint(lam * 1.01, vec))

    class FakeBasis:
        def __init__(self, lam, vec):
            self.lam, self.vec = lam, vec

        def spectrum_fingerprint(self):
            return S.spectrum_fingerprint(self.lam, self.vec)

    class FakePLS:
        observation_noise = 0.25

        def __init__(self, basis):
            self.basis = basis

    monkeypatch.setattr(checkpoint, "_dev", lambda t: t.double())  # CPU-only box: skip the device move
    u = torch.randn(12, 5, generator=g, dtype=torch.float64)
    path = str(tmp_path / "pls.pth")
    checkpoint.save_pls(FakePLS(FakeBasis(lam, vec)), u, path)
    assert set(torch.load(path)) >= {"particles", "observation_noise", "best_lr", "number_of_epochs", "spectrum_fingerprint"}
    _, got, _, _ = checkpoint.load_pls(FakePLS(FakeBasis(lam, vec.clone())), path)
    assert torch.equal(got, u)
    other = FakePLS(FakeBasis(lam, vec * flips[None, :]))
    with pytest.raises(ValueError, match="opposite sign"):
        checkpoint.load_pls(other, path)
    with pytest.warns(UserWarning, match="opposite sign"):
        checkpoint.load_pls(other, path, on_gauge_mismatch="warn")
    checkpoint.load_pls(other, path, on_gauge_mismatch="ignore")

    class NoGauge:  # inducing-point basis, user-defined bases: particles are not gauge dependent
        observation_noise = 0.25
        basis = object()

    checkpoint.load_pls(NoGauge(), path)  # a basis without a fingerprint accepts any file ...
    checkpoint.save_pls(NoGauge(), u, path)
    assert "spectrum_fingerprint" not in torch.load(path)
    checkpoint.load_pls(other, path)  # ... and a file without one (the reference's own) loads into any basis


def test_normal_stream_resolution_and_shard_offset_forwarding():
    """samplers.resolve_normal_stream: the shipped default ("auto") is the reference's host stream for an unsharded run -- the
    pinned contract of the reference's tests/test_samplers.py:19-26 -- and the device stream as soon as the run is J-sharded;
    PLS.sample_observation_noise hands the shard's column offset to a native cost (costs/base.py:86-115 has no such
    argument: a user-defined cost is called with the reference's signature)."""
    from projected_langevin_sampling_amd import samplers

    assert samplers.DEFAULT_NORMAL_STREAM == "auto"
    assert samplers.resolve_normal_stream(None) == "reference"
    assert samplers.resolve_normal_stream(None, j_offset=128) == "device"
    assert samplers.resolve_normal_stream("reference", j_offset=128) == "reference"  # an explicit request wins
    assert samplers.resolve_normal_stream("device") == "device"
    with pytest.raises(AssertionError):
        samplers.resolve_normal_stream("philox")

    calls = []

    class Basis:
        j_offset = 0

    class NativeCost:
        observation_noise = 0.1

        def is_native(self):
            return True

        def sample_observation_noise(self, number_of_particles, seed=None, j_offset=0):
            calls.append(("native", number_of_particles, seed, j_offset))

    class UserCost:
        observation_noise = 0.1

        def sample_observation_noise(self, number_of_particles, seed=None):  # the reference's signature
            calls.append(("user", number_of_particles, seed))

    b = Basis()
    pkg.PLS(b, NativeCost()).sample_observation_noise(7, seed=3)
    b.j_offset = 40
    pkg.PLS(b, NativeCost()).sample_observation_noise(7, seed=3)
    pkg.PLS(b, UserCost()).sample_observation_noise(7, seed=3)
    assert calls == [("native", 7, 3, 0), ("native", 7, 3, 40), ("user", 7, 3)]


def test_boundary_dtypes_without_a_gpu():
    """_lib.require_gpu_tensor: host tensors are refused whatever their dtype (no CPU fallback); the TypeError for a wrong
    dtype names the conversion (exercised on the GPU in tests/test_gpu_parity.py)."""
    from projected_langevin_sampling_amd import _lib as L

    with pytest.raises(L.PlsHipError, match="no CPU fallback"):
        L.require_gpu_tensor(torch.zeros(2, dtype=torch.float32), "particles", promote=True)
    with pytest.raises(TypeError):
        L.require_gpu_tensor([1.0], "particles")
    assert torch.float32 in L.PROMOTED_DTYPES and torch.float64 not in L.PROMOTED_DTYPES


def test_the_partial_sum_exchange_of_the_balanced_products_is_tested_across_xcds():
    """csrc/gemm_tn_f64_kg.h, gemm_tn_f64_kg_tri_kernel: the two workgroups that share a pair of tile rows meet through
    write-through slots and one agent-scope atomic -- a protocol that must hold when they sit on DIFFERENT XCDs (whose L2s are
    not coherent with each other).  Blocks are dealt round-robin over the eight XCDs (blockIdx % 8 labels the XCD group), so a
    host-side replay of the kernel's block -> (virtual row, tile column) map says which SHAPES of
    tests/test_gpu_tri_balance.py put the two roles of a pair into different groups: the GPU test must contain such shapes
    by design, not by accident of a remap."""
    from test_gpu_tri_balance import SHAPES

    def tile_coords(bid, nti, ntj):  # gemm_tile_coords of csrc/gemm_tn_f64.h
        nwg = nti * ntj
        xcd, q, r = bid & 7, nwg >> 3, nwg & 7
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        ident = base + (bid >> 3)
        gi_max = 8
        group = gi_max * ntj
        g = ident // group
        first_i = g * gi_max
        gi = min(nti - first_i, gi_max)
        in_g = ident - g * group
        return first_i + in_g % gi, in_g // gi

    split = {}
    for (m, j) in SHAPES:
        nti, ntj = (m + 63) // 64, (j + 63) // 64
        pairs = (nti + 1) // 2
        where = {}
        for bid in range(2 * pairs * ntj):
            vrow, tj = tile_coords(bid, 2 * pairs, ntj)
            assert 0 <= vrow < 2 * pairs and 0 <= tj < ntj
            assert (vrow, tj) not in where, "the map must be a bijection"
            where[(vrow, tj)] = bid & 7
        assert len(where) == 2 * pairs * ntj
        split[(m, j)] = sum(1 for p in range(pairs) for tj in range(ntj) if where[(2 * p, tj)] != where[(2 * p + 1, tj)])
    crossing = {k: v for k, v in split.items() if v > 0}
    assert len(crossing) >= 3, f"shapes whose pairs cross XCD groups: {crossing}"
    assert all(split[k] > 0 for k in ((192, 192), (200, 130), (1088, 320))), split  # (the ones the round-4 review replayed)


def test_slab_plan_of_the_one_launch_step(tmp_path):
    """csrc/small_rank_step.h, small_rank_step_splits: the host-side plan of the one-launch small-rank step (slabs per column block,
    rows per slab), compiled from the header itself and checked over a grid of sizes -- rows in whole rounds of 64, every row
    covered, no empty slab, several slabs only where they fit one workgroup per CU and are at least two rounds long; and the cases
    the plan was tuned on (a few prior rows past a round must not cost every slab a round)."""
    import os
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "plan.cpp"
    src.write_text('''#include <cstdio>
#include <cstdint>
#include <hip/hip_runtime.h>
#include "small_rank_step.h"
int main() {
  const int64_t js[] = {1, 16, 48, 64, 100, 330, 512, 1000, 4096};
  const int64_t ns[] = {1, 40, 100, 110, 600, 1000, 1032, 1530, 3000, 4096, 4224, 20000};
  const int ks[] = {1, 10, 16, 32, 33, 64, 100, 128};
  for (int64_t j : js) for (int64_t n : ns) for (int k : ks) {
    int64_t rows = 0;
    const int64_t s = plship::small_rank_step_splits(j, n, k, &rows);
    printf("%ld %ld %d %ld %ld\\n", (long)j, (long)n, k, (long)s, (long)rows);
  }
  return 0;
}
''')
    exe = tmp_path / "plan"
    subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(root, "projected-langevin-sampling_amd", "csrc"),
                    "-o", str(exe), str(src)], check=True, capture_output=True, timeout=300)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=60).stdout
    plans = {}
    for line in out.splitlines():
        j, n, k, s, rows = (int(v) for v in line.split())
        plans[(j, n, k)] = (s, rows)
        ncb = (j + 15) // 16
        assert s >= 1 and rows >= 64 and rows % 64 == 0, line
        assert s * rows >= n and (s - 1) * rows < n, f"rows not covered exactly once / an empty slab: {line}"
        if s > 1:
            assert ncb * s <= 256, f"more workgroups than CUs: {line}"
            assert rows >= 128, f"a slab shorter than two rounds: {line}"
    assert plans[(64, 100, 10)] == (1, 128) and plans[(64, 110, 10)] == (1, 128)  # launch-bound: one slab
    assert plans[(100, 1000, 32)] == (8, 128)
    assert plans[(100, 1032, 32)] == (9, 128)  # 1000 data rows + 32 prior rows: one short slab more, not a third round for all
    assert plans[(512, 4096, 128)] == (8, 512)
    assert plans[(512, 4224, 128)] == (8, 576)  # (nine slabs would be 288 workgroups)
