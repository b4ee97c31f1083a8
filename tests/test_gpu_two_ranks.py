"""GPU, two processes on ONE card over gloo: the J-sharded run end to end through the drop-in classes.

No multi-GPU node is reachable from here, so RCCL with more than one rank cannot run; what CAN run is every line of the
sharded control flow with real device kernels: both ranks build the orthonormal basis under an initialised process group
(rank 0's eigh, broadcast: basis/spectrum.py), attach their column shards, step with the library's noise keyed by GLOBAL
column indices, reduce the energy and the predictive moments across ranks, and save / resume a checkpoint -- and the parent
process holds the concatenated shards to the unsharded run, bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    g = torch.Generator().manual_seed(77)
    n, m, d, j = 3000, 200, 3, 330
    x = torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1
    z = x[torch.randperm(n, generator=g)[:m]].clone()
    w = torch.randn(d, generator=g, dtype=torch.float64)
    y = torch.sin(2.0 * (x @ w)) + 0.1 * torch.randn(n, generator=g, dtype=torch.float64)
    ls = 0.3 + 0.3 * torch.rand(d, generator=g, dtype=torch.float64)
    xs = torch.rand(11, d, generator=g, dtype=torch.float64) * 2 - 1
    return x, z, y, ls, xs, j


def _build(eigh_device, group=None):
    import projected_langevin_sampling_amd as pkg
    from projected_langevin_sampling_amd.basis import OrthonormalBasis
    from projected_langevin_sampling_amd.costs import GaussianCost
    from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

    x, z, y, ls, xs, j = _problem()
    basis = OrthonormalBasis(pkg.PLSKernel(pkg.ARDKernel(ls, 1.2), z), z, x, 1e-6, verbose=False, eigh_device=eigh_device, group=group)
    cost = GaussianCost(0.3, y, IdentityLinkFunction())
    return pkg, basis, cost, xs, j


def _build_small_ipb():
    """the inducing-point basis at the reference's curve-experiment scale with a cost without the Gaussian algebra: the loop that
    keeps its particles whitened (one launch per iteration, csrc/small_rank_step.h over pls_ipb_desc.Awa)"""
    import projected_langevin_sampling_amd as pkg
    from projected_langevin_sampling_amd.basis import InducingPointBasis
    from projected_langevin_sampling_amd.costs import BernoulliCost
    from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction

    x, z, y, ls, xs, j = _problem()
    x, y, z = x[:700].contiguous(), y[:700].contiguous(), z[:24].contiguous()
    basis = InducingPointBasis(pkg.PLSKernel(pkg.ARDKernel(ls, 1.2), z), z, y[:24], x)
    cost = BernoulliCost((y > 0).double(), SigmoidLinkFunction())
    eta = 0.25 * float(torch.linalg.eigvalsh(basis.base_gram_induce.cpu()).min()) / 24
    u0 = torch.randn(24, j, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    return pkg, basis, cost, u0, eta, j


def _steps(basis, cost, u, first, count, seed=4242):
    from projected_langevin_sampling_amd.basis import NoiseSpec

    eta = 0.5 * float(basis.eigenvalues.min())  # eta / lambda_min < 2: a stable chain, rounding stays rounding
    cur, nxt = u.clone(), torch.empty_like(u)
    for t in range(first, first + count):
        basis.fused_step(cost, cur, eta, out=nxt, new_state=True, noise=NoiseSpec(seed=seed, step=t, j_offset=basis.j_offset))
        cur, nxt = nxt, cur
    return cur


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_default_dtype(torch.float64)
    torch.cuda.set_device(0)  # both ranks share the one card of this box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from projected_langevin_sampling_amd import checkpoint, samplers
    from projected_langevin_sampling_amd import distributed as D

    samplers.DEFAULT_NORMAL_STREAM = "device"
    pkg, basis, cost, xs, j = _build("cuda", group=True)  # collective (opt-in): rank 0 factorises, everybody receives the same bits
    mk = basis.approximation_dimension
    j0, j1 = D.attach_shard(basis, j, rank, world)
    u0 = torch.randn(mk, j, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    mine = _steps(basis, cost, u0[:, j0:j1].contiguous().cuda(), 0, 6)
    pls = pkg.PLS(basis, cost)
    # C1: the mean energy over ALL particles
    energy = D.mean_over_particles(pls.particle_energy_potential(mine), j)
    # C2: predictive moments over ALL particles, noise keyed by the global column
    torch.manual_seed(9)
    samples = pls.predict_samples(mine, xs)
    mean, var = D.predictive_moments(samples, j)
    # checkpoint: save mid-run, rebuild the basis (the same collective), resume
    path = os.path.join(out_dir, f"rank{rank}.pth")
    mid = _steps(basis, cost, u0[:, j0:j1].contiguous().cuda(), 0, 3)
    checkpoint.save_pls(pls, mid, path, noise_step=3, number_of_particles=j)
    pkg2, basis2, cost2, _, _ = _build("cuda", group=True)
    D.attach_shard(basis2, j, rank, world)
    _, restored, _, _ = checkpoint.load_pls(pkg2.PLS(basis2, cost2), path)
    resumed = _steps(basis2, cost2, restored, 3, 3)
    assert torch.equal(resumed, mine), "resume under the shared device gauge != uninterrupted shard"
    # the training loop of the sharded run: the stop rule looks at the mean energy over ALL particles after every step.
    # Pipelined loop + the ranks' host-side exchange (distributed.EnergyMean: shared-memory board) against the plain loop with
    # a blocking all-reduce per iteration: the same particles bit for bit, the same energies to rounding, the same stop
    from projected_langevin_sampling_amd.trainers import train_pls

    eta = 0.5 * float(basis.eigenvalues.min())
    # (particles drawn from the prior, u_m ~ N(0, lambda_m): the chain starts near equilibrium, so the mean energy
    # fluctuates from the first steps on and a short patience stops the run somewhere in the middle)
    ueq = u0 * basis.eigenvalues.cpu().sqrt()[:, None]
    shard0 = ueq[:, j0:j1].contiguous().cuda()
    em = D.EnergyMean(j)
    assert em.uses_board
    runs = {}
    # (the runs that must stop: a step size beyond the stability bound of the stiffest mode, eta / lambda_min = 2.5 > 2 -- its
    # energy grows by 2.25 per step and turns the mean energy round within a few steps; patience = three such steps)
    blocking = lambda e: D.mean_over_particles(e, j)
    for name, red, step, patience in (("board", em, eta, 1e9), ("blocking", blocking, eta, 1e9),
                                      ("board_stop", em, 5.0 * eta, 12.5 * eta), ("blocking_stop", blocking, 5.0 * eta, 12.5 * eta)):
        torch.manual_seed(21)
        runs[name] = train_pls(pls, shard0.clone(), 40, step, patience, energy_reduce=red)
    for a, b in (("board", "blocking"), ("board_stop", "blocking_stop")):
        assert torch.equal(runs[a][0], runs[b][0]), (a, b)
        assert len(runs[a][1]) == len(runs[b][1]) and max(abs(x - y) / abs(y) for x, y in zip(runs[a][1], runs[b][1])) < 1e-12
    train = {k: (v[0].cpu(), v[1]) for k, v in runs.items() if k.startswith("board")}
    # a cost without the Gaussian chunk sums (the mean travels through pls_block_means and an event): same exchange
    from projected_langevin_sampling_amd.costs import PoissonCost
    from projected_langevin_sampling_amd.link_functions import SquareLinkFunction

    counts = torch.poisson(torch.full_like(cost.y_train, 3.0), generator=torch.Generator().manual_seed(8))
    ppls = pkg.PLS(basis, PoissonCost(counts, SquareLinkFunction()))
    pstart = (1.0 + 0.1 * u0[:, j0:j1]).contiguous().cuda()
    torch.manual_seed(22)
    pa = train_pls(ppls, pstart.clone(), 12, 0.2 * eta, 1e9, energy_reduce=em)
    torch.manual_seed(22)
    pb = train_pls(ppls, pstart.clone(), 12, 0.2 * eta, 1e9, energy_reduce=blocking)
    assert torch.equal(pa[0], pb[0]) and len(pa[1]) == len(pb[1]) == 12
    assert max(abs(x - y) / abs(y) for x, y in zip(pa[1], pb[1])) < 1e-12
    # ... and the inducing-point basis' whitened loop for a cost without the Gaussian algebra (16-column energy sums, one launch
    # per iteration): the same exchange, the same particles whichever way the mean travels
    ipkg, ibasis, icost, iu0, ieta, _ = _build_small_ipb()
    D.attach_shard(ibasis, j, rank, world)
    ipls = ipkg.PLS(ibasis, icost)
    assert ibasis.whitened_generic_applies(icost, j1 - j0)
    istart = iu0[:, j0:j1].contiguous().cuda()
    torch.manual_seed(23)
    ia = train_pls(ipls, istart.clone(), 15, ieta, 1e9, energy_reduce=em)
    torch.manual_seed(23)
    ib = train_pls(ipls, istart.clone(), 15, ieta, 1e9, energy_reduce=blocking)
    assert torch.equal(ia[0], ib[0]) and len(ia[1]) == len(ib[1]) == 15
    assert max(abs(x - y) / abs(y) for x, y in zip(ia[1], ib[1])) < 1e-12
    train["ipb_whitened"] = (ia[0].cpu(), ia[1])
    torch.save({"particles": mine.cpu(), "energy": energy, "mean": mean.cpu(), "var": var.cpu(), "samples": samples.cpu(),
                "lam": basis.eigenvalues.cpu(), "vec": basis.eigenvectors.cpu(), "j0": j0, "j1": j1, "train": train},
               os.path.join(out_dir, f"out{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_the_unsharded_run(tmp_path):
    assert torch.cuda.is_available(), "this test needs the MI355X"
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
        outs = [torch.load(tmp_path / f"out{r}.pt", weights_only=False) for r in range(2)]
        assert torch.equal(outs[0]["lam"], outs[1]["lam"]) and torch.equal(outs[0]["vec"], outs[1]["vec"])  # ONE spectrum
        # the unsharded run in this process, with the spectrum the job agreed on
        from projected_langevin_sampling_amd import distributed as D
        from projected_langevin_sampling_amd import samplers

        pkg, basis, cost, xs, j = _build("cuda")
        assert torch.equal(basis.eigenvalues.cpu(), outs[0]["lam"]) and torch.equal(basis.eigenvectors.cpu(), outs[0]["vec"])
        mk = basis.approximation_dimension
        u0 = torch.randn(mk, j, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
        whole = _steps(basis, cost, u0.cuda(), 0, 6)
        got = torch.cat([o["particles"] for o in outs], dim=1)
        assert [(o["j0"], o["j1"]) for o in outs] == [D.shard_bounds(j, r, 2) for r in range(2)]
        # (a shard of 165 columns and the full 330 may take different tile configurations: equal to rounding, not bit for bit)
        assert _relerr(got, whole.cpu()) < 1e-12, "sharded particles != unsharded run"
        pls = pkg.PLS(basis, cost)
        e_whole = pls.particle_energy_potential(whole).mean().item()
        assert all(abs(o["energy"] - e_whole) <= 1e-12 * abs(e_whole) for o in outs)
        # the sharded training loop against the unsharded one: same energies (both ranks hold the SAME floats), same stop
        from projected_langevin_sampling_amd.trainers import train_pls

        eta = 0.5 * float(basis.eigenvalues.min())
        for name, step, patience in (("board", eta, 1e9), ("board_stop", 5.0 * eta, 12.5 * eta)):
            torch.manual_seed(21)
            u_w, e_w = train_pls(pls, (u0 * basis.eigenvalues.cpu().sqrt()[:, None]).cuda(), 40, step, patience)
            assert outs[0]["train"][name][1] == outs[1]["train"][name][1], "the ranks disagree about the energies they stopped on"
            e_s = outs[0]["train"][name][1]
            assert len(e_s) == len(e_w) and max(abs(x - y) / abs(y) for x, y in zip(e_s, e_w)) < 1e-10, name
            assert _relerr(torch.cat([o["train"][name][0] for o in outs], dim=1), u_w.cpu()) < 1e-10, name
        assert len(outs[0]["train"]["board_stop"][1]) < 40, "test construction: the patience never stopped the run"
        # the inducing-point basis' whitened loop, sharded against unsharded
        ipkg, ibasis, icost, iu0, ieta, _ = _build_small_ipb()
        torch.manual_seed(23)
        iu_w, ie_w = train_pls(ipkg.PLS(ibasis, icost), iu0.cuda(), 15, ieta, 1e9)
        assert outs[0]["train"]["ipb_whitened"][1] == outs[1]["train"]["ipb_whitened"][1]
        ie_s = outs[0]["train"]["ipb_whitened"][1]
        assert len(ie_s) == len(ie_w) == 15 and max(abs(x - y) / abs(y) for x, y in zip(ie_s, ie_w)) < 1e-10
        assert _relerr(torch.cat([o["train"]["ipb_whitened"][0] for o in outs], dim=1), iu_w.cpu()) < 1e-10
        prev_stream = samplers.DEFAULT_NORMAL_STREAM
        samplers.DEFAULT_NORMAL_STREAM = "device"
        try:
            torch.manual_seed(9)
            s_whole = pls.predict_samples(whole, xs).cpu()
        finally:
            samplers.DEFAULT_NORMAL_STREAM = prev_stream
        assert _relerr(torch.cat([o["samples"] for o in outs], dim=1), s_whole) < 1e-11, "predictive samples depend on the rank count"
        for o in outs:
            assert torch.allclose(o["mean"], s_whole.mean(dim=1), rtol=1e-12, atol=1e-14)
            assert torch.allclose(o["var"], s_whole.var(dim=1), rtol=1e-10)
    finally:
        torch.set_default_dtype(prev)


def test_bench_line_of_two_ranks_checks_its_own_shards(tmp_path):
    """`PLS_BENCH_BACKEND=gloo python bench.py --gpus 2` on the one card of this box (the control flow of the driver's
    multi-GPU run, RCCL replaced by gloo): the JSON line carries the process group as the ranks saw it, every rank's own
    clock, and a shard check -- the ranks' gathered particles are, bit for bit, what rank 0 computes for the same columns
    (distributed.py:1-7), energies included."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLS_BENCH_BACKEND="gloo")
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--sustained-steps", "0", "--profiler-steps", "0", "--ipb-steps", "0",
                          "--converge-steps", "0"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0, run.stderr[-3000:]
    line = json.loads(run.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert [r["rank"] for r in line["ranks_seen"]] == [0, 1]
    per = line["per_rank_ms_per_step"]
    assert len(per["all"]) == 2 and per["min"] <= per["max"] and per["slowest_rank"] in (0, 1)
    assert abs(per["max"] - line["ms_per_step"]) < 1e-6 * line["ms_per_step"]  # the line's value is the slowest rank's
    chk = line["shard_check"]
    assert chk["max_abs_diff"] == 0.0 and chk["energies_equal"] is True and chk["all_reduce_of_ones_equals_world"] is True
    assert [s["columns"] for s in chk["shards"]] == [[0, 4096], [4096, 8192]]
    assert chk["steps"]["gaussian_fast_path"] == 20 and chk["steps"]["like_for_like"] >= 2
    assert chk["max_rel_diff_vs_one_unsharded_matrix"] < 1e-12
    assert abs(chk["mean_energy_all_reduce"] - chk["mean_energy_rank0_recomputed"]) <= 1e-12 * abs(chk["mean_energy_rank0_recomputed"])
