"""CPU, world_size 2 over gloo: the J-sharded layout and the three reduction points.
The Langevin step itself has no collective; what is distributed is the bookkeeping around it."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, j, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from projected_langevin_sampling_amd import distributed as D

    g = torch.Generator().manual_seed(0)
    full = torch.randn(6, j, generator=g, dtype=torch.float64)  # same on every rank (seeded, like initialise_particles)
    samples_full = torch.randn(4, j, generator=g, dtype=torch.float64)
    j0, j1 = D.shard_bounds(j, rank, world)
    local = D.shard_particles(full, rank, world)
    assert local.shape[1] == j1 - j0 and torch.equal(local, full[:, j0:j1])

    class B:
        j_offset = -1

    b = B()
    assert D.attach_shard(b, j, rank, world) == (j0, j1) and b.j_offset == j0
    # C1: mean energy over all particles
    e_local = (local * local).sum(dim=0)
    mean = D.mean_over_particles(e_local, j)
    assert np.isclose(mean, (full * full).sum(dim=0).mean().item(), rtol=1e-13)
    # C2: predictive moments
    m, v = D.predictive_moments(samples_full[:, j0:j1], j)
    assert np.allclose(m, samples_full.mean(dim=1), rtol=1e-12) and np.allclose(v, samples_full.var(dim=1), rtol=1e-10)
    # C3: all-gather of ragged shards
    gathered = D.gather_particles(local, j)
    assert torch.equal(gathered, full)
    # C3 as the conformal wrapper uses it: quantiles over all J samples of every test point, rows dealt out over the ranks
    # (all-to-all), each rank sorts its share, quantiles all-gathered -- bit for bit the single-process quantiles
    for n_star in (7, 4, 1):
        pred_full = torch.randn(n_star, j, generator=torch.Generator().manual_seed(5 + n_star), dtype=torch.float64)
        qs = [0.05, 0.5, 0.95]
        want = torch.quantile(pred_full, torch.tensor(qs, dtype=torch.float64), dim=1).T
        got = D.sharded_row_quantiles(pred_full[:, j0:j1].contiguous(), qs)
        assert got.shape == (n_star, 3) and torch.equal(got, want), (n_star, rank)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def _run(world, j, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, j, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_two_ranks_even_shards(tmp_path):
    _run(2, 64, tmp_path)


def test_two_ranks_ragged_shards(tmp_path):
    _run(2, 37, tmp_path)


def test_four_ranks_ragged_shards(tmp_path):
    """the shape of the 4-GPU configuration (configs[3]) in miniature: J not divisible by the world size"""
    _run(4, 37, tmp_path)


def test_three_ranks_fewer_particles_than_some_shards_expect(tmp_path):
    """a shard may be a single column (J = 4 over 3 ranks: 2, 1, 1)"""
    _run(3, 4, tmp_path)


def _spectrum_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from projected_langevin_sampling_amd.basis import spectrum as S

    g = torch.Generator().manual_seed(11)
    a = torch.randn(40, 40, generator=g, dtype=torch.float64)
    gram = a @ a.T / 40
    if rank == 1:  # this rank's Gram matrix rounded differently: one ulp in one entry
        gram[3, 3] = torch.nextafter(gram[3, 3], torch.tensor(float("inf"), dtype=torch.float64))
    own = torch.linalg.eigh(gram)
    for canonical in (None, True):
        lam, vec = S.shared_spectrum(gram, "cpu", canonical_signs=canonical, group=True)
        both = [torch.empty(41, 40, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, torch.cat([lam[None, :], vec], dim=0))
        assert all(torch.equal(b, both[0]) for b in both), "ranks hold different spectra"
        if rank == 0:  # rank 0's own decomposition, signs as asked
            assert torch.equal(lam, own[0]) and torch.equal(vec, S.canonicalise_signs(own[1]) if canonical else own[1])
        # the same count on every rank at a threshold that sits ON an eigenvalue of rank 0's spectrum
        mk = int((lam > lam[7]).sum())
        S.assert_same_count(mk, group=True)
    # the default is the reference's behaviour: a local eigh, NO collective -- so a basis ONE rank builds on its own (a rank-0
    # evaluation inside a running job) returns instead of waiting for peers that never come
    if rank == 0:
        lam_alone, vec_alone = S.shared_spectrum(gram, "cpu")
        assert torch.equal(lam_alone, own[0]) and torch.equal(vec_alone, own[1])
        S.assert_same_count(5)
    lam_own, vec_own = S.shared_spectrum(gram, "cpu", group=False)
    assert torch.equal(lam_own, own[0]) and torch.equal(vec_own, own[1])
    S.assert_same_count(rank, group=False)
    # ranks that disagree are told so (every rank raises: the reduction is symmetric)
    try:
        S.assert_same_count(30 + rank, group=True)
        raise SystemExit("assert_same_count accepted different counts")
    except RuntimeError as e:
        assert "disagree" in str(e)
    # ranks that were handed different matrices do not adopt rank 0's spectrum: other hyper-parameters ...
    try:
        S.shared_spectrum(gram * (1.0 + 0.01 * rank), "cpu", group=True)
        raise SystemExit("shared_spectrum accepted different Gram matrices")
    except RuntimeError as e:
        assert "different k(Z,Z)" in str(e)
    # ... or another number of inducing points (the broadcast would have hung on mismatched shapes)
    try:
        S.shared_spectrum(gram[: 40 - rank, : 40 - rank], "cpu", group=True)
        raise SystemExit("shared_spectrum accepted different M")
    except RuntimeError as e:
        assert "different numbers of inducing points" in str(e)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_one_spectrum_for_all_ranks(tmp_path):
    """OrthonormalBasis(group=True) under torch.distributed (basis/spectrum.py): rank 0 factorises k(Z,Z)/M and broadcasts,
    so a rank whose eigh input differs by one ulp still ends with the same eigenvalues, eigenvectors and count, bit for bit
    (reference: ONE process, ONE torch.linalg.eigh, orthonormal.py:46-68).  Sharing is opt-in: without ``group`` a rank
    building alone does not hang, and ranks holding different matrices (or different M) are refused."""
    port = _free_port()
    mp.spawn(_spectrum_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(2))


def _subgroup_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from projected_langevin_sampling_amd import distributed as D
    from projected_langevin_sampling_amd.basis import spectrum as S

    members = [1, 3]  # a sub-group whose group ranks (0, 1) are NOT its global ranks
    sub = dist.new_group(ranks=members)
    if rank in members:
        r, j = members.index(rank), 23
        pred_full = torch.randn(5, j, generator=torch.Generator().manual_seed(2), dtype=torch.float64)
        j0, j1 = D.shard_bounds(j, r, 2)
        qs = [0.1, 0.5, 0.9]
        want = torch.quantile(pred_full, torch.tensor(qs, dtype=torch.float64), dim=1).T
        got = D.sharded_row_quantiles(pred_full[:, j0:j1].contiguous(), qs, group=sub)
        assert torch.equal(got, want), rank
        a = torch.randn(9, 9, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
        gram = a @ a.T
        gram[2, 2] = gram[2, 2] * (1.0 + 1e-15 * rank)  # the same matrix up to rounding, different bits per rank
        lam, vec = S.shared_spectrum(gram, "cpu", group=sub)
        both = [torch.empty(10, 9, dtype=torch.float64) for _ in range(2)]
        dist.all_gather(both, torch.cat([lam[None, :], vec], dim=0), group=sub)
        assert torch.equal(both[0], both[1])  # group rank 0 (global rank 1) decided
        S.assert_same_count(4, group=sub)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_reductions_inside_a_sub_group(tmp_path):
    """ConformalisePLS / OrthonormalBasis forward a caller's process group: peers of the point-to-point exchange and the
    broadcast source are GROUP ranks and must be mapped to global ranks (4 processes, group = global ranks 1 and 3)."""
    port = _free_port()
    mp.spawn(_subgroup_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(4))


def _energy_board_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time

    from projected_langevin_sampling_amd import distributed as D

    j = 1000
    board = D.EnergyMean(j, board=True)
    plain = D.EnergyMean(j, board=False)
    assert board.uses_board and not plain.uses_board
    rng = np.random.default_rng(7)  # the same stream on every rank: column r is rank r's local sum
    vals = rng.standard_normal((400, world)) * 10.0 ** rng.integers(-3, 6, size=(400, 1))
    got = []
    for t in range(400):  # 25 times round the ring of 16, the ranks drifting apart and catching up
        if (t + rank) % 37 == 0:
            time.sleep(0.002)
        got.append(board.reduce_local_sum(vals[t, rank]))
    want = []
    for t in range(400):
        tot = 0.0
        for r in range(world):
            tot += float(vals[t, r])
        want.append(tot / j)
    assert got == want, "rank-order sum, bit for bit"
    for t in (0, 1, 2):  # the all-reduce fallback agrees to rounding
        assert np.isclose(plain.reduce_local_sum(vals[t, rank]), want[t], rtol=1e-13, atol=1e-300)
    # called with a tensor it is the blocking collective
    e_local = torch.full((10,), float(rank + 1), dtype=torch.float64)
    assert np.isclose(board(e_local), 10.0 * sum(range(1, world + 1)) / j)
    mine = torch.tensor(got, dtype=torch.float64)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    assert all(torch.equal(everyone[0], e) for e in everyone), "every rank holds the same bits (the same stop decisions)"
    dist.barrier()
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("pls_energy_")], "the board's file name is gone (rank 0 unlinks it)"
    # a rank that leaves its loop through an exception (iteration 5) says so on the board: its peers raise at once instead of
    # polling for ``timeout_s`` (600 s by default) -- the way out trainers._train_pls_in_flight takes
    again = D.EnergyMean(j, board=True, timeout_s=120.0)
    t0 = time.monotonic()
    try:
        for t in range(20):
            if rank == 1 and t == 5:
                again.abort("ValueError('a failing launch')")
                raise ValueError("a failing launch")
            again.reduce_local_sum(float(t))
        raise SystemExit("the peers of an aborted rank went on")
    except ValueError:
        assert rank == 1
    except RuntimeError as e:
        assert rank != 1 and "rank(s) [1] left the training loop" in str(e), str(e)
    assert time.monotonic() - t0 < 10.0, "peers must learn of the abort from the board, not from the time-out"
    try:  # and the exchange stays unusable: nobody takes half a ring for a result
        again.reduce_local_sum(0.0)
        raise SystemExit("an aborted exchange accepted another value")
    except RuntimeError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_energy_mean_over_a_shared_memory_board(tmp_path):
    """distributed.EnergyMean: the per-iteration exchange of the ranks' local energy sums through /dev/shm (ranks of one
    node), against the sum in rank order and the all-reduce fallback; 3 ranks, 400 iterations, deliberately skewed ranks;
    then a rank that raises at iteration 5: its peers fail within seconds (the abort word), not after the time-out."""
    port = _free_port()
    mp.spawn(_energy_board_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(3))
