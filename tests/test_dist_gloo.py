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
