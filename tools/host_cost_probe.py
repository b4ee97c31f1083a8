"""Host cost of one train_pls iteration on a narrow particle shard (M_k = 1024, J = 1024: the GPU side is one ~41 us
launch): the loop with the GPU idle-free (wall per iteration), the same Python path with the launch stubbed out (pure host
cost per iteration), a cProfile of it, and what a CapturedTraining costs to build."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
from projected_langevin_sampling_amd.trainers import train_pls, train_pls_captured
from projected_langevin_sampling_amd.graph import CapturedTraining

torch.set_default_dtype(torch.float64)
mk, n, j = 1024, 20000, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator().manual_seed(0)
a = (torch.randn(mk, n, generator=g) / mk ** 0.5).cuda()
lam = (torch.rand(mk, generator=g) + 0.5).cuda()
basis = OrthonormalBasis.from_projection(a, lam)
y = torch.randn(n, generator=g)
cost = GaussianCost(0.5, y, IdentityLinkFunction())
pls = P.PLS(basis, cost)
u = torch.randn(mk, j, generator=g).cuda()
train_pls(pls, u.clone(), 20, 1e-7, 1e9)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, e = train_pls(pls, u.clone(), 4000, 1e-7, 1e9)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    print(f"J = {j}: train_pls {len(e)} iterations, {wall / len(e) * 1e6:.2f} us per iteration", flush=True)
for k in (16, 64):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _, e = train_pls_captured(pls, u.clone(), 4000, 1e-7, 1e9, steps_per_replay=k, seed=1)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        print(f"J = {j}: train_pls_captured({k} per replay) {len(e)} iterations, {wall / len(e) * 1e6:.2f} us per iteration incl. capture", flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cap = CapturedTraining(pls, u.clone(), 1e-7, k, seed=1)
    torch.cuda.synchronize(); t_build = time.perf_counter() - t0
    cap.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        cap.replay()
    torch.cuda.synchronize(); t_rep = (time.perf_counter() - t0) / 20
    print(f"   CapturedTraining({k}): build {t_build * 1e3:.2f} ms, replay {t_rep * 1e6:.1f} us = {t_rep / k * 1e6:.2f} us per iteration", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
train_pls(pls, u.clone(), 2000, 1e-7, 1e9)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
