"""Where does a train_pls iteration go at configs[1] (Gaussian fast path)?  Wall time of the pipelined loop, three
repetitions in one process (and, for loops that wait on events, the time spent in Event.synchronize())."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
from projected_langevin_sampling_amd.trainers import train_pls

torch.set_default_dtype(torch.float64)
mk, n, j = 1024, 100000, int(sys.argv[1]) if len(sys.argv) > 1 else 8192
g = torch.Generator().manual_seed(0)
a = (torch.randn(mk, n, generator=g) / mk ** 0.5).cuda()
lam = (torch.rand(mk, generator=g) + 0.5).cuda()
basis = OrthonormalBasis.from_projection(a, lam)
y = torch.randn(n, generator=g)
cost = GaussianCost(0.5, y, IdentityLinkFunction())
pls = P.PLS(basis, cost)
u = torch.randn(mk, j, generator=g).cuda()
wait = [0.0, 0]
orig = torch.cuda.Event.synchronize
def timed_sync(self):
    t0 = time.perf_counter(); orig(self); wait[0] += time.perf_counter() - t0; wait[1] += 1
train_pls(pls, u.clone(), 5, 1e-7, 1e9)
for rep in range(3):
    torch.cuda.Event.synchronize = timed_sync
    wait[0], wait[1] = 0.0, 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, e = train_pls(pls, u.clone(), 1000, 1e-7, 1e9)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    torch.cuda.Event.synchronize = orig
    print(f"J = {j}: {len(e)} iterations, {wall / len(e) * 1e3:.4f} ms per iteration; host waited in Event.synchronize {wait[0] / max(wait[1], 1) * 1e3:.4f} ms "
          f"per iteration ({wait[1]} waits) -> host busy {(wall - wait[0]) / len(e) * 1e3:.4f} ms per iteration", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
train_pls(pls, u.clone(), 300, 1e-7, 1e9)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
