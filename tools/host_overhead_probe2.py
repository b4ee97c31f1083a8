"""Where does the host time of one fused_step call go?  cProfile over 2000 eager calls of the Gaussian fast path on a small
problem (the GPU work is a few microseconds: the loop is host-bound), plus the plain per-call wall time."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

torch.manual_seed(0)
mk, n, j = 64, 512, 256
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
basis = OrthonormalBasis.from_projection(a, torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5)
cost = GaussianCost(0.5, torch.randn(n, dtype=torch.float64), IdentityLinkFunction())
u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
out = torch.empty_like(u)
en = torch.empty(j, dtype=torch.float64, device="cuda")


def loop(k, energy):
    for s in range(k):
        basis.fused_step(cost, u, 1e-6, out=out, new_state=True, noise=NoiseSpec(seed=1, step=s), input_energy=en if energy else None)


loop(50, True)
torch.cuda.synchronize()
for energy in (False, True):
    t0 = time.perf_counter()
    loop(2000, energy)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"eager fused_step, energy={energy}: {(t1 - t0) / 2000 * 1e6:.1f} us of host time per call", flush=True)
pr = cProfile.Profile()
pr.enable()
loop(2000, True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
