"""Host-side cost of one eager fused step at a launch-bound size (configs[0]): wall per step vs kernel time, and a
cProfile of where the Python time goes."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as pkg
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction
torch.manual_seed(0)
n, m, j = 100, 10, 64
x = torch.linspace(-1, 1, n, dtype=torch.float64)[:, None]
z = x[::10].clone()
y = torch.sin(6.28 * x[:, 0]) + 0.1 * torch.randn(n, dtype=torch.float64)
basis = OrthonormalBasis(pkg.PLSKernel(pkg.ARDKernel(torch.tensor([0.15], dtype=torch.float64), 3.0), z), z, x, verbose=False)
for name, cost, fg in (("gaussian fast path", GaussianCost(0.25, y, IdentityLinkFunction()), False),
                       ("poisson generic", PoissonCost(torch.poisson(torch.rand(n, dtype=torch.float64) * 3), SquareLinkFunction()), True)):
    pls = pkg.PLS(basis, cost)
    u = pls.initialise_particles(j, seed=0)
    def loop(k):
        for _ in range(k):
            pls.step_(u, 1e-4)
    loop(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(2000); t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"{name}: host issue {t_host / 2000 * 1e6:.1f} us/step, wall {t_all / 2000 * 1e6:.1f} us/step", flush=True)
    pr = cProfile.Profile(); pr.enable(); loop(1000); pr.disable(); torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(12)
