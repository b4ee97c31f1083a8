"""Forward GEMM + cost-derivative epilogue at a configs[3]-like shape, per cost, through the C ABI step with the
back-projection excluded by the timeline (gemm_cost_deriv tag only).  PLSHIP_LIBRARY selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, BernoulliCost, PoissonCost, StudentTCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SigmoidLinkFunction, SquareLinkFunction, ProbitLinkFunction
torch.manual_seed(0)
n, mk, j = 50000, 2048, 8192
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
basis = OrthonormalBasis.from_projection(a, lam)
basis.workspace_bytes = 8 << 30
yb = (torch.rand(n) < 0.5).double()
yc = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
yr = torch.randn(n, dtype=torch.float64)
u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
out = torch.empty_like(u)
e = torch.empty(j, dtype=torch.float64, device="cuda")
costs = {"gaussian/identity": GaussianCost(0.5, yr, IdentityLinkFunction()), "bernoulli/sigmoid": BernoulliCost(yb, SigmoidLinkFunction()),
         "bernoulli/probit": BernoulliCost(yb, ProbitLinkFunction()), "poisson/square": PoissonCost(yc, SquareLinkFunction()),
         "student_t/identity": StudentTCost(3.0, yr, IdentityLinkFunction(), 0.7)}
for name, cost in costs.items():
    for with_e in (False, True):
        f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True,
                                     input_energy=e if with_e else None)
        for _ in range(2): f()
        torch.cuda.synchronize()
        with L.Timeline(capacity=256) as tl:
            for _ in range(4): f()
        s = tl.summary()
        print(f"{name:20s} energy by-product {str(with_e):5s}: forward+epilogue {s['gemm_cost_deriv']['avg_ms']:.3f} ms   back-projection {s['gemm_store']['avg_ms']:.3f} ms", flush=True)
