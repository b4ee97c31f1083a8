"""Ten like-for-like steps at one rank (default 160), N = 1e5, J = 8192, Gaussian: the workload of the rocprofv3 counter passes
behind `traffic per step` in profiles/r03_step_sweep_ranks.txt (tools/profile_cmd.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
mk = int(sys.argv[1]) if len(sys.argv) > 1 else 160
n, j = 100000, 8192
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
basis = OrthonormalBasis.from_projection(a, lam)
basis.workspace_bytes = 8 << 30
cost = GaussianCost(0.5, torch.randn(n, dtype=torch.float64), IdentityLinkFunction())
u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
out = torch.empty_like(u)
for _ in range(10):
    basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
torch.cuda.synchronize()
