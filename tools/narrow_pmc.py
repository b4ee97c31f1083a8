"""200 launches each of the Gaussian fast-path step and the bare contraction at M_k = 1024 for J = 1024 and 2048 (k-split 64 x 64
kernel): a target for `tools/profile_cmd.sh <tag> <counters> -- python3 tools/narrow_pmc.py`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction

lib = L.load()
torch.manual_seed(0)
mk, n = 1024, 4096
a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
basis = OrthonormalBasis.from_projection(a, torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5)
cost = GaussianCost(0.5, torch.randn(n, dtype=torch.float64), IdentityLinkFunction())
for j in (1024, 2048):
    bufs = [torch.randn(mk, j, dtype=torch.float64, device="cuda"), torch.empty(mk, j, dtype=torch.float64, device="cuda")]
    c = torch.empty(mk, j, dtype=torch.float64, device="cuda")
    for s in range(200):
        basis.fused_step(cost, bufs[s & 1], 1e-6, out=bufs[(s + 1) & 1], new_state=True, noise=NoiseSpec(seed=1, step=s))
    B = basis._B
    for s in range(200):
        lib.pls_gemm_tn(B.data_ptr(), L.ld(B), bufs[0].data_ptr(), j, c.data_ptr(), j, mk, j, mk, 1.0, 0.0, L.stream_ptr())
    torch.cuda.synchronize()
