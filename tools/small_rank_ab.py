"""A/B of two builds of libplship (PLSHIP_LIBRARY) on the fused small-rank step: N = 5e4, J = 16384, ranks 16 .. 64, Gaussian and
Poisson.  Run once per library; prints ms per step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction
n, j = 50000, 16384
for mk in (16, 32, 48, 64, 89):
    a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
    lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
    basis = OrthonormalBasis.from_projection(a, lam)
    y = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
    u = torch.randn(mk, j, dtype=torch.float64, device="cuda").abs() + 0.5
    out = torch.empty_like(u)
    for cname, cost in (("gaussian", GaussianCost(0.5, y, IdentityLinkFunction())), ("poisson", PoissonCost(y, SquareLinkFunction()))):
        f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{os.environ.get('PLSHIP_LIBRARY', 'default'):40s} M_k {mk:3d} {cname:8s} {ms:7.3f} ms/step  ({4.0 * n * mk * j / ms / 78.6e9:.3f} of peak)", flush=True)
