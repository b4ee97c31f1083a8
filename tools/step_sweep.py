"""Achieved TF/s of the like-for-like step (4 N M_k J flop) over a grid of sizes: looks for cliffs between the kernel
families (fused small-rank <= 128, 64x64 tiles, 128x128 tiles, split-K)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction, SquareLinkFunction
torch.manual_seed(0)
print(f"{'N':>8s} {'M_k':>5s} {'J':>6s}  {'gauss ms':>9s} {'TF/s':>6s}  {'poisson ms':>10s} {'TF/s':>6s}")
for n in (2000, 20000, 100000):
    for mk in (32, 96, 128, 129, 192, 256, 512, 1024):
        for j in (256, 1024, 8192):
            if n * mk * j > 2e12: continue
            a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
            lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
            basis = OrthonormalBasis.from_projection(a, lam)
            y = torch.poisson(torch.rand(n, dtype=torch.float64) * 4)
            u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
            out = torch.empty_like(u)
            res = []
            for cost in (GaussianCost(0.5, y, IdentityLinkFunction()), PoissonCost(y, SquareLinkFunction())):
                f = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(none=True), force_generic=True)
                for _ in range(3): f()
                torch.cuda.synchronize()
                reps = 5 if n * mk * j > 1e10 else 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): f()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                res.append((ms, 4.0 * n * mk * j / ms / 1e9))
            print(f"{n:8d} {mk:5d} {j:6d}  {res[0][0]:9.3f} {res[0][1]:6.1f}  {res[1][0]:10.3f} {res[1][1]:6.1f}", flush=True)
            del a, basis, u, out
