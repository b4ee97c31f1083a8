"""Gaussian fast path (B = A A^T) step and stand-alone energy over (M_k, J): time and TF/s (2 M_k^2 J flop)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
torch.manual_seed(0)
n = 4000
print(f"{'M_k':>5s} {'J':>6s}  {'step ms':>8s} {'TF/s':>6s}  {'step+E ms':>9s}  {'energy ms':>9s}")
for mk in (16, 64, 128, 129, 256, 512, 1024, 2048, 4096):
    for j in (64, 512, 2048, 8192):
        a = torch.randn(mk, n, dtype=torch.float64, device="cuda") / mk ** 0.5
        lam = torch.rand(mk, dtype=torch.float64, device="cuda") + 0.5
        basis = OrthonormalBasis.from_projection(a, lam)
        y = torch.randn(n, dtype=torch.float64)
        cost = GaussianCost(0.5, y, IdentityLinkFunction())
        u = torch.randn(mk, j, dtype=torch.float64, device="cuda")
        out = torch.empty_like(u); e = torch.empty(j, dtype=torch.float64, device="cuda")
        def t(f, reps=20):
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): f()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        s1 = t(lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(seed=1, step=2)))
        s2 = t(lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, noise=NoiseSpec(seed=1, step=2), input_energy=e))
        s3 = t(lambda: basis.fused_particle_energy(cost, u))
        print(f"{mk:5d} {j:6d}  {s1:8.4f} {2.0 * mk * mk * j / s1 / 1e9:6.1f}  {s2:9.4f}  {s3:9.4f}", flush=True)
