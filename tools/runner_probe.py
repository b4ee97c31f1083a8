"""The step-size search (experiments/runners.py:331-446) at configs[1] data sizes: S candidate step sizes as
S column blocks of one launch per epoch (batched=True) against the candidates trained one after the other with train_pls
(batched=False: what the reference's loop does, on the same kernels) and the default choice (by the step's size).  N = 1e5, M_k = 1024, Gaussian; J particles per
candidate, metric "loss" (no prediction in the timed region)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import OrthonormalBasis
from projected_langevin_sampling_amd.costs import GaussianCost
from projected_langevin_sampling_amd.link_functions import IdentityLinkFunction
from projected_langevin_sampling_amd.runners import train_pls_runner

torch.set_default_dtype(torch.float64)
mk, n = 1024, 100000
g = torch.Generator().manual_seed(0)
a = (torch.randn(mk, n, generator=g) / mk ** 0.5).cuda()
lam = (torch.rand(mk, generator=g) + 0.5).cuda()
basis = OrthonormalBasis.from_projection(a, lam)
y = torch.randn(n, generator=g)
cost = GaussianCost(0.5, y, IdentityLinkFunction())
pls = P.PLS(basis, cost)
x_dummy = torch.zeros(4, 1)
for j, s in ((256, 8), (1024, 8), (1024, 4)):
    u = torch.randn(mk, j, generator=g).cuda()
    kw = dict(pls=pls, particle_name="probe", x_train=x_dummy, y_train=y[:4], simulation_duration=2e-4, maximum_number_of_steps=2000,
              early_stopper_patience=1e9, number_of_step_searches=s, step_size_upper=2e-6, minimum_change_in_energy_potential=0.0,
              seed=0, metric_to_optimise="loss")
    out = {}
    for name, kind in (("blocks", True), ("one by one", False), ("default", None), ("blocks", True), ("one by one", False), ("default", None)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        best, lr, epochs = train_pls_runner(particles=u.clone(), batched=kind, **kw)
        torch.cuda.synchronize(); out[name] = time.perf_counter() - t0
        print(f"J = {j:5d} S = {s}: {name:11s} {out[name]:7.3f} s  (best step size {lr:.3e}, {epochs} epochs)", flush=True)
