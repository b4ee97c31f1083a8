"""Is the batched step-size search at J = 256 'bimodal' (profiles/r03_runner_probe.txt: 0.340 s and 0.177 s for the same call)
or is the first call of a process paying one-time costs?  The costs a first call can carry are timed on their own (first
pinned allocation, Gaussian constants B = A A^T, first launch of each kernel), then the two routes alternate six times."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter(); torch.zeros(1).cuda(); torch.cuda.synchronize(); print(f"HIP context: {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter(); torch.empty(64, dtype=torch.float64).pin_memory(); print(f"first pinned allocation: {time.perf_counter() - t0:.3f} s")
t0 = time.perf_counter(); torch.empty(64, dtype=torch.float64).pin_memory(); print(f"second pinned allocation: {time.perf_counter() - t0:.6f} s")
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
torch.cuda.synchronize(); t0 = time.perf_counter(); basis.prepare_gaussian(cost.y_device()); torch.cuda.synchronize()
print(f"Gaussian constants (B = A A^T at N = 1e5, first libplship launches included): {time.perf_counter() - t0:.3f} s")
x_dummy = torch.zeros(4, 1)
j, s = 256, 8
u = torch.randn(mk, j, generator=g).cuda()
kw = dict(pls=pls, particle_name="probe", x_train=x_dummy, y_train=y[:4], simulation_duration=2e-4, maximum_number_of_steps=2000,
          early_stopper_patience=1e9, number_of_step_searches=s, step_size_upper=2e-6, minimum_change_in_energy_potential=0.0,
          seed=0, metric_to_optimise="loss")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
train_pls_runner(particles=u.clone(), batched=True, **kw)
torch.cuda.synchronize()
pr.disable()
print("---- the FIRST block launch of the process under cProfile (top by own time) ----")
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
for rep in range(6):
    for name, kind in (("blocks", True), ("one by one", False)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        best, lr, epochs = train_pls_runner(particles=u.clone(), batched=kind, **kw)
        torch.cuda.synchronize()
        print(f"rep {rep} J = {j} S = {s}: {name:11s} {time.perf_counter() - t0:7.3f} s  ({epochs} epochs)", flush=True)
