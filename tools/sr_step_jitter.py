"""Run-to-run spread of the one-launch training iteration at N = 4096, M = 128, J = 512 (a grid of exactly 256 workgroups at one
workgroup per CU): `train_pls` of 2000 iterations, repeated, per basis and cost."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd.basis import InducingPointBasis, OrthonormalBasis
from projected_langevin_sampling_amd.costs import BernoulliCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction, SquareLinkFunction
from projected_langevin_sampling_amd.trainers import train_pls
torch.set_default_dtype(torch.float64)
print("CUs:", torch.cuda.get_device_properties(0).multi_processor_count, flush=True)
n, m, d = 4096, 128, 4
g = torch.Generator().manual_seed(0)
x = torch.rand(n, d, generator=g) * 2 - 1
z = x[torch.randperm(n, generator=g)[:m]].clone()
y = torch.sin(2.0 * x.sum(dim=1)) + 0.1 * torch.randn(n, generator=g)
kern = P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z)
bases = (("onb", OrthonormalBasis(kern, z, x, 1e-8, verbose=False)), ("ipb", InducingPointBasis(kern, z, y[:m], x)))
costs = (("bernoulli", BernoulliCost((y > 0).double(), SigmoidLinkFunction())), ("poisson", PoissonCost(torch.poisson(y * y + 0.5, generator=g), SquareLinkFunction())))
for j in (512, 496, 448):
    for bname, basis in bases:
        for cname, cost in costs:
            pls = P.PLS(basis, cost)
            u = (1.0 + 0.1 * torch.randn(basis.approximation_dimension, j, generator=g)).cuda()
            train_pls(pls, u.clone(), 30, 1e-9, 1e9)
            ts = []
            for rep in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                _, e = train_pls(pls, u.clone(), 2000, 1e-9, 1e9)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / len(e) * 1e6)
            print(f"J={j} {bname} {cname:9s}: us per iteration over 8 runs: " + " ".join(f"{t:6.1f}" for t in ts), flush=True)
