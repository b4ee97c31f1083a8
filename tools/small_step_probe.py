"""One step at a UCI-sized problem (N = 4096, M = 128, J = 512) for three costs, with and without the energy by-product: time per
step in a hipGraph replay and the library timeline's per-launch durations (fused small-rank kernel, update kernel)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import projected_langevin_sampling_amd as P
from projected_langevin_sampling_amd import _lib as L
from projected_langevin_sampling_amd.basis import OrthonormalBasis, NoiseSpec
from projected_langevin_sampling_amd.costs import BernoulliCost, GaussianCost, PoissonCost
from projected_langevin_sampling_amd.link_functions import SigmoidLinkFunction, IdentityLinkFunction, SquareLinkFunction
torch.set_default_dtype(torch.float64)
n, m, j, d = 4096, 128, 512, 4
g = torch.Generator().manual_seed(0)
x = torch.rand(n, d, generator=g) * 2 - 1
z = x[torch.randperm(n, generator=g)[:m]].clone()
f = torch.sin(2.0 * x.sum(dim=1))
basis = OrthonormalBasis(P.PLSKernel(P.ARDKernel(torch.full((d,), 0.5), 1.0), z.cuda()), z.cuda(), x.cuda(), 1e-8, verbose=False)
u = (1.0 + 0.1 * torch.randn(basis.approximation_dimension, j, generator=g)).cuda()
out = torch.empty_like(u); e = torch.empty(j, device="cuda")
costs = {"gaussian(generic)": (GaussianCost(0.1, f, IdentityLinkFunction()), True),
         "poisson/square": (PoissonCost(torch.poisson(f * f + 0.5, generator=g), SquareLinkFunction()), False),
         "bernoulli/sigmoid": (BernoulliCost((f > 0).double(), SigmoidLinkFunction()), False)}
for name, (cost, fg) in costs.items():
    for with_e in (False, True):
        call = lambda: basis.fused_step(cost, u, 1e-9, out=out, new_state=True, force_generic=fg, noise=NoiseSpec(seed=1, step=2), input_energy=e if with_e else None)
        for _ in range(5): call()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20): call()
        torch.cuda.current_stream().wait_stream(side)
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): gr.replay()
        e1.record(); torch.cuda.synchronize()
        with L.Timeline(256) as tl:
            for _ in range(10): call()
        print(f"{name:20s} energy by-product {with_e!s:5s}: {e0.elapsed_time(e1) / 400 * 1e3:7.2f} us per step in a graph | per launch (with events): "
              + ", ".join(f"{k} {v['avg_ms'] * 1e3:.1f}" for k, v in tl.summary().items()), flush=True)
